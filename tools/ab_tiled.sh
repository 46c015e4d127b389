#!/bin/bash
# In-frame A/B of experiment switches for bench.py --mode tiled (1024^3 -> 4K on one GPU), interleaved, two rounds on ONE box.
# usage: bash tools/ab_tiled.sh "VAR=VAL ..." "VAR=VAL ..." ...        (BENCH_ALLOW_SWITCHES=1 is set: debug switches are the point here)
cd $GRAFT_REPO_ROOT
for rnd in 1 2; do for s in "$@"; do
 out=$(env BENCH_ALLOW_SWITCHES=1 $s python3 bench.py --mode tiled --no-cpu-baseline --steps 20 --warmup 4 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.1f fps %.3f ms | %s' % (d['value'], d['ms_per_step'], ' '.join('%s=%.3f' % (k.replace('conv3x3_','').replace('_kernel',''), v['ms_per_frame']) for k,v in d['roofline']['kernels'].items())))")
 echo "round $rnd [$s] $out"; done; done
