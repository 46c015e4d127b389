#!/bin/bash
# In-frame A/B of experiment switches: bench.py (no extra legs) per setting, interleaved, three rounds, in ONE process group on one
# box (boxes differ by several percent).  usage: bash tools/ab_bench.sh "VAR=VAL ..." "VAR=VAL ..." ...
cd $GRAFT_REPO_ROOT
for rnd in 1 2 3; do
  for setting in "$@"; do
    out=$(env $setting python3 bench.py --no-cpu-baseline --no-fast-mode --no-exact-leg --steps 40 --warmup 15 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%.1f frames/s %.3f ms | %s' % (d['value'], d['ms_per_step'], ' '.join('%s=%.3f' % (k.replace('conv3x3_','').replace('_kernel',''), v['ms_per_frame']) for k,v in (d.get('kernels') or {}).items())))")
    echo "round $rnd [$setting] $out"
  done
done
