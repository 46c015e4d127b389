#!/bin/bash
# The exact fp32 path (bench.py --exact: every convolution on the fmaf-chain MFMA kernels) between settings, interleaved, three rounds on one box.
cd $GRAFT_REPO_ROOT
for rnd in 1 2 3; do
  for setting in "$@"; do
    out=$(env $setting python3 bench.py --exact --no-cpu-baseline --no-fast-mode --sustained-frames 0 --steps 30 --warmup 5 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%.1f frames/s %.3f ms' % (d['value'], d['ms_per_step']))")
    echo "round $rnd [$setting] $out"
  done
done
