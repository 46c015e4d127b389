#!/bin/bash
# How often is a bench run slow as a whole (2.8 instead of 1.85 ms per frame, seen in ~1 of 9 runs), and does the side stream's
# priority class change it?  usage: bash tools/outlier_hunt.sh N "VAR=VAL ..." "VAR=VAL ..."
cd $GRAFT_REPO_ROOT
N=$1; shift
for setting in "$@"; do
  for i in $(seq 1 $N); do
    env $setting BENCH_PROFILE_TIMED=0 python3 bench.py --no-cpu-baseline --no-fast-mode --no-exact-leg --steps 40 --warmup 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$setting] %.1f frames/s %.3f ms' % (d['value'], d['ms_per_step']))"
  done
done
