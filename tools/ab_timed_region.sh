#!/bin/bash
# Why are K = 20-30 timed frames ~4 % slower per frame than the 2 000-frame `sustained` run of the same pipeline?  Candidates: the dispatch-packet
# events on the convolution launches (BENCH_PROFILE_TIMED), the length of the window (clock ramp after the idle gap in front of it).
cd $GRAFT_REPO_ROOT
for rnd in 1 2; do
  for setting in "BENCH_PROFILE_TIMED=1 STEPS=30" "BENCH_PROFILE_TIMED=0 STEPS=30" "BENCH_PROFILE_TIMED=1 STEPS=300" "BENCH_PROFILE_TIMED=0 STEPS=300"; do
    steps=${setting##*STEPS=}
    out=$(env ${setting% STEPS*} python3 bench.py --no-cpu-baseline --no-fast-mode --no-exact-leg --steps $steps --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%.1f frames/s %.4f ms | sustained %.1f (windows min %.1f max %.1f)' % (d['value'], d['ms_per_step'], d['sustained']['value'], d['sustained']['window_frames_per_s']['min'], d['sustained']['window_frames_per_s']['max']))")
    echo "round $rnd [$setting] $out"
  done
done
