#!/bin/bash
# A/B of the training step (bench.py --mode train) between settings, interleaved, three rounds on one box.
# usage: bash tools/ab_train.sh "VAR=VAL ..." "VAR=VAL ..." ...   (e.g. ISR_SR_LIB=/root/repo/gpurun_out/lib_base/libisr_sr.so)
cd $GRAFT_REPO_ROOT
for rnd in 1 2 3; do
  for setting in "$@"; do
    out=$(env $setting python3 bench.py --mode train --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%.1f clips/s %.3f ms' % (d['value'], d['ms_per_step']))")
    echo "round $rnd [$setting] $out"
  done
done
