#!/bin/bash
cd $GRAFT_REPO_ROOT
python3 tools/lab/split_soak.py
for a in 1 2; do
  for bits in 0 6; do
    python3 tools/lab/split_soak.py $a 5 $bits &
    pid=$!
    sleep 3.2
    for i in 1 2; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.5; done
    wait $pid
  done
done
