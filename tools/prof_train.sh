#!/bin/bash
# rocprofv3 kernel stats of the training step (B clips, T frames): bash tools/prof_train.sh [B] [T]
set -e
B=${1:-16}; T=${2:-10}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/prof_train
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -o train -- python3 tools/bench_train.py $B $T 5 $3 > gpurun_out/prof_train.log 2>&1
python3 - <<'PY'
import csv, glob
p = glob.glob('gpurun_out/prof_train/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(p)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms (7 steps):", tot / 1e6)
for r in rows[:40]:
    print("%-110s calls %6s  total %9.2f ms  avg %8.1f us  %5.1f%%" % (r['Name'][:110], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
tail -2 gpurun_out/prof_train.log
