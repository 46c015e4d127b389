#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench.py run only (the quick pass of tools/prof_r03.sh)
# usage: bash tools/prof_stats.sh [tag] [extra bench args...]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-stats}; shift
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --no-cpu-baseline --no-fast-mode --no-exact-leg "$@" > $OUT/stats.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
p = glob.glob(sys.argv[1] + '/stats/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(p)))[:16]:
    print("%-80s calls %6s  avg %8.1f us  %5.1f%%" % (r['Name'][:80], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
find $OUT -name "*kernel_trace.csv" -size +8M -delete
tail -c 300 $OUT/stats.log
