#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench.py run; copies the summary into profiles/ on the caller's side
# usage (GPU box): bash tools/prof_bench.sh  ->  gpurun_out/prof_bench/bench_kernel_stats.csv, gpurun_out/bench.json
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/prof_bench
python3 bench.py 2>/dev/null | tail -1 > gpurun_out/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o bench -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
python3 - <<'PY'
import csv, glob, json
p = glob.glob('gpurun_out/prof_bench/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(p)))[:12]:
    print("%-90s calls %6s  avg %8.1f us  %5.1f%%" % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
d = json.load(open('gpurun_out/bench.json'))
print(d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['f16_fast_mode']['value'])
PY
