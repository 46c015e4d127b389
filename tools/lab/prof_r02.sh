#!/bin/bash
# Round-2 profile set of the default bench.py run (GPU box): rocprofv3 kernel stats, then the three PMC passes the
# HBM section of MI355X_MICROARCH.md prescribes (one counter group per run, no other trace domains).
# usage: bash tools/lab/prof_r02.sh   ->  gpurun_out/r02_prof/{stats,fetch,write,sq}/..., gpurun_out/r02_prof/*.log
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r02_prof
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --no-cpu-baseline --no-fast-mode > $OUT/stats.log 2>&1
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o run -- python3 bench.py --no-cpu-baseline --no-fast-mode --steps 5 --warmup 2 > $OUT/fetch.log 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o run -- python3 bench.py --no-cpu-baseline --no-fast-mode --steps 5 --warmup 2 > $OUT/write.log 2>&1
echo "write done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $OUT/sq -o run -- python3 bench.py --no-cpu-baseline --no-fast-mode --steps 5 --warmup 2 > $OUT/sq.log 2>&1
echo "sq done"
python3 - <<'PY'
import csv, glob
p = glob.glob('gpurun_out/r02_prof/stats/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(p)))[:14]:
    print("%-90s calls %6s  avg %8.1f us  %5.1f%%" % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
# keep the merged output small: the per-dispatch traces are large
find $OUT -name "*kernel_trace.csv" -size +8M -delete
du -sh $OUT
