#!/bin/bash
# Round-4 profile set of the default bench.py run (GPU box): rocprofv3 kernel stats, then the three PMC passes the
# HBM section of MI355X_MICROARCH.md prescribes (one counter group per run, no other trace domains).
# usage: bash tools/lab/prof_r03.sh   ->  gpurun_out/r04_prof/{stats,fetch,write,sq}/..., gpurun_out/r04_prof/*.log
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r04_prof
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --no-cpu-baseline --no-fast-mode --no-exact-leg > $OUT/stats.log 2>&1
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o run -- python3 bench.py --no-cpu-baseline --no-fast-mode --no-exact-leg --steps 5 --warmup 2 > $OUT/fetch.log 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o run -- python3 bench.py --no-cpu-baseline --no-fast-mode --no-exact-leg --steps 5 --warmup 2 > $OUT/write.log 2>&1
echo "write done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $OUT/sq -o run -- python3 bench.py --no-cpu-baseline --no-fast-mode --no-exact-leg --steps 5 --warmup 2 > $OUT/sq.log 2>&1
echo "sq done"
# effective shader clock per dispatch: GRBM_GUI_ACTIVE / 8 / duration (MI355X_MICROARCH.md, DVFS give-back; the sum over the 8 XCDs is reported)
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/clk -o run -- python3 bench.py --no-cpu-baseline --no-fast-mode --no-exact-leg --steps 5 --warmup 2 > $OUT/clk.log 2>&1
echo "clk done"
# the exact fp32 MFMA kernels (bench.py --exact): per-kernel evidence for the line's `exact_f32` leg
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/exact -o bench -- python3 bench.py --exact --no-cpu-baseline --no-fast-mode > $OUT/exact.log 2>&1
echo "exact done"
python3 - <<'PY'
import csv, glob, collections
# clock: join the counter rows with the kernel trace of the same run by dispatch id
for d in glob.glob('gpurun_out/r04_prof/clk/**/', recursive=True):
    cc = glob.glob(d + '*counter_collection.csv'); kt = glob.glob(d + '*kernel_trace.csv')
    if not cc or not kt:
        continue
    dur = {r['Dispatch_Id']: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in csv.DictReader(open(kt[0]))}
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(cc[0])):
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE' and r['Dispatch_Id'] in dur and dur[r['Dispatch_Id']] > 0:
            agg[r['Kernel_Name'][:70]].append(float(r['Counter_Value']) / 8.0 / dur[r['Dispatch_Id']])      # cycles per ns = GHz
    for k, v in sorted(agg.items(), key=lambda kv: -len(kv[1]))[:12]:
        print("clock %-72s %5.2f GHz (n=%d)" % (k, sum(v) / len(v), len(v)))
p = glob.glob('gpurun_out/r04_prof/exact/**/*kernel_stats.csv', recursive=True)
if p:
    for r in list(csv.DictReader(open(p[0])))[:8]:
        print("exact %-90s calls %6s  avg %8.1f us  %5.1f%%" % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
python3 - <<'PY'
import csv, glob
p = glob.glob('gpurun_out/r04_prof/stats/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(p)))[:14]:
    print("%-90s calls %6s  avg %8.1f us  %5.1f%%" % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
# keep the merged output small: the per-dispatch traces are large
find $OUT -name "*kernel_trace.csv" -size +8M -delete
du -sh $OUT
