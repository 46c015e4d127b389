#!/bin/bash
# SQ counters of the 1080p upsampling layer (conv3x3_split_kernel<true>, packed-split output), two passes.  bash tools/lab/pmc_ups.sh
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r03_pmc_ups
rm -rf $OUT; mkdir -p $OUT
for form in 0; do
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $OUT/a$form -o run -- python3 tools/lab/ups_one.py > $OUT/a$form.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/b$form -o run -- python3 tools/lab/ups_one.py > $OUT/b$form.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/r03_pmc_ups/*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(float); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            if 'conv3x3_split' in r['Kernel_Name']:
                agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
        print(d, {k: "%.3g" % (v / max(n[k], 1)) for k, v in agg.items()})
PY
find $OUT -name "*kernel_trace.csv" -size +4M -delete
