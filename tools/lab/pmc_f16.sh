#!/bin/bash
# HBM traffic (FETCH_SIZE, WRITE_SIZE in separate passes) of the fp16 fast-mode kernel per layer shape
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_f16_rd gpurun_out/pmc_f16_wr
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f16_rd -o rd -- python3 tools/lab/bench_conv_f16.py > gpurun_out/pmc_f16_rd.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f16_wr -o wr -- python3 tools/lab/bench_conv_f16.py > gpurun_out/pmc_f16_wr.log 2>&1
python3 - <<'PY'
import csv, glob, collections
def load(d, name):
    out = collections.defaultdict(list)
    for p in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(p)):
            if r['Counter_Name'] == name and 'conv3x3_f16' in r['Kernel_Name']:
                out[int(r['Grid_Size'])].append(float(r['Counter_Value']))
    return out
rd, wr = load('gpurun_out/pmc_f16_rd', 'FETCH_SIZE'), load('gpurun_out/pmc_f16_wr', 'WRITE_SIZE')
for g in sorted(rd):
    r = sorted(rd[g])[len(rd[g]) // 2] * 1024 * 2 / 1e6      # KiB units, x2 on gfx950 (MI355X_MICROARCH.md)
    w = sorted(wr.get(g, [0]))[len(wr.get(g, [0])) // 2] * 1024 / 1e6
    print("grid %8d (%d workgroups): HBM read %.1f MB  write %.1f MB per launch" % (g, g // 256, r, w))
PY
