#!/bin/bash
# Package power (rocm-smi) while a bare fp32-MFMA loop keeps every CU busy, next to the conv kernel's figure.
# usage (GPU box): bash tools/lab/power_soak.sh   -> gpurun_out/power_soak.log
set -e
mkdir -p gpurun_out
hipcc -O3 -w --offload-arch=gfx950 tools/lab/mfma_peak.hip -o gpurun_out/mfma_peak
log=gpurun_out/power_soak.log
: > $log
for v in 0 1 2 3; do
    echo "== variant $v (bit0: zeros instead of random data, bit1: operands re-read from LDS)" >> $log
    ./gpurun_out/mfma_peak $v 6 &
    pid=$!
    sleep 2.5
    for i in 1 2 3; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" >> $log; sleep 0.7; done
    wait $pid
done
echo "== idle" >> $log
sleep 2
rocm-smi --showpower 2>/dev/null | grep -E "Power" >> $log
./gpurun_out/mfma_peak >> $log
cat $log
