#!/bin/bash
# per-launch durations of one training step (bench.py --mode train): which launches of a kernel family are the long ones
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/train_trace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o run -- python3 bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline > $OUT/run.log 2>&1
python3 - <<'PY'
import csv, glob, collections
p = glob.glob('gpurun_out/train_trace/t/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last step = the last len/steps launches; find by adam_flat_kernel boundaries
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('adam_flat_kernel')]
a, b = idx[-2] + 1, idx[-1] + 1
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp']); t1 = int(step[-1]['End_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step)
print("last step: %d launches, span %.3f ms, kernel time %.3f ms" % (len(step), (t1 - t0) / 1e6, busy / 1e6))
fam = collections.defaultdict(list)
for r in step:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    fam[n[:60]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for n, d in sorted(fam.items(), key=lambda kv: -sum(kv[1])):
    d2 = sorted(d)
    print("%-60s n %4d  sum %8.1f us  min %6.1f med %6.1f max %6.1f" % (n, len(d), sum(d), d2[0], d2[len(d2) // 2], d2[-1]))
for key in ('conv3x3_wgrad_split_kernel', 'conv3x3_split_stream_kernel', 'wgrad_absmax_kernel'):
    for n, d in fam.items():
        if n.startswith(key):
            print(key, ' '.join('%.0f' % v for v in d))
# gaps between consecutive launches
gaps = [(int(step[i + 1]['Start_Timestamp']) - int(step[i]['End_Timestamp'])) / 1e3 for i in range(len(step) - 1)]
print("gaps: sum %.1f us, max %.1f, >2us: %d" % (sum(gaps), max(gaps), sum(1 for g in gaps if g > 2)))
PY
rm -rf $OUT/t
