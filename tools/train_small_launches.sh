#!/bin/bash
# which PyTorch elementwise launches are left in the graphed training step: (kernel, grid x block) counts of the last step
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
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('adam_flat_kernel')]
step = rows[idx[-2] + 1:idx[-1] + 1]
print(list(rows[0].keys()))
c = collections.Counter()
seq = []
for i, r in enumerate(step):
    n = r['Kernel_Name']
    if n.startswith('void at::') or n.startswith('at::') or 'rocclr' in n:
        key = (n.replace('void at::native::', '')[:70], r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')))
        c[key] += 1
        prev = step[i - 1]['Kernel_Name'].replace('(anonymous namespace)::', '')[:40] if i else ''
        nxt = step[i + 1]['Kernel_Name'].replace('(anonymous namespace)::', '')[:40] if i + 1 < len(step) else ''
        seq.append((i, key[0][:40], key[1], prev, nxt))
for k, v in c.most_common():
    print(v, k)
for s in seq[:60]:
    print(s)
PY
rm -rf $OUT/t
