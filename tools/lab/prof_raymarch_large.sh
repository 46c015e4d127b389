#!/bin/bash
# Round 4: the ray-march kernel where the volume exceeds the caches -- cloud512 (537 MB dense) and ejecta1024
# (4.3 GB dense) at 1920x1080: kernel variants (0 flat gather = default, 1 LDS brick cache, 5 flat one-sample), cost-ordered tile
# dispatch (modes 1, 2), and the PMC passes of the default kernel (one counter group per run, as MI355X_MICROARCH.md prescribes).
#   bash tools/lab/prof_raymarch_large.sh  ->  gpurun_out/r04_rm_large/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r04_rm_large
rm -rf $OUT; mkdir -p $OUT
for vol in cloud512 ejecta1024; do
  for cfg in "0 0" "1 0" "5 0" "0 1" "0 2"; do
    set -- $cfg
    python3 tools/raymarch_only.py $vol 1920x1080 10 $1 $2 2>&1 | tail -1 | tee -a $OUT/times.txt
  done
done
for vol in cloud512 ejecta1024; do
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/${vol}_g$i -o run -- python3 tools/raymarch_only.py $vol 1920x1080 6 > $OUT/${vol}_g$i.log 2>&1 || echo "group $i ($grp) failed"
    find $OUT/${vol}_g$i -name "*kernel_trace.csv" -delete 2>/dev/null
    echo "$vol group $i done"
  done
done
python3 - <<'PY'
import csv, glob, collections
for vol in ("cloud512", "ejecta1024"):
    agg = collections.defaultdict(list)
    for f in glob.glob('gpurun_out/r04_rm_large/%s_g*/**/*counter_collection.csv' % vol, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'iso_render' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(vol, {k: "%.4g (n=%d)" % (sum(v) / len(v), len(v)) for k, v in sorted(agg.items())})
PY
du -sh $OUT
