#!/bin/bash
# PMC passes of the ray-march kernel alone (GPU box): bash tools/lab/prof_raymarch.sh -> gpurun_out/r02_rm/<case>_<group>/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r02_rm${RM_TAG:-}
rm -rf $OUT; mkdir -p $OUT
for case in "ejecta256 480x270" "ejecta256 1920x1080" "cloud512 1920x1080"; do
  set -- $case
  tag=$1_$2
  python3 tools/raymarch_only.py $1 $2 12 > $OUT/$tag.time.log 2>&1; tail -1 $OUT/$tag.time.log
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/${tag}_g$i -o run -- python3 tools/raymarch_only.py $1 $2 6 > $OUT/${tag}_g$i.log 2>&1 || echo "group $i ($grp) failed"
    find $OUT/${tag}_g$i -name "*kernel_trace.csv" -delete 2>/dev/null
  done
  echo "$tag done"
done
du -sh $OUT
