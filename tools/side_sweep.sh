#!/bin/bash
# Frame rate of bench.py under different settings of the overlapped ray-march (GPU box): bash tools/side_sweep.sh
cd $GRAFT_REPO_ROOT
run() {
  tag=$1; shift
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-fast-mode "$@" > gpurun_out/sweep_$tag.json 2> gpurun_out/sweep_$tag.err || { echo "$tag failed"; return 1; }
  python - "$tag" <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/sweep_%s.json" % sys.argv[1]) if l.startswith("{")][-1])
print("%-28s %7.1f frames/s  %.3f ms" % (sys.argv[1], d["value"], d["ms_per_step"]))
PY
}
run side2_start --side-variant 2 &&
run side0_start --side-variant 0 &&
run side0_trunk --side-variant 0 --prefetch-at trunk &&
run side0_up1 --side-variant 0 --prefetch-at up1 &&
run side0_up2 --side-variant 0 --prefetch-at up2 &&
run side5_trunk --side-variant 5 --prefetch-at trunk &&
run no_overlap --no-overlap &&
run side0_start_again --side-variant 0
