#!/bin/bash
# Frame rate of bench.py under different settings of the overlapped ray-march (GPU box): bash tools/lab/side_sweep.sh
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
run default &&
run side2_capped --side-variant 2 &&
run side5 --side-variant 5 &&
run no_overlap --no-overlap &&
run default_again
