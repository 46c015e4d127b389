#!/bin/bash
# Stand-alone ray-march kernel, two builds of libGPURendererDirect.so interleaved on one box (three rounds, two workloads).
# usage: bash tools/ab_raymarch.sh /root/repo/tools/lib_head/libGPURendererDirect_head.so ""
cd $GRAFT_REPO_ROOT
for rnd in 1 2 3; do
  for lib in "$@"; do
    for wl in "ejecta256 480x270 40" "cloud512 1920x1080 12"; do
      echo "round $rnd [${lib:-working tree}] $(ISR_RENDERER_LIB=$lib PYTHONPATH=. python3 tools/raymarch_only.py $wl 2>/dev/null)"
    done
  done
done
