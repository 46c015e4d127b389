#!/bin/bash
# GPU box: builds and runs the grid-barrier probe (every spin in it is bounded by a 50 ms timeout).
cd $GRAFT_REPO_ROOT/tools/probes && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -o /tmp/grid_barrier_probe grid_barrier_probe.hip && timeout -k 10 60 /tmp/grid_barrier_probe
