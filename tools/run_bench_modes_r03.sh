#!/bin/bash
# Round 3: the bench lines of the other modes (training step, tiled config #5) at N = 1 and, as a rehearsal of the N > 1 paths, two
# gloo ranks sharing the one GPU.  bash tools/run_bench_modes_r03.sh
cd $GRAFT_REPO_ROOT; set -x
python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/r03_mode_train_n1.json 2> gpurun_out/r03_mode_train_n1.err && tail -c 1500 gpurun_out/r03_mode_train_n1.json && echo
python bench.py --mode tiled --steps 10 --warmup 3 > gpurun_out/r03_mode_tiled_n1.json 2> gpurun_out/r03_mode_tiled_n1.err && tail -c 1200 gpurun_out/r03_mode_tiled_n1.json && echo
BENCH_SHARE_DEVICE=1 BENCH_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --mode tiled --gpus 2 --steps 5 --warmup 2 --tiled-n 512 > gpurun_out/r03_mode_tiled_gloo2.json 2> gpurun_out/r03_mode_tiled_gloo2.err && tail -c 1200 gpurun_out/r03_mode_tiled_gloo2.json && echo
BENCH_SHARE_DEVICE=1 BENCH_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29556 bench.py --mode train --gpus 2 --steps 5 --warmup 2 > gpurun_out/r03_mode_train_gloo2.json 2> gpurun_out/r03_mode_train_gloo2.err && tail -c 1500 gpurun_out/r03_mode_train_gloo2.json && echo
BENCH_SHARE_DEVICE=1 BENCH_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29557 bench.py --gpus 2 --steps 10 --warmup 3 2> gpurun_out/r03_mode_infer_gloo2.err | grep "^{" > gpurun_out/r03_mode_infer_gloo2.json && python -c "
import json; d=json.load(open('gpurun_out/r03_mode_infer_gloo2.json')); print(d['value'], d['n_gpus'], d['rccl_ranks'])"
