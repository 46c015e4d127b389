#!/bin/bash
# Round-6 final verification on one box: the whole -m gpu suite, the default bench line, the two other modes, and the three modes with two
# ranks sharing the card over gloo (the rehearsal of the N > 1 code path that one GPU allows).  Output under gpurun_out/.
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/r06_gpu_suite_final.log 2>&1; echo "suite rc=$?"; tail -2 gpurun_out/r06_gpu_suite_final.log
python bench.py > gpurun_out/r06_bench_line_final.json 2> gpurun_out/r06_bench_line_final.err; echo "bench rc=$?"
python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/r06_mode_train_final.json 2> gpurun_out/r06_mode_train_final.err; echo "train rc=$?"
python bench.py --mode tiled --steps 10 --warmup 3 > gpurun_out/r06_mode_tiled_final.json 2> gpurun_out/r06_mode_tiled_final.err; echo "tiled rc=$?"
export BENCH_SHARE_DEVICE=1 BENCH_DIST_BACKEND=gloo
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --mode tiled --gpus 2 --steps 5 --warmup 2 --tiled-n 512 > gpurun_out/r06_mode_tiled_gloo2.json 2> gpurun_out/r06_mode_tiled_gloo2.err; echo "tiled gloo2 rc=$?"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29556 bench.py --mode train --gpus 2 --steps 5 --warmup 2 > gpurun_out/r06_mode_train_gloo2.json 2> gpurun_out/r06_mode_train_gloo2.err; echo "train gloo2 rc=$?"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29557 bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/r06_mode_infer_gloo2.json 2> gpurun_out/r06_mode_infer_gloo2.err; echo "infer gloo2 rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_*final.json') + glob.glob('gpurun_out/r06_*gloo2.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], d['value'], d['unit'], d['ms_per_step'], d['n_gpus'], (d.get('roofline') or {}).get('frac'))
    except Exception as e:
        print(f, 'unreadable', e)
PY
