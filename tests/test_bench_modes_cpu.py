"""bench.py's multi-rank modes rehearsed with two gloo ranks on CPU tensors (the RCCL runs themselves happen on the
driver's multi-GPU node): the JSON line of each mode, the collectives it times, and that a failed launch cannot hide --
``rccl_ranks`` is a sum of ones over an all-reduce."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from isosurfacesuperresolution_amd import volumes as V

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline", "rccl_ranks", "backend", "ranks_joined"}
ROOFLINE = {"kernel", "bound", "achieved", "peak", "unit", "frac", "traffic"}


def _check_line(d, world):
    """Every mode's line is self-sufficient: the contract's keys, a non-null roofline, a cpu_baseline on the one-rank
    run (rank 0 at N = 1 only, as the contract says), and an honest account of which backend joined how many ranks."""
    assert REQUIRED <= set(d)
    assert d["roofline"] is not None and ROOFLINE <= set(d["roofline"])
    assert d["roofline"]["achieved"] > 0 and d["roofline"]["peak"] > 0
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-9
    if world == 1:
        assert d["cpu_baseline"] is not None and {"value", "unit", "cores", "kind", "sample"} <= set(d["cpu_baseline"])
        assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] in ("port", "reference")
        assert d["backend"] is None and d["ranks_joined"] is None and d["rccl_ranks"] is None
    else:
        # these rehearsals run over gloo: the line must not claim that RCCL saw the ranks
        assert d["backend"] == "gloo" and d["ranks_joined"] == world and d["rccl_ranks"] is None


def _launch(script_args, port, tmp_path, nproc=2, launcher=True, expect_ok=True):
    """``launcher``: start the ranks with torch.distributed.run (what the driver does for N > 1); False = plain
    ``python bench.py --gpus N``, which must start its N ranks itself (bench.launch_ranks)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", BENCH_DEVICE="cpu", BENCH_DIST_BACKEND="gloo", OMP_NUM_THREADS="2",
               BENCH_CPU_THREADS="2")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % nproc,
           "--master-addr", "127.0.0.1", "--master-port", str(port)] if (nproc > 1 and launcher) else [sys.executable]
    out = subprocess.run(cmd + script_args,
                         env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
    if not expect_ok:
        return out
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly ONE JSON line, from rank 0"
    return json.loads(lines[0])


def test_train_mode_two_gloo_ranks(tmp_path):
    d = _launch([os.path.join(ROOT, "bench.py"), "--mode", "train", "--gpus", "2", "--steps", "2", "--warmup", "1",
                 "--train-batch", "2", "--train-frames", "2", "--train-crop", "16"], 29731, tmp_path)
    _check_line(d, 2)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "strong" and d["unit"] == "clips/s"
    assert d["value"] > 0 and abs(d["value"] - 2 * 2 / (d["ms_per_step"] * 2 / 1e3)) < 1e-6 * d["value"]
    assert d["allreduce"]["bytes"] == 911046 * 4 and d["allreduce"]["buckets"] == 1 and d["allreduce"]["us"] > 0
    assert d["config"]["clips_per_rank"] == 1 and np.isfinite(d["loss"])
    assert d["roofline"]["bound"] == "mfma" and d["roofline"]["flops_per_step"] > 0
    assert d["config"]["step"] == "eager"


def test_train_mode_starts_its_own_ranks_without_a_launcher(tmp_path):
    """``python bench.py --gpus 2 --mode train`` with no WORLD_SIZE in the environment: the process starts two ranks itself,
    relays rank 0's ONE line, and the collective really joined two ranks."""
    d = _launch([os.path.join(ROOT, "bench.py"), "--mode", "train", "--gpus", "2", "--steps", "1", "--warmup", "1",
                 "--train-batch", "2", "--train-frames", "2", "--train-crop", "16"], 0, tmp_path, launcher=False)
    _check_line(d, 2)
    assert d["n_gpus"] == 2 and d["ranks_joined"] == 2 and d["config"]["clips_per_rank"] == 1
    assert d["allreduce"]["bytes"] == 911046 * 4


def test_self_launched_ranks_that_fail_give_a_nonzero_exit_and_no_line(tmp_path):
    """A rank that dies must fail the whole run: batch 3 over 2 ranks trips every rank's divisibility assert."""
    out = _launch([os.path.join(ROOT, "bench.py"), "--mode", "train", "--gpus", "2", "--steps", "1", "--warmup", "0",
                   "--train-batch", "3", "--train-frames", "2", "--train-crop", "16"], 0, tmp_path, launcher=False, expect_ok=False)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "launcher: rank" in out.stderr


def test_train_mode_one_rank_line_is_self_sufficient(tmp_path):
    d = _launch([os.path.join(ROOT, "bench.py"), "--mode", "train", "--gpus", "1", "--steps", "1", "--warmup", "1",
                 "--train-batch", "1", "--train-frames", "2", "--train-crop", "16"], 0, tmp_path, nproc=1)
    _check_line(d, 1)
    assert d["allreduce"] is None and d["cpu_baseline"]["unit"] == "clips/s"


def test_train_mode_capture_failure_falls_back_to_the_eager_step():
    """If capturing the step (with its RCCL all-reduce) fails on the driver's node, the run goes on eagerly in the same
    process and says so in the line -- never a re-exec of a process that has touched the GPU."""
    sys.path.insert(0, ROOT)
    import bench

    class Trainer:
        zeroed = steps = 0
        def graphed(self, batch, **kw):
            raise RuntimeError("hipErrorStreamCaptureUnsupported: operation not permitted when stream is capturing\nmore")
        def zero_grad(self):
            self.zeroed += 1
        def step(self, batch, **kw):
            self.steps += 1
            return 1.5
    t = Trainer()
    synced = []
    step, kind = bench.make_train_step(t, ("b",), True, sync=lambda: synced.append(1))
    assert kind == "eager (capture failed: RuntimeError: hipErrorStreamCaptureUnsupported: operation not permitted when stream is capturing)"
    assert t.zeroed == 1 and synced == [1] and step() == 1.5 and t.steps == 1
    step2, kind2 = bench.make_train_step(t, ("b",), False)
    assert kind2 == "eager" and step2() == 1.5

    class Good(Trainer):
        def graphed(self, batch, **kw):
            return lambda b: 2.5
    step3, kind3 = bench.make_train_step(Good(), ("b",), True)
    assert kind3.startswith("one HIP graph") and step3() == 2.5


def test_tiled_mode_two_gloo_ranks(tmp_path, oracle):
    """The tiled mode with the oracle as each rank's local renderer (bench.py itself never imports oracle/ outside its
    cpu_baseline leg: the renderer is injected here): tile-wise generation with the max / bbox reductions over gloo, the
    G-buffer all-gather, the composite and the strip super-resolution with its second all-gather."""
    script = tmp_path / "tiled.py"
    script.write_text('''
import sys
sys.path.insert(0, %r)
import torch
import bench
from oracle import iso_oracle as O
from isosurfacesuperresolution_amd import volumes as V

class Local:
    def __init__(self, tile):
        self.vol = O.OracleVolume(tile["data"], tile=tile)
        self.last = V.quantize3(V.orbit_camera(-1))
    def render(self, tensor, origin):
        q = V.quantize3(origin)
        p = O.make_params(tensor.shape[1], tensor.shape[0], origin=q, fov=30.0, isovalue=0.34, last_origin=self.last)
        img, _ = O.render(self.vol, p, threads=2)
        tensor.copy_(torch.from_numpy(img))
        self.last = q

bench.main(sys.argv[1:], make_local_renderer=Local)
''' % ROOT)
    d = _launch([str(script), "--mode", "tiled", "--gpus", "2", "--steps", "2", "--warmup", "1", "--tiled-n", "64", "--low", "48x32"],
                29733, tmp_path)
    _check_line(d, 2)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["unit"] == "frames/s"
    assert set(d["phases_ms_max_over_ranks"]) == {"render", "allgather", "composite", "sr_strip_and_allgather"}
    assert all(v >= 0 for v in d["phases_ms_max_over_ranks"].values()) and d["value"] > 0
    assert d["tile"]["voxels"] == [64, 64, 40]                      # 32 owned + 8 halo along z, whole extent in x and y
    # the composite of the last timed frame is the unsplit render: same number of hit pixels as the oracle on the whole volume
    vol = V.ejecta(64)
    q, last = V.quantize3(V.orbit_camera(1)), V.quantize3(V.orbit_camera(0))
    ref, _ = oracle.render(oracle.OracleVolume(vol), oracle.make_params(48, 32, origin=q, fov=30.0, isovalue=0.34, last_origin=last), threads=2)
    assert d["hit_pixels"] == int(ref[..., 3].sum()) > 50
    assert 0.0 < d["rgb_mean"] < 1.0
    # the same two ranks started by bench.py itself (no torch.distributed.run): the same composite
    d2 = _launch([str(script), "--mode", "tiled", "--gpus", "2", "--steps", "1", "--warmup", "1", "--tiled-n", "64", "--low", "48x32"],
                 0, tmp_path, launcher=False)
    _check_line(d2, 2)
    assert d2["ranks_joined"] == 2 and d2["tile"]["voxels"] == [64, 64, 40] and d2["hit_pixels"] > 50
    # one rank: the same line with a cpu_baseline (oracle tile render + CPU network on a bounded sample)
    d1 = _launch([str(script), "--mode", "tiled", "--gpus", "1", "--steps", "1", "--warmup", "1", "--tiled-n", "64", "--low", "48x32"],
                 0, tmp_path, nproc=1)
    _check_line(d1, 1)
    assert d1["hit_pixels"] > 50 and d1["cpu_baseline"]["unit"] == "frames/s"


def test_kernel_tally_and_gap_accounting():
    """The default line's ``frame_time``: what part of a frame is NOT inside a main-stream kernel, and the host's enqueue time --
    an outlier run must explain itself from the one record (gpurun_out/r04_bench0.json had 0.53 ms of gap and no
    field that said so)."""
    sys.path.insert(0, ROOT)
    import bench
    K = 4
    records = []
    for _ in range(K):
        records += [("assemble_input_kernel", 0.0, 0.04), ("trunk_pack_input_kernel", 0.0, 0.016), ("trunk_dataflow_kernel", 206e9, 0.60),
                    ("flow_fill_one_kernel", 0.0, 0.045), ("conv3x3_split_ups3_kernel", 38.2e9, 0.14), ("conv3x3_split_ups3_kernel", 152.9e9, 0.46),
                    ("conv3x3_split_tail_kernel", 167e9, 0.47), ("tail_finish_kernel", 0.0, 0.05)]
    per = bench.tally_kernels(records)
    assert per["conv3x3_split_ups3_kernel"][2] == 2 * K and abs(per["conv3x3_split_ups3_kernel"][1] - K * 0.60e-3) < 1e-12
    g = bench.gap_accounting(2.30, per, K, [0.9, 1.0, 1.1, 2.9])
    main = 0.04 + 0.016 + 0.60 + 0.14 + 0.46 + 0.47 + 0.05
    assert abs(g["main_stream_kernels_ms_per_frame"] - main) < 1e-9
    assert abs(g["gap_ms_per_frame"] - (2.30 - main)) < 1e-9                       # the flow fill runs on the render stream: not in the sum
    assert abs(g["side_stream_kernels_ms_per_frame"] - 0.045) < 1e-9
    assert g["host_enqueue_ms_max"] == 2.9 and abs(g["host_enqueue_ms_per_frame"] - 1.475) < 1e-9 and g["host_enqueue_ms_median"] == 1.1
    # kernels without matrix work never become the roofline's dominant kernel
    dom = max(((n, v) for n, v in per.items() if v[0] > 0), key=lambda kv: kv[1][1])
    assert dom[0] == "conv3x3_split_ups3_kernel" or dom[0] == "trunk_dataflow_kernel"


def test_sustained_object_of_the_default_line():
    """The shape of the default line's `sustained` object (>= 2 000 more frames of the same pipeline after the timed
    region, whole-run frames/s, min / median / max over 100-frame windows, the guard words read at the end) from raw measurements."""
    sys.path.insert(0, ROOT)
    import bench
    window_ms = [175.0, 174.0, 180.0, 350.0] + [176.0] * 16                      # one window with an outlier in it
    d = bench.sustained_summary(2000, 3.6, window_ms)
    assert {"frames", "seconds", "value", "unit", "ms_per_step", "window_frames", "windows", "window_frames_per_s", "guards", "note"} <= set(d)
    assert d["frames"] == 2000 and d["unit"] == "frames/s" and d["window_frames"] == bench.SUSTAINED_WINDOW == 100 and d["windows"] == 20
    assert abs(d["value"] - 2000 / 3.6) < 1e-9 and abs(d["ms_per_step"] - 1.8) < 1e-9
    w = d["window_frames_per_s"]
    assert abs(w["min"] - 100e3 / 350.0) < 1e-9 and abs(w["max"] - 100e3 / 174.0) < 1e-9 and abs(w["median"] - 100e3 / 176.0) < 1e-9
    assert w["min"] <= w["median"] <= w["max"] and d["guards"] == "clean"
    assert bench.parse([]).sustained_frames >= 2000                              # on by default, in the driver's own command
    assert bench.parse(["--sustained-frames", "0"]).sustained_frames == 0


def test_a_rank_that_never_joins_ends_the_run_with_rank_and_phase_named(tmp_path):
    """the first real multi-rank run must fail loudly, not hang to the driver's limit.  World size 2, only rank 0
    started: the rendezvous is bounded (BENCH_DIST_TIMEOUT_S, default 120 s), the process says which rank and phase, exits 3."""
    import time
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29741", BENCH_DEVICE="cpu", BENCH_DIST_BACKEND="gloo",
               WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", BENCH_DIST_TIMEOUT_S="5", OMP_NUM_THREADS="2")
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "train", "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--train-batch", "2", "--train-frames", "2", "--train-crop", "16"],
                         env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=150, text=True)
    assert out.returncode == 3 and time.time() - t0 < 150
    assert "rank 0/2: phase 'init_process_group'" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_a_rank_that_stalls_in_the_first_collective_is_ended_by_the_deadline(tmp_path):
    """Both ranks pass the rendezvous; rank 1 then never returns from its first all-reduce (a stuck RCCL bootstrap is a C call
    that Python cannot interrupt: simulated by a sleeping all_reduce).  Rank 0's collective times out and names the phase; rank
    1 is ended by the watchdog thread of ITS phase; the self-launching parent returns non-zero and prints no line."""
    script = tmp_path / "stall.py"
    script.write_text('''
import os, sys, time
sys.path.insert(0, %r)
import torch.distributed as dist
import bench
if os.environ.get("RANK") == "1":
    dist.all_reduce = lambda *a, **k: time.sleep(600)
bench.main(sys.argv[1:])
''' % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", BENCH_DEVICE="cpu", BENCH_DIST_BACKEND="gloo", BENCH_DIST_TIMEOUT_S="5", OMP_NUM_THREADS="2")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, str(script), "--mode", "train", "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--train-batch", "2", "--train-frames", "2", "--train-crop", "16"],
                         env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=150, text=True)
    assert out.returncode != 0
    assert "phase 'first collective (all-reduce)'" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
