"""bench.py with TWO ranks sharing the one GPU of the box (BENCH_SHARE_DEVICE=1, gloo as the collective backend on device tensors):
the self-launched ranks, the frame chunks of the default mode and the tiled mode's all-gathers with the next frame's render +
composite on a side stream -- against the same command with one rank.  (RCCL itself needs one GPU per rank: the driver's node.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, ranks, port):
    env = dict(os.environ, BENCH_SHARE_DEVICE="1", BENCH_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks)] + args, env=env, cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout.decode()[-2000:]
    return json.loads(lines[0])


def test_tiled_mode_two_ranks_on_one_gpu_composite_and_output_equal_one_rank():
    common = ["--mode", "tiled", "--tiled-n", "256", "--low", "480x270", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"]
    one = _bench(common, 1, 29811)
    two = _bench(common, 2, 29812)
    serial = _bench(common + ["--no-overlap"], 2, 29813)
    assert two["n_gpus"] == 2 and two["ranks_joined"] == 2 and one["n_gpus"] == 1
    # ranks that share a device never take the all-resident spin kernels (two processes' grids interleaved on the CUs would
    # wait for each other until the deadline) -- in EVERY mode, not only the default one
    for d in (one, two, serial):
        assert d["config"]["spin_kernel_forms"] == {"trunk_dataflow": False, "flow_fill_one": False}
    assert "side HIP stream" in two["config"]["overlap"] and serial["config"]["overlap"] == "none"
    # tiles walk the global ray: the two-tile composite IS the unsplit render; strips reproduce the whole frame
    assert one["hit_pixels"] == two["hit_pixels"] == serial["hit_pixels"] > 1000
    assert two["rgb_mean"] == serial["rgb_mean"]                                      # the prefetched sequence is the serial one, bit for bit
    assert abs(two["rgb_mean"] - one["rgb_mean"]) <= 1e-6 * max(1.0, abs(one["rgb_mean"]))


def test_default_mode_two_ranks_on_one_gpu():
    d = _bench(["--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-fast-mode", "--no-exact-leg"], 2, 29814)
    assert d["n_gpus"] == 2 and d["ranks_joined"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["frames_per_rank"] == 6
    assert d["config"]["spin_kernel_forms"] == {"trunk_dataflow": False, "flow_fill_one": False}
    ft = d["frame_time"]
    assert ft["gap_ms_per_frame"] == pytest.approx(d["ms_per_step"] - ft["main_stream_kernels_ms_per_frame"]) and ft["host_enqueue_ms_per_frame"] > 0


def test_train_mode_on_the_gpu_reports_clips_per_rank():
    """``config.clips_per_rank`` was overwritten by the per-kernel tally on the GPU path (the line a SCALE run emits)."""
    d = _bench(["--mode", "train", "--steps", "2", "--warmup", "1", "--train-batch", "4", "--train-frames", "3", "--no-cpu-baseline"], 1, 29815)
    assert d["config"]["clips_per_rank"] == 4 and d["unit"] == "clips/s" and d["value"] > 0
    assert d["roofline"]["kernels"] and all(v["frac"] > 0 for v in d["roofline"]["kernels"].values())
