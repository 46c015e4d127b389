"""The 20 block layers of the 480 x 270 trunk as a chain of dependent launches, per form of the plain split kernel
(isrDebugSetSplitAlgo: 0 one workgroup per tile, 1 persistent streaming (default), 2 wide 512-thread workgroups with
hand-pipelined fragment reads) and as ten fused blocks.   usage: PYTHONPATH=. python tools/lab/bench_trunk_algos.py"""
import ctypes
import torch
from isosurfacesuperresolution_amd import ops

lib = ops._sr()
lib.isrDebugSetSplitAlgo.argtypes = [ctypes.c_int]
g = torch.Generator().manual_seed(0)
x = (torch.rand(1, 64, 270, 480, generator=g) - 0.3).cuda()
ws = [((torch.rand(64, 64, 3, 3, generator=g) - 0.5) * 0.05).cuda() for _ in range(20)]
b = torch.zeros(64, device="cuda")


def chain():
    f = x
    for k in range(10):
        t = ops.conv3x3_split(f, ws[2 * k], b, act='relu')
        f = ops.conv3x3_split(t, ws[2 * k + 1], b, residual=f)
    return f


def packed():
    f = x
    for k in range(10):
        t = ops.conv3x3_split_packed(f, ws[2 * k], b, act='relu')
        f = ops.conv3x3_split_from_packed(t, ws[2 * k + 1], b, residual=f)
    return f


def fused():
    f = x
    for k in range(10):
        f = ops.residual_block_fused(f, ws[2 * k], b, ws[2 * k + 1], b)
    return f


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


with torch.no_grad():
    # clocks ramp over the first tens of milliseconds of load and differ between boxes: warm up long, then measure the forms
    # interleaved, three rounds, and report every round
    lib.isrDebugSetSplitAlgo(1)
    ref = chain()
    timed(chain, 200)
    forms = [(1, "default (per tile when one round)", chain), (3, "persistent streaming", chain), (0, "one workgroup per tile", chain),
             (2, "wide, pipelined reads", chain), (1, "fused blocks", fused), (1, "per tile, packed intermediate", packed)]
    results = {name: [] for _, name, _ in forms}
    for rnd in range(3):
        for algo, name, fn in forms:
            lib.isrDebugSetSplitAlgo(algo)
            same = torch.equal(fn(), ref)
            results[name].append(timed(fn, 30) / 20)
            assert same, name
    lib.isrDebugSetSplitAlgo(1)
    for _, name, _ in forms:
        print("%-36s %s us per layer" % (name, "  ".join("%6.1f" % v for v in results[name])))
