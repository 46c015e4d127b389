"""Per-workgroup timeline of one launch of the fused residual block (resblock_split_kernel) on the chip's common 100 MHz clock,
and ablations of the same launch.  usage: python tools/lab/block_timeline.py [H W]"""
import ctypes
import sys

import numpy as np
import torch

from isosurfacesuperresolution_amd import ops

h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (270, 480)
g = torch.Generator().manual_seed(0)
x = (torch.rand(1, 64, h, w, generator=g) - 0.3).cuda()
w1 = ((torch.rand(64, 64, 3, 3, generator=g) - 0.5) * 0.1).cuda()
w2 = ((torch.rand(64, 64, 3, 3, generator=g) - 0.5) * 0.1).cuda()
b1 = torch.zeros(64, device="cuda")
lib = ops._sr()
lib.isrDebugSetBlockStampBuffer.argtypes = [ctypes.c_void_p]
lib.isrDebugSetBlockAblation.argtypes = [ctypes.c_int]


def timed(fn, n=20):
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
    fused = lambda: ops.residual_block_fused(x, w1, b1, w2, b1)
    two = lambda: ops.conv3x3_split(ops.conv3x3_split(x, w1, b1, act='relu'), w2, b1, residual=x)
    print("fused block %.1f us, two launches %.1f us (chained, no other work)" % (timed(fused), timed(two)))
    for bits, name in [(1, "no conv1 MFMAs"), (2, "no conv2 MFMAs"), (3, "no MFMAs"), (4, "no scratch stores"), (8, "no DMA"), (15, "staging + epilogues only")]:
        lib.isrDebugSetBlockAblation(bits)
        print("  %-28s %.1f us" % (name, timed(fused)))
    for bits, name in [(16, "younger half prio 1 in conv2"), (32, "younger half prio 1 always"), (64, "alternating prio per k-step")]:
        lib.isrDebugSetBlockAblation(bits)
        print("  %-28s %.1f us" % (name, timed(fused)))
    lib.isrDebugSetBlockAblation(0)
    stamps = torch.zeros(512 * 8, dtype=torch.int64, device="cuda")
    lib.isrDebugSetBlockStampBuffer(ctypes.c_void_p(stamps.data_ptr()))
    fused()
    torch.cuda.synchronize()
    lib.isrDebugSetBlockStampBuffer(None)
s = stamps.cpu().numpy().reshape(512, 8)[:, :6].astype(np.float64)
s = s[s[:, 0] > 0]
t0 = s[:, 0].min()
us = (s - t0) / 100.0
names = ["start", "first staging done", "conv1 MFMAs done", "t stored", "conv2 MFMAs done", "epilogue done"]
print("%d workgroups; median (min .. max) microseconds since the first workgroup's start" % len(us))
for k, n in enumerate(names):
    print("  %-20s %6.1f (%6.1f .. %6.1f)" % (n, np.median(us[:, k]), us[:, k].min(), us[:, k].max()))
d = np.diff(us, axis=1)
print("  phase medians: staging %.1f, conv1 %.1f, t epilogue %.1f, conv2 %.1f, epilogue %.1f" % tuple(np.median(d, axis=0)))
# where are the slow workgroups?  by XCD (blockIdx % 8), by position in the XCD's tile range, slowest ten
full = stamps.cpu().numpy().reshape(512, 8)[:, :6].astype(np.float64)
ids = np.nonzero(full[:, 0] > 0)[0]
dur = (full[ids, 2] - full[ids, 1]) / 100.0
life = (full[ids, 5] - full[ids, 0]) / 100.0
for xcd in range(8):
    m = (ids % 8) == xcd
    print("  XCD slot %d: conv1 median %.1f max %.1f | life median %.1f max %.1f" % (xcd, np.median(dur[m]), dur[m].max(), np.median(life[m]), life[m].max()))
order = np.argsort(-life)[:12]
print("  slowest workgroups (blockIdx, conv1 us, life us):", [(int(ids[k]), round(float(dur[k]), 1), round(float(life[k]), 1)) for k in order])
order = np.argsort(life)[:6]
print("  fastest workgroups:", [(int(ids[k]), round(float(dur[k]), 1), round(float(life[k]), 1)) for k in order])
pairs = {}
