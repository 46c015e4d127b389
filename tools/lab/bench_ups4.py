"""The role-split upsampling kernel (csrc/sr_conv_ups4.h) on the frame's 1080p layer with parts switched off (isrDebugSetSplitAblation:
1 no MFMAs, 2 no activation staging (loads, copy, interpolation), 4 no epilogue stores, 8 no weight DMA), next to the three-per-CU form.
PYTHONPATH=. python tools/lab/bench_ups4.py"""
import ctypes
import torch
from isosurfacesuperresolution_amd import ops

lib = ops._sr()
lib.isrDebugSetSplitUpsForm.argtypes = [ctypes.c_int]
lib.isrDebugSetSplitAblation.argtypes = [ctypes.c_int]
ops.RANGE_GUARD = False
wt = (torch.rand(64, 64, 3, 3, device='cuda') - 0.5) * 0.1
b = torch.rand(64, device='cuda')
x2 = torch.rand(1, 64, 540, 960, device='cuda') - 0.5
x1 = torch.rand(1, 64, 270, 480, device='cuda') - 0.5


def timed(fn, n=12):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


with torch.no_grad():
    for rnd in range(2):
        for form, bits in ((3, 0), (4, 0), (4, 1 << 8), (4, 2 << 8), (4, 3 << 8), (4, 4 << 8), (4, 6 << 8), (4, 4), (4, 2 | 8), (4, 2 | 4 | 8), (4, 2 | 8 | (2 << 8)), (3, 0)):
            lib.isrDebugSetSplitUpsForm(form)
            lib.isrDebugSetSplitAblation(bits)
            t2 = timed(lambda: ops.conv3x3_split_packed(x2, wt, b, act='relu', upsample2x=True))
            t1 = timed(lambda: ops.conv3x3_split(x1, wt, b, act='relu', upsample2x=True))
            print("round %d form %d ablation %#6x: 1080p layer %.0f us, 540p layer %.0f us" % (rnd, form, bits, t2, t1), flush=True)
lib.isrDebugSetSplitAblation(0)
lib.isrDebugSetSplitUpsForm(3)
