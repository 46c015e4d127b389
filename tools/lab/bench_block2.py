"""The fused small-image residual block (csrc/sr_conv_block2.h) against two conv3x3_split_rows2_kernel launches, as a HIP graph of 40
block passes (the training trunk's shape: 16 x 64 x 32 x 32).  ISR_SPLIT_ABLATE: 1 skip the MFMAs, 4 skip the stores (timing only).
usage: python tools/lab/bench_block2.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isosurfacesuperresolution_amd import ops   # noqa: E402

lib = ops._sr()
lib.isrDebugSetSplitAblation.argtypes = [ctypes.c_int]
ablate = int(os.environ.get("ISR_SPLIT_ABLATE", "0"))
n, h, w = 16, 32, 32
x = (torch.rand(n, 64, h, w, device="cuda") - 0.3)
ws = [((torch.rand(64, 64, 3, 3, device="cuda") - 0.5) * 0.1) for _ in range(2)]
bs = [((torch.rand(64, device="cuda") - 0.5) * 0.2) for _ in range(2)]
t = torch.relu(torch.randn(n, 64, h, w, device="cuda"))


def fused_fwd():
    return ops._block2(x, ws[0], bs[0], None, ws[1], bs[1], False)[1]


def fused_bwd():
    return ops._block2(x, ws[1], None, t, ws[0], None, True)[1]


def two_fwd():
    a = ops._train_conv(x, ws[0], False, bs[0], None, 'relu')
    return ops._train_conv(a, ws[1], False, bs[1], x, 'none')


def two_bwd():
    a = ops._train_conv(x, ws[1], True, None, t, 'gate')
    return ops._train_conv(a, ws[0], True, None, x, 'none')


for name, fn in (("two launches, forward", two_fwd), ("fused, forward", fused_fwd), ("two launches, backward", two_bwd), ("fused, backward", fused_bwd)):
    lib.isrDebugSetSplitAblation(0)
    fn()
    torch.cuda.synchronize()
    lib.isrDebugSetSplitAblation(ablate)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        with ops.graph_capture(g, stream=s):
            for _ in range(40):
                y = fn()
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(s)
        for _ in range(5):
            g.replay()
        e1.record(s)
        torch.cuda.synchronize()
    lib.isrDebugSetSplitAblation(0)
    print("%-26s %6.2f us per block pass%s" % (name, e0.elapsed_time(e1) * 1e3 / 200, "  (ablation %d)" % ablate if ablate else ""), flush=True)
