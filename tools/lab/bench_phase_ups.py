"""Feasibility of the upsampling layers as four phase convolutions on the LOW-resolution tensor (conv3x3(U(x)) = for each output
phase (Y % 2, X % 2) a 3x3 convolution of x with combined weights): times four launches of the plain split kernel with packed-split
input (LDS-DMA staging, no vector staging work) and packed-split output at 540 x 960 against the 1080p upsampling launch it would
replace, and the same at 270 x 480.  PYTHONPATH=. python tools/lab/bench_phase_ups.py"""
import torch
from isosurfacesuperresolution_amd import ops

ops.RANGE_GUARD = False
wt = (torch.rand(64, 64, 3, 3, device='cuda') - 0.5) * 0.1
b = torch.rand(64, device='cuda')


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
    for h, w in ((540, 960), (270, 480)):
        x = torch.rand(1, 64, h, w, device='cuda') - 0.5
        xp = ops.conv3x3_split_packed(x, wt, b, act='relu')            # some packed-split tensor of that size
        for rnd in range(3):
            t_ups = timed(lambda: ops.conv3x3_split_packed(x, wt, b, act='relu', upsample2x=True))
            t_four = timed(lambda: [ops.conv3x3_split_from_packed(xp, wt, b, act='relu', packed_out=True) for _ in range(4)])
            print("%dx%d -> x2: upsampling launch %.0f us | four packed-in / packed-out plain launches %.0f us" % (w, h, t_ups, t_four))
