"""Split-operand weight gradient on its own (isrConv3x3WeightGradSegmentsSplit: absmax + scale + kernel + slab reduction), the layer
shapes of the B=16 / T=10 training step; ISR_WGRAD_FORM=1 selects the one-wave-per-SIMD kernel.
usage: python tools/lab/bench_wgrad.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isosurfacesuperresolution_amd import ops   # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
import ctypes
ablate = int(os.environ.get("ISR_WGRAD_ABLATE", "0"))        # 1 no MFMAs, 2 no split / park, 4 no fetch (timing only: results are wrong)
ops._sr().isrDebugSetAblation.argtypes = [ctypes.c_int]
ops._sr().isrDebugSetAblation(ablate)
if ablate:
    print("ablation", ablate)
for segs, n, cin, cout, h, w in ((10, 16, 64, 64, 32, 32), (10, 16, 64, 64, 64, 64), (10, 16, 64, 64, 128, 128), (10, 16, 64, 6, 128, 128)):
    xs = [torch.relu(torch.randn(n, cin, h, w, device="cuda")) for _ in range(segs)]
    gzs = [torch.randn(n, cout, h, w, device="cuda") * 1e-3 for _ in range(segs)]
    weight = torch.zeros(cout, cin, 3, 3, device="cuda")
    ops.TRAIN_SPLIT = True
    for _ in range(2):
        ops._weight_grad(xs, gzs, weight, True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        ops._weight_grad(xs, gzs, weight, True)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    flops = 2.0 * 9 * cin * cout * n * segs * h * w
    print("%2d x [%d, %d -> %d, %d x %d]: %8.1f us  %6.1f TFLOP/s algorithmic (x3 on the fp16 pipe)" % (segs, n, cin, cout, h, w, us, flops / us / 1e6), flush=True)
