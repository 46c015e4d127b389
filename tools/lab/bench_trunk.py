"""The 480 x 270 trunk (preblock + 10 residual blocks) on its own: the dataflow launch (csrc/sr_conv_trunk.hip) against the per-layer
launches, and the dataflow launch with parts ablated (isrDebugSetTrunkAblation: results wrong, time only) to see what bounds it.
usage: PYTHONPATH=. python tools/lab/bench_trunk.py [H W]"""
import sys

import torch

from isosurfacesuperresolution_amd import ops

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (270, 480)
g = torch.Generator().manual_seed(0)
x = (torch.rand(1, 101, H, W, generator=g) - 0.3).cuda()
convs = [(((torch.rand(64, 101 if k == 0 else 64, 3, 3, generator=g) - 0.5) * 0.1).cuda(), ((torch.rand(64, generator=g) - 0.5) * 0.1).cuda())
         for k in range(21)]
lib = ops._sr()


def per_layer():
    f = ops.conv3x3(x, convs[0][0], convs[0][1], act='relu')
    for k in range(10):
        f = ops.residual_block(f, convs[2 * k + 1][0], convs[2 * k + 1][1], convs[2 * k + 2][0], convs[2 * k + 2][1])
    return f


def dataflow():
    return ops.trunk_dataflow(x, convs)


def timed(fn, n=30):
    for _ in range(5):
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
    a, b = per_layer(), dataflow()
    torch.cuda.synchronize()
    ops.trunk_check()
    print("bit-identical:", bool(torch.equal(a, b)))
    for rnd in range(3):
        print("round %d: per-layer %.0f us, dataflow %.0f us" % (rnd, timed(per_layer), timed(dataflow)))
    for guard in (False,):
        ops.RANGE_GUARD = guard
        print("range guard %s: per-layer %.0f us, dataflow %.0f us" % ("on" if guard else "off", timed(per_layer), timed(dataflow)))
        names = {64: "everything, on the DIAGNOSTICS build (the build every ablation below runs on: 24 registers spilled)", 1: "no MFMAs", 2: "no activation DMA", 4: "no stores", 8: "no waits", 16: "no weight DMA", 3: "no MFMAs, no activation DMA",
                 5: "no MFMAs, no stores", 9: "no MFMAs, no waits", 18: "no DMA at all", 22: "no DMA, no stores", 23: "no MFMAs, DMA, stores",
                 31: "barriers + bias only", 30: "MFMAs only", 62: "MFMAs only, operands read once",
                 126: "MFMAs only on operands read once, the taps' LDS reads issued and dropped", 32: "everything, operands read once",
                 96: "everything, MFMAs on operands read once, the taps' LDS reads issued and dropped"}
        for mask, name in names.items():
            lib.isrDebugSetTrunkAblation(mask)
            t = timed(dataflow)
            print("  ablation %2d (%s): %.0f us" % (mask, name, t))
        lib.isrDebugSetTrunkAblation(0)
    ops.RANGE_GUARD = True
    print("dataflow again: %.0f us" % timed(dataflow))
