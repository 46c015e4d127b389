"""EXPERIMENT: L plain 64 -> 64 conv + ReLU layers of a 480 x 270 image as ONE persistent dataflow launch (csrc/sr_conv_chain.hip)
against L dependent launches of the per-tile split kernel.  Outputs must be bit-identical.
usage: PYTHONPATH=. python tools/bench_chain.py [layers]"""
import ctypes
import sys

import torch

from isosurfacesuperresolution_amd import ops

L = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lib = ops._sr()
vp, ci, ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
lib.isrDebugConvChain.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ll, vp, ci, vp]
lib.isrDebugConvChain.restype = ci
g = torch.Generator().manual_seed(0)
H, W = 270, 480
x = (torch.rand(1, 64, H, W, generator=g) - 0.3).cuda()
ws = [((torch.rand(64, 64, 3, 3, generator=g) - 0.5) * 0.12).cuda() for _ in range(L)]
bs = [((torch.rand(64, generator=g) - 0.5) * 0.1).cuda() for _ in range(L)]
wqs = [ops._prepare_split(w) for w in ws]
bufA, bufB = torch.empty_like(x), torch.empty_like(x)
tiles = ((H + 7) // 8) * ((W + 31) // 32)
work = torch.zeros(tiles + 1, dtype=torch.int32, device="cuda")
pw = (ctypes.c_void_p * L)(*[t.data_ptr() for t in wqs])
pb = (ctypes.c_void_p * L)(*[t.data_ptr() for t in bs])


def launches():
    f = x
    for k in range(L):
        f = ops.conv3x3_split(f, ws[k], bs[k], act='relu')
    return f


def chain(delay=0):
    rc = lib.isrDebugConvChain(x.data_ptr(), bufA.data_ptr(), bufB.data_ptr(), pw, pb, L, H, W, H * W, work.data_ptr(), delay,
                               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
    return bufA if (L - 1) % 2 == 0 else bufB


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
    ref = launches()
    out = chain().clone()
    torch.cuda.synchronize()
    err = int(work[tiles].item())
    print("error word %d; bit-identical to %d launches: %s (max diff %.3g)" % (err, L, torch.equal(out, ref), (out - ref).abs().max().item()))
    timed(launches, 100)
    for rnd in range(3):
        t_l = timed(launches)
        res = [("launches", t_l)]
        for delay in (0, 800, 1500, 2500):
            t = timed(lambda: chain(delay))
            ok = torch.equal(chain(delay), ref) and int(work[tiles].item()) == 0
            res.append(("chain delay %d%s" % (delay, "" if ok else " WRONG"), t))
        print("  ".join("%s: %.1f us/layer" % (n, t / L) for n, t in res))
