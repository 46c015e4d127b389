"""The 1080p tail (packed-split input, as in the frame) per form of isrDebugSetTailFused: 0 two kernels with 54 partial planes,
2 the S form (18 planes of horizontally pre-added partials, the default), 3 = timing experiment with a third of form 0's planes
written and read (wrong output).  PYTHONPATH=. python tools/lab/bench_tail_forms.py"""
import ctypes
import torch
from isosurfacesuperresolution_amd import ops
from isosurfacesuperresolution_amd.pipeline import default_shading

lib = ops._sr()
lib.isrDebugSetTailFused.argtypes = [ctypes.c_int]
ops.RANGE_GUARD = False
g = torch.Generator().manual_seed(1)
h, w = 270, 480
f2 = torch.rand(1, 64, 2 * h, 2 * w, generator=g).cuda()
w4 = ((torch.rand(64, 64, 3, 3, generator=g) - 0.5) * 0.08).cuda()
w6 = ((torch.rand(64, 64, 3, 3, generator=g) - 0.5) * 0.08).cuda()
b6 = ((torch.rand(64, generator=g) - 0.5) * 0.1).cuda()
w8 = ((torch.rand(6, 64, 3, 3, generator=g) - 0.5) * 0.08).cuda()
b8 = ((torch.rand(6, generator=g) - 0.5) * 0.1).cuda()
x = torch.rand(1, 101, h, w, generator=g).cuda()
sh = default_shading("cuda", 30.0)


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
    f4p = ops.conv3x3_split_packed(f2, w4, None, act='relu', upsample2x=True)
    for rnd in range(3):
        for form in (0, 3, 2):
            lib.isrDebugSetTailFused(form)
            t = timed(lambda: ops.tail_conv_finish(f4p, w6, b6, w8, b8, x, sh))
            ops.profile_enable(True)
            for _ in range(6):
                ops.tail_conv_finish(f4p, w6, b6, w8, b8, x, sh)
            torch.cuda.synchronize()
            ks = [ms for n, _, ms in ops.profile_records() if n == "conv3x3_split_tail_kernel"]
            ops.profile_enable(False)
            print("round %d form %d: tail (conv + combine / finish) %.0f us, of which the convolution kernel %.0f us" % (rnd, form, t, sum(ks) / len(ks) * 1e3), flush=True)
lib.isrDebugSetTailFused(2)
