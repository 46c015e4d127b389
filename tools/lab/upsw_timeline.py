"""The 1080p upsampling layer on the one-stream persistent kernel (csrc/sr_conv_upsw.h, ISR_UPS_FORM=8): per-workgroup phase sums of one stamped
launch (diagnostics build) next to the launch time of forms 3 / 7 / 8.  PYTHONPATH=. python tools/lab/upsw_timeline.py"""
import ctypes
import numpy as np
import torch
from isosurfacesuperresolution_amd import ops

lib = ops._sr()
lib.isrDebugSetSplitStampBuffer.argtypes = [ctypes.c_void_p]
lib.isrDebugSetSplitUpsForm.argtypes = [ctypes.c_int]
ops.RANGE_GUARD = False
x = torch.rand(1, 64, 540, 960, device='cuda') - 0.5
wt = (torch.rand(64, 64, 3, 3, device='cuda') - 0.5) * 0.1
b = torch.rand(64, device='cuda')
run = lambda: ops.conv3x3_split_packed(x, wt, b, act='relu', upsample2x=True)


def timed(n=20):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


with torch.no_grad():
    for rnd in range(2):
        row = []
        for form in (3, 7, 8):
            lib.isrDebugSetSplitUpsForm(form)
            row.append(timed())
        print("1080p layer: form 3 %.0f us, form 7 %.0f us, form 8 %.0f us" % tuple(row), flush=True)
    lib.isrDebugSetSplitUpsForm(8)
    buf = torch.zeros(256 * 8, dtype=torch.int64, device='cuda')
    lib.isrDebugSetSplitStampBuffer(ctypes.c_void_p(buf.data_ptr()))
    run()
    torch.cuda.synchronize()
    lib.isrDebugSetSplitStampBuffer(None)
    lib.isrDebugSetSplitUpsForm(3)
    st = buf.cpu().numpy().reshape(-1, 8).astype(np.float64)
    us = st * 0.01
    steps = 4080 * 4 / 256.0
    pct = lambda a: "10%% %.1f / median %.1f / 90%% %.1f us" % tuple(np.percentile(a, [10, 50, 90]))
    print("form 8, 256 workgroups, ~%.1f steps each (%.1f of them stage a border tile):" % (steps, st[:, 7].mean()))
    print("   k-steps (216 MFMAs + slices)", pct(us[:, 1]), "  -> %.2f us per step" % (us[:, 1].sum() / (steps * 256)))
    print("   k-steps in shader cycles: %.0f per step -> %.2f GHz inside the k-steps (216 MFMAs = 6 912 cycles)" % (st[:, 2].sum() / (steps * 256), st[:, 2].sum() / (st[:, 1].sum() * 10.0)))
    print("   border staging              ", pct(us[:, 3]))
    print("   vmcnt(0) + barrier          ", pct(us[:, 4]), "  -> %.2f us per step" % (us[:, 4].sum() / (steps * 256)))
    print("   epilogues                   ", pct(us[:, 5]), "  -> %.2f us per tile" % (us[:, 5].sum() / 4080))
    print("   life                        ", pct(us[:, 6] - us[:, 0]))
