"""The 1080p upsampling layer (540 x 960 x 64 -> 1080 x 1920 x 64, packed-split out) on the four-rows-per-wave kernel (csrc/sr_conv_ups4r.h,
ISR_UPS_FORM=7) against the three-per-CU default (form 3): ablations (isrDebugSetSplitAblation: results wrong, time only) and the
per-workgroup phase sums of one stamped launch.  Diagnostics build.  PYTHONPATH=. python tools/lab/ups4r_timeline.py"""
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


names = {0: "everything", 1: "no MFMAs", 2: "no interpolation", 8: "no epilogue", 16: "no epilogue stores", 3: "no MFMAs, no interpolation",
         9: "no MFMAs, no epilogue", 10: "no interpolation, no epilogue (loads, parks, barriers, MFMAs)", 11: "loads, parks and barriers only"}
with torch.no_grad():
    for rnd in range(2):
        for bits, name in names.items():
            row = []
            for form in (3, 7):
                lib.isrDebugSetSplitUpsForm(form)
                lib.isrDebugSetSplitAblation(int(bits))
                row.append(timed())
            print("%-66s form 3 %6.0f us   form 7 %6.0f us" % (name, row[0], row[1]), flush=True)
    lib.isrDebugSetSplitAblation(0)
    lib.isrDebugSetSplitUpsForm(7)
    nwg = ((1080 + 15) // 16) * (1920 // 32)
    buf = torch.zeros(nwg * 8, dtype=torch.int64, device='cuda')
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    lib.isrDebugSetSplitStampBuffer(ctypes.c_void_p(buf.data_ptr()))
    run()
    torch.cuda.synchronize()
    lib.isrDebugSetSplitStampBuffer(None)
    lib.isrDebugSetSplitUpsForm(3)
    st = buf.cpu().numpy().reshape(-1, 8).astype(np.float64) * 10.0     # ns
    t0 = st[:, 0].min()
    pct = lambda a: "10%% %.1f / median %.1f / 90%% %.1f / mean %.2f us" % (tuple(np.percentile(a, [10, 50, 90]) / 1e3) + (a.mean() / 1e3,))
    print("form 7, %d workgroups, stamped launch:" % nwg)
    print("   staging (4 k-steps: park, barrier, interpolate)   ", pct(st[:, 1]))
    print("   MFMA rows (12 rows: wait, barrier, DMA, 72 MFMAs) ", pct(st[:, 2]))
    print("   epilogue                                          ", pct(st[:, 4] - st[:, 3]))
    print("   life                                              ", pct(st[:, 4] - st[:, 0]), " -> kernel span %.1f us; sum of lives / 512 slots = %.1f us"
          % ((st[:, 4].max() - t0) / 1e3, (st[:, 4] - st[:, 0]).sum() / 512e3))
