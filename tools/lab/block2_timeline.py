"""Per-workgroup phase times of the fused small-image residual block (csrc/sr_conv_block2.h; 16 x 64 x 32 x 32, forward and backward):
s_memrealtime stamps of one launch in a chain of launches.  PYTHONPATH=. python tools/lab/block2_timeline.py"""
import ctypes
import numpy as np
import torch
from isosurfacesuperresolution_amd import ops

lib = ops._sr()
lib.isrDebugSetSplitStampBuffer.argtypes = [ctypes.c_void_p]
lib.isrDebugSetSplitAblation.argtypes = [ctypes.c_int]
import os
lib.isrDebugSetSplitAblation(int(os.environ.get("ISR_SPLIT_ABLATE", "0")))
n, h, w = 16, 32, 32
zero = os.environ.get("ISR_ZERO_DATA", "0") == "1"        # all-zero activations and weights: the matrix pipe at its lowest power
x = (torch.rand(n, 64, h, w, device="cuda") - 0.3) * (0.0 if zero else 1.0)
ws = [((torch.rand(64, 64, 3, 3, device="cuda") - 0.5) * 0.1) for _ in range(2)]
if zero:
    ws = [torch.full((64, 64, 3, 3), 1e-30, device="cuda") for _ in range(2)]
bs = [((torch.rand(64, device="cuda") - 0.5) * 0.2) for _ in range(2)]
t = torch.relu(torch.randn(n, 64, h, w, device="cuda"))
nwg = n * ((h + 1) // 2)
for name, fn in (("forward", lambda v: ops._block2(v, ws[0], bs[0], None, ws[1], bs[1], False)[1]),
                 ("backward", lambda v: ops._block2(v, ws[1], None, t, ws[0], None, True)[1])):
    buf = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
    v = x
    for _ in range(4):
        v = fn(v) * 0.5
    torch.cuda.synchronize()
    v = fn(x); v = fn(v)                                   # the stamped launch follows a dependent one, as in the step
    lib.isrDebugSetSplitStampBuffer(ctypes.c_void_p(buf.data_ptr()))
    v = fn(v)
    lib.isrDebugSetSplitStampBuffer(None)
    v = fn(v)
    torch.cuda.synchronize()
    st = buf.cpu().numpy().reshape(-1, 8).astype(np.float64) * 10.0 / 1e3      # us
    t0 = st[:, 0].min()
    names = ["entry (after the first workgroup's)", "x patch parked, first weights landed", "stage 1 MFMAs", "z epilogue", "stage 2 MFMAs", "y epilogue (stores issued)", "stores drained"]
    print("%s: %d workgroups" % (name, nwg))
    prev = None
    for k in range(7):
        a = st[:, k] - (t0 if k == 0 else prev)
        print("   %-40s median %5.2f  90%% %5.2f" % (names[k], np.median(a), np.percentile(a, 90)))
        prev = st[:, k]
    raw = buf.cpu().numpy().reshape(-1, 8).astype(np.float64)
    print("   shader clock over stage 1: %.2f GHz (median), %.0f cycles per MFMA" % (np.median(raw[:, 7] / ((raw[:, 2] - raw[:, 1]) * 10.0)), np.median(raw[:, 7]) / (216 * (8 if int(os.environ.get('ISR_SPLIT_ABLATE', '0')) & 32 else 1))))
    print("   life median %.2f us, span of the launch (first entry -> last drain) %.2f us" % (np.median(st[:, 6] - st[:, 0]), st[:, 6].max() - t0))
