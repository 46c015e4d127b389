"""Per-workgroup phase times of the 1080p upsampling layer (conv3x3_split_kernel<true>, 540 x 960 x 64 -> 1080 x 1920 x 64):
s_memrealtime stamps of one launch (start | first chunk staged | MFMAs done | end).  PYTHONPATH=. python tools/lab/ups_timeline.py"""
import ctypes
import numpy as np
import torch
from isosurfacesuperresolution_amd import ops

lib = ops._sr()
lib.isrDebugSetSplitStampBuffer.argtypes = [ctypes.c_void_p]
with torch.no_grad():
    for (h, w), packed in (((540, 960), True), ((540, 960), False), ((270, 480), False)):
        x = torch.rand(1, 64, h, w, device='cuda') - 0.5
        wt = (torch.rand(64, 64, 3, 3, device='cuda') - 0.5) * 0.1
        b = torch.rand(64, device='cuda')
        H, W = 2 * h, 2 * w
        nwg = ((H + 7) // 8) * ((W + 31) // 32)
        buf = torch.zeros(nwg * 4, dtype=torch.int64, device='cuda')
        run = (lambda: ops.conv3x3_split_packed(x, wt, b, act='relu', upsample2x=True)) if packed else \
              (lambda: ops.conv3x3_split(x, wt, b, act='relu', upsample2x=True))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        lib.isrDebugSetSplitStampBuffer(ctypes.c_void_p(buf.data_ptr()))
        run()
        torch.cuda.synchronize()
        lib.isrDebugSetSplitStampBuffer(None)
        st = buf.cpu().numpy().reshape(-1, 4).astype(np.float64) * 10.0     # ns
        t0 = st[:, 0].min()
        start, staged, mfma, end = (st[:, k] - t0 for k in range(4))
        pct = lambda a: "10%% %.1f / median %.1f / 90%% %.1f / mean %.2f" % (tuple(np.percentile(a, [10, 50, 90]) / 1e3) + (a.mean() / 1e3,))
        print("%dx%d -> %dx%d%s, %d workgroups, %.0f us per launch:" % (w, h, W, H, " packed-split out" if packed else "", nwg, e0.elapsed_time(e1) * 100))
        print("   first staging  ", pct(staged - start))
        print("   MFMA phase     ", pct(mfma - staged))
        print("   epilogue       ", pct(end - mfma))
        print("   life           ", pct(end - start), " -> kernel span %.1f us; sum of lives / (2 x 256 slots) = %.1f us" % (end.max() / 1e3, (end - start).sum() / 512e3))
