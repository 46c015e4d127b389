"""Where a 480x270 block layer's 33-38 us go: per-workgroup start / end times (s_memrealtime, one 100 MHz clock for the chip)
of one launch of the one-workgroup-per-tile split kernel.  python tools/lab/conv_layer_timeline.py"""
import ctypes, sys
import numpy as np
sys.path.insert(0, '.')
import torch
from isosurfacesuperresolution_amd import ops

lib = ops._sr()
lib.isrDebugSetSplitStampBuffer.argtypes = [ctypes.c_void_p]
lib.isrDebugSetSplitAlgo.argtypes = [ctypes.c_int]
with torch.no_grad():
    for h, w in ((270, 480), (540, 960)):
        x = torch.rand(1, 64, h, w, device='cuda') - 0.5
        wt = (torch.rand(64, 64, 3, 3, device='cuda') - 0.5) * 0.1
        b = torch.rand(64, device='cuda')
        res = torch.rand(1, 64, h, w, device='cuda')
        nwg = ((h + 7) // 8) * ((w + 31) // 32)
        buf = torch.zeros(nwg * 4, dtype=torch.int64, device='cuda')
        lib.isrDebugSetSplitAlgo(0)                      # stamps exist in the tile form
        for _ in range(3):
            ops.conv3x3_split(x, wt, b, act='none', residual=res)
        torch.cuda.synchronize()
        lib.isrDebugSetSplitStampBuffer(ctypes.c_void_p(buf.data_ptr()))
        ops.conv3x3_split(x, wt, b, act='none', residual=res)
        torch.cuda.synchronize()
        lib.isrDebugSetSplitStampBuffer(None)
        lib.isrDebugSetSplitAlgo(1)
        st = buf.cpu().numpy().reshape(-1, 4).astype(np.float64) * 10.0     # ns
        t0 = st[:, 0].min()
        start, staged, mfma, end = (st[:, k] - t0 for k in range(4))
        life = end - start
        pct = lambda a: "min %.1f / 10%% %.1f / median %.1f / 90%% %.1f / max %.1f" % tuple(np.percentile(a, [0, 10, 50, 90, 100]) / 1e3)
        print("%dx%d, %d workgroups (us after the first one started):" % (w, h, nwg))
        print("   start          ", pct(start))
        print("   first staging  ", pct(staged - start))
        print("   MFMA phase     ", pct(mfma - staged))
        print("   epilogue       ", pct(end - mfma))
        print("   life           ", pct(life))
        print("   end            ", pct(end), " -> kernel span %.1f us" % (end.max() / 1e3))
        # by XCD (workgroup id % 8)
        ids = np.arange(nwg)
        print("   median life by XCD:", " ".join("%.1f" % (np.median(life[ids % 8 == k]) / 1e3) for k in range(8)))
