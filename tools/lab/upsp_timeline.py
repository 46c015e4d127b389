"""Per-workgroup phase times of the phase-decomposed upsampling layer (conv3x3_split_upsp_kernel): s_memrealtime stamps of one launch.
PYTHONPATH=. python tools/lab/upsp_timeline.py [h w]"""
import ctypes
import sys
import numpy as np
import torch
from isosurfacesuperresolution_amd import ops

lib = ops._sr()
lib.isrDebugSetSplitStampBuffer.argtypes = [ctypes.c_void_p]
sizes = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(540, 960), (270, 480)]
lib.isrDebugSetSplitAblation.argtypes = [ctypes.c_int]
dbg = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # 1 no MFMAs, 2 no patch requests, 4 no weight requests, 8 no epilogue, 16 no epilogue stores, 32 every wait is vmcnt(0)
lib.isrDebugSetSplitAblation(dbg)
print("ablation mask", dbg)
with torch.no_grad():
    for h, w in sizes:
        x = torch.rand(1, 64, h, w, device='cuda') - 0.5
        wt = (torch.rand(64, 64, 3, 3, device='cuda') - 0.5) * 0.1
        b = torch.rand(64, device='cuda')
        xp = ops.pack_split(x)
        nwg = ((h + 7) // 8) * ((w + 31) // 32)
        buf = torch.zeros(nwg * 10, dtype=torch.int64, device='cuda')
        run = lambda: ops.conv3x3_ups_phase(xp, wt, b, act='relu')
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        ops.profile_enable(True, small_kernels=True)
        for _ in range(10):
            run()
        torch.cuda.synchronize()
        rec = ops.profile_records()
        ops.profile_enable(False)
        main = np.mean([ms for n, _, ms in rec if n == "conv3x3_split_upsp_kernel"]) * 1e3
        frame = np.mean([ms for n, _, ms in rec if n == "ups_frame_kernel"]) * 1e3
        lib.isrDebugSetSplitStampBuffer(ctypes.c_void_p(buf.data_ptr()))
        run()
        torch.cuda.synchronize()
        lib.isrDebugSetSplitStampBuffer(None)
        st = buf.cpu().numpy().reshape(-1, 10).astype(np.float64) * 10.0     # ns
        t0 = st[:, 0].min()
        st -= t0
        pct = lambda a: "10%% %.1f / median %.1f / 90%% %.1f / mean %.2f us" % (tuple(np.percentile(a, [10, 50, 90]) / 1e3) + (a.mean() / 1e3,))
        print("%dx%d -> %dx%d, %d workgroups: body %.0f us, frame kernel %.0f us per launch" % (w, h, 2 * w, 2 * h, nwg, main, frame))
        print("   first operands  ", pct(st[:, 1] - st[:, 0]))
        for m in range(4):
            prev = st[:, 1] if m == 0 else st[:, 1 + 2 * m]
            print("   image %d MFMAs   " % m, pct(st[:, 2 + 2 * m] - prev), "  epilogue", pct(st[:, 3 + 2 * m] - st[:, 2 + 2 * m]))
        life = st[:, 9] - st[:, 0]
        print("   life            ", pct(life), " -> kernel span %.1f us; sum of lives / (2 x 256 slots) = %.1f us" % (st[:, 9].max() / 1e3, life.sum() / 512e3))
        print("   starts (us): ", np.round(np.percentile(st[:, 0], [0, 25, 50, 75, 100]) / 1e3, 1))
