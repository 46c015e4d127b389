"""fp16 fast-mode convolution vs the fp32 kernel, per layer shape of the 1080p frame (HIP events, 20 launches each)."""
import sys, torch
sys.path.insert(0, '.')
from isosurfacesuperresolution_amd import ops
shapes = [(64, 64, 270, 480), (64, 64, 540, 960), (64, 64, 1080, 1920), (101, 64, 270, 480)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.no_grad():
    for cin, cout, h, w in shapes:
        x = torch.rand(1, cin, h, w, device='cuda') - 0.5
        wt = (torch.rand(cout, cin, 3, 3, device='cuda') - 0.5) * 0.1
        b = torch.rand(cout, device='cuda')
        res = {}
        for name, fn in (("fp32", lambda t: ops.conv3x3(t, wt, b, act='relu')), ("f16", lambda t: ops.conv3x3_f16(t, wt, b, act='relu'))):
            y = fn(x); y = fn(x)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20): y = fn(x)
            e1.record(); torch.cuda.synchronize()
            res[name] = e0.elapsed_time(e1) / 20
        import ctypes, numpy as np
        lib = ops._sr()
        nwg = ((h + 7) // 8) * ((w + 31) // 32) * ((cout + 63) // 64)
        buf = torch.zeros(nwg * 4, dtype=torch.int64, device='cuda')
        lib.isrDebugSetF16StampBuffer(ctypes.c_void_p(buf.data_ptr()))
        ops.conv3x3_f16(x, wt, b, act='relu'); torch.cuda.synchronize()
        lib.isrDebugSetF16StampBuffer(None)
        st = buf.cpu().numpy().reshape(-1, 4).astype(np.float64); d = np.diff(st, axis=1)
        print("   cycles per workgroup: staging %.0f  MFMA loop %.0f  epilogue %.0f  life %.0f" % (
            np.median(d[:, 0]), np.median(d[:, 1]), np.median(d[:, 2]), np.median(st[:, 3] - st[:, 0])))
        if (h, w) == (1080, 1920):
            for bits, name in ((1, "no MFMAs"), (2, "no staging loads"), (4, "no stores"), (3, "stores only"), (5, "staging only"), (6, "MFMAs only")):
                lib.isrDebugSetF16Ablation(bits)
                ops.conv3x3_f16(x, wt, b, act='relu'); torch.cuda.synchronize()
                e0.record()
                for _ in range(10): ops.conv3x3_f16(x, wt, b, act='relu')
                e1.record(); torch.cuda.synchronize()
                print("   ablation %-18s %.3f ms" % (name, e0.elapsed_time(e1) / 10))
            lib.isrDebugSetF16Ablation(0)
        gb = (cin * h * w * 4 * 1.2 + cout * h * w * 4) / 1e9
        print("%3d->%d %4dx%-4d fp32 %.3f ms  f16 %.3f ms  (x%.1f; %.0f GB/s of activations, %.0f TFLOP/s)" % (
            cin, cout, w, h, res["fp32"], res["f16"], res["fp32"] / res["f16"], gb / res["f16"] * 1e3,
            2 * 9 * cin * cout * h * w / res["f16"] / 1e9))
