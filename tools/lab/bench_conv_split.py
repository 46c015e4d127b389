"""Split-operand convolution vs the exact fp32 kernel and the fp16 fast mode, per layer shape of the 1080p frame
(HIP events, 20 launches each; in-kernel stamps and ablations of the 1080p layer)."""
import ctypes, sys
import numpy as np
import torch
sys.path.insert(0, '.')
from isosurfacesuperresolution_amd import ops
shapes = [(64, 64, 270, 480, False), (101, 64, 270, 480, False), (64, 64, 270, 480, True), (64, 64, 540, 960, True), (64, 64, 1080, 1920, False)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
lib = ops._sr()
lib.isrDebugSetSplitStampBuffer.argtypes = [ctypes.c_void_p]
lib.isrDebugSetSplitAblation.argtypes = [ctypes.c_int]
lib.isrDebugSetSplitAlgo.argtypes = [ctypes.c_int]


def timed(fn, n=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    for cin, cout, h, w, ups in shapes:
        x = torch.rand(1, cin, h, w, device='cuda') - 0.5
        wt = (torch.rand(cout, cin, 3, 3, device='cuda') - 0.5) * 0.1
        b = torch.rand(cout, device='cuda')
        H, W = (2 * h, 2 * w) if ups else (h, w)
        res = {}
        ops.SPLIT_F16 = False
        res["fp32"] = timed(lambda: ops.conv3x3(x, wt, b, act='relu', upsample2x=ups))
        ops.SPLIT_F16 = True
        res["split"] = timed(lambda: ops.conv3x3_split(x, wt, b, act='relu', upsample2x=ups))
        if not ups:      # one workgroup per tile (no prefetch) vs the persistent streaming kernel, same process
            lib.isrDebugSetSplitAlgo(0)
            res["split_tile"] = timed(lambda: ops.conv3x3_split(x, wt, b, act='relu'))
            lib.isrDebugSetSplitAlgo(1)
            res["split_stream"] = timed(lambda: ops.conv3x3_split(x, wt, b, act='relu'))
            lib.isrDebugSetSplitAlgo(2)
            print("   plain layer: one workgroup per tile %.3f ms, persistent streaming %.3f ms, wide (512 threads, pipelined reads) %.3f ms" % (
                res["split_tile"], res["split_stream"], res["split"]))
        res["f16"] = timed(lambda: ops.conv3x3_f16(x, wt, b, act='relu', upsample2x=ups))
        nwg = ((H + 7) // 8) * ((W + 31) // 32) * ((cout + 63) // 64)
        buf = torch.zeros(nwg * 4, dtype=torch.int64, device='cuda')
        lib.isrDebugSetSplitStampBuffer(ctypes.c_void_p(buf.data_ptr()))
        ops.conv3x3_split(x, wt, b, act='relu', upsample2x=ups); torch.cuda.synchronize()
        lib.isrDebugSetSplitStampBuffer(None)
        st = buf.cpu().numpy().reshape(-1, 4).astype(np.float64); d = np.diff(st, axis=1)
        print("   cycles per workgroup: first staging %.0f  rest (MFMA + staging of later chunks) %.0f  epilogue %.0f  life %.0f; kernel span %.0f" % (
            np.median(d[:, 0]), np.median(d[:, 1]), np.median(d[:, 2]), np.median(st[:, 3] - st[:, 0]), st[:, 3].max() - st[:, 0].min()))
        if (H, W) == (1080, 1920) and not ups:
            for bits, name in ((1, "no MFMAs"), (2, "no staging loads"), (4, "no stores"), (3, "stores only"), (5, "staging only"), (6, "MFMAs only")):
                lib.isrDebugSetSplitAblation(bits)
                t = timed(lambda: ops.conv3x3_split(x, wt, b, act='relu'), 10)
                print("   ablation %-18s %.3f ms" % (name, t))
            lib.isrDebugSetSplitAblation(0)
        gb = (cin * h * w * 4 + cout * H * W * 4) / 1e9
        fl = 2 * 9 * cin * cout * H * W
        print("%3d->%d %4dx%-4d%s fp32 %.3f ms (%.0f TF)  split %.3f ms (x%.2f; %.0f TFLOP/s algorithmic, %.0f matrix; %.0f GB/s of activations)  f16 %.3f ms" % (
            cin, cout, W, H, " ups" if ups else "    ", res["fp32"], fl / res["fp32"] / 1e9, res["split"], res["fp32"] / res["split"],
            fl / res["split"] / 1e9, 3 * fl / res["split"] / 1e9, gb / res["split"] * 1e3, res["f16"]), flush=True)
