"""Times the fused conv3x3 kernel alone at the three shapes of the 1080p frame."""
import sys, ctypes, torch
sys.path.insert(0, '.')
from isosurfacesuperresolution_amd import ops
lib = ops._sr()
shapes = [(64, 64, 270, 480, False), (64, 64, 1080, 1920, False), (64, 64, 540, 960, True), (64, 6, 1080, 1920, False), (101, 64, 270, 480, False)]
def run(dbg):
    lib.isrDebugSetAblation(dbg)
    out = []
    for cin, cout, h, w, ups in shapes:
        hin, win = (h // 2, w // 2) if ups else (h, w)
        x = torch.rand(1, cin, hin, win, device='cuda') - 0.5
        wt = (torch.rand(cout, cin, 3, 3, device='cuda') - 0.5) * 0.1
        b = torch.rand(cout, device='cuda')
        for _ in range(3): ops.conv3x3(x, wt, b, act='relu', upsample2x=ups)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n): ops.conv3x3(x, wt, b, act='relu', upsample2x=ups)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        fl = 2.0 * 9 * cin * cout * h * w
        out.append("%dx%d %d->%d%s: %.1f us %.1f TF" % (w, h, cin, cout, " ups" if ups else "", ms * 1e3, fl / ms / 1e9))
    print("dbg=%d | " % dbg + " | ".join(out))
with torch.no_grad():
    for dbg in [0, 1, 2, 3]:
        run(dbg)

# in-kernel stamps (diagnostic): prologue / main loop / epilogue shares per workgroup
import numpy as np
with torch.no_grad():
  for dbgbits in (8, 9, 10, 11):
    for cin, cout, h, w, ups in shapes[:3]:
        hin, win = (h // 2, w // 2) if ups else (h, w)
        nwg = ((h + 15) // 16) * ((w + 31) // 32)
        buf = torch.zeros(nwg * 4, dtype=torch.int64, device='cuda')
        lib.isrDebugSetStampBuffer(ctypes.c_void_p(buf.data_ptr()))
        lib.isrDebugSetAblation(dbgbits)
        x = torch.rand(1, cin, hin, win, device='cuda') - 0.5
        wt = (torch.rand(cout, cin, 3, 3, device='cuda') - 0.5) * 0.1
        b = torch.rand(cout, device='cuda')
        for _ in range(2): ops.conv3x3(x, wt, b, act='relu', upsample2x=ups)
        torch.cuda.synchronize()
        st = buf.cpu().numpy().reshape(-1, 4).astype(np.float64)
        d = np.diff(st, axis=1)
        span = (st[:, 3].max() - st[:, 0].min())
        print("dbg %d " % dbgbits + "%dx%d%s: cycles/WG prologue %.0f  main %.0f  epilogue %.0f  | total/WG %.0f  kernel span %.0f ticks; main per MFMA %.1f" % (
            w, h, " ups" if ups else "", np.median(d[:, 0]), np.median(d[:, 1]), np.median(d[:, 2]), np.median(st[:, 3] - st[:, 0]), span,
            np.median(d[:, 1]) / (4 * 9 * 64)))
    lib.isrDebugSetAblation(0)
    lib.isrDebugSetStampBuffer(None)
