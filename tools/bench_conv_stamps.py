"""In-kernel s_memtime stamps of the two-workgroups-per-CU forward conv: prologue / main loop / epilogue per workgroup."""
import sys, ctypes, torch
import numpy as np
sys.path.insert(0, '.')
from isosurfacesuperresolution_amd import ops
lib = ops._sr()
shapes = [(64, 64, 270, 480, False), (64, 64, 1080, 1920, False), (64, 64, 540, 960, True), (64, 64, 1080, 1920, True)]
with torch.no_grad():
    for cin, cout, h, w, ups in shapes:
        hin, win = (h // 2, w // 2) if ups else (h, w)
        nwg = ((h + 15) // 16) * ((w + 31) // 32) * ((cout + 31) // 32)
        buf = torch.zeros(nwg * 4, dtype=torch.int64, device='cuda')
        lib.isrDebugSetStampBuffer(ctypes.c_void_p(buf.data_ptr()))
        lib.isrDebugSetAblation(8)
        x = torch.rand(1, cin, hin, win, device='cuda') - 0.5
        wt = (torch.rand(cout, cin, 3, 3, device='cuda') - 0.5) * 0.1
        b = torch.rand(cout, device='cuda')
        for _ in range(2): ops.conv3x3(x, wt, b, act='relu', upsample2x=ups)
        torch.cuda.synchronize()
        st = buf.cpu().numpy().reshape(-1, 4).astype(np.float64)
        d = np.diff(st, axis=1)
        nz = st[:, 0] > 0
        st = st[0::8]; d = d[0::8]   # one XCD (blockIdx % 8 == 0): the counters of different XCDs are not aligned
        span = st[:, 3].max() - st[:, 0].min()
        t0 = st[:, 0] - st[:, 0].min(); t3 = st[:, 3] - st[:, 0].min()
        print('   start skew pct [50,90,99,100]:', np.percentile(t0, [50, 90, 99, 100]).round(0), ' end pct [1,10,50,100]:', np.percentile(t3, [1, 10, 50, 100]).round(0))
        nmfma = 9 * ((cin + 7) // 8) * 4 * 4
        print("%dx%d%s: ticks/WG prologue %.0f main %.0f epilogue %.0f | life %.0f | kernel span %.0f ticks (100 MHz) | main ns per MFMA %.2f (26.7 ns = 64 clk at 2.4 GHz, x2 when shared)" % (
            w, h, " ups" if ups else "", np.median(d[:, 0]), np.median(d[:, 1]), np.median(d[:, 2]), np.median(st[:, 3] - st[:, 0]), span,
            np.median(d[:, 1]) * 10.0 / nmfma))
    lib.isrDebugSetAblation(0)
    lib.isrDebugSetStampBuffer(None)
