"""Ablations of the forward conv (diagnostic bits: 1 = no staging of later chunks, 2 = no epilogue stores)."""
import sys, torch
sys.path.insert(0, '.')
from isosurfacesuperresolution_amd import ops
lib = ops._sr()
shapes = [(64, 64, 270, 480, False), (64, 64, 1080, 1920, False), (64, 64, 540, 960, True), (64, 64, 1080, 1920, True)]
with torch.no_grad():
    for cin, cout, h, w, ups in shapes:
        hin, win = (h // 2, w // 2) if ups else (h, w)
        x = torch.rand(1, cin, hin, win, device='cuda') - 0.5
        wt = (torch.rand(cout, cin, 3, 3, device='cuda') - 0.5) * 0.1
        b = torch.rand(cout, device='cuda')
        line = []
        for dbg in (0, 1, 2, 3):
            lib.isrDebugSetAblation(dbg)
            ops.profile_enable(True)
            for _ in range(12): ops.conv3x3(x, wt, b, act='relu', upsample2x=ups)
            torch.cuda.synchronize()
            rec = ops.profile_records()[2:]
            ops.profile_enable(False)
            ms = sum(r[2] for r in rec) / len(rec)
            fl = 2.0 * 9 * cin * cout * h * w
            line.append("dbg%d %.1f us (%.1f TF)" % (dbg, ms * 1e3, fl / ms / 1e9))
        lib.isrDebugSetAblation(0)
        print("%dx%d %d->%d%s: " % (w, h, cin, cout, " ups" if ups else "") + " | ".join(line), flush=True)
