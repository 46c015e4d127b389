"""Same work, different plane sizes: 1 x 1080p vs 16 x (480x270) vs 4 x (960x540) -- separates plane-stride (page
locality) effects from everything else in the forward conv."""
import sys, torch
sys.path.insert(0, '.')
from isosurfacesuperresolution_amd import ops
lib = ops._sr()
with torch.no_grad():
    for n, h, w in [(1, 1080, 1920), (4, 540, 960), (16, 270, 480), (64, 135, 240), (1, 270, 480)]:
        x = torch.rand(n, 64, h, w, device='cuda') - 0.5
        wt = (torch.rand(64, 64, 3, 3, device='cuda') - 0.5) * 0.1
        b = torch.rand(64, device='cuda')
        line = []
        for dbg in (0, 16, 2, 18):
            lib.isrDebugSetAblation(dbg)
            ops.profile_enable(True)
            for _ in range(8): ops.conv3x3(x, wt, b, act='relu')
            torch.cuda.synchronize()
            rec = ops.profile_records()[2:]
            ops.profile_enable(False)
            ms = sum(r[2] for r in rec) / len(rec)
            fl = 2.0 * 9 * 64 * 64 * h * w * n
            line.append("dbg%d %.1f us (%.1f TF)" % (dbg, ms * 1e3, fl / ms / 1e9))
        lib.isrDebugSetAblation(0)
        print("%d x %dx%d: " % (n, w, h) + " | ".join(line), flush=True)
