"""Does the channel-plane stride matter?  Same kernel, nearly the same work, different H x W (plane size in bytes)."""
import sys, torch
sys.path.insert(0, '.')
from isosurfacesuperresolution_amd import ops
lib = ops._sr()
with torch.no_grad():
    for n, h, w in [(1, 1080, 1920), (1, 1080, 1928), (1, 1081, 1920), (1, 1082, 1920), (1, 1088, 1920), (1, 1080, 1952), (1, 1072, 1920), (1, 1024, 2048), (1, 1000, 2000)]:
        x = torch.rand(n, 64, h, w, device='cuda') - 0.5
        wt = (torch.rand(64, 64, 3, 3, device='cuda') - 0.5) * 0.1
        b = torch.rand(64, device='cuda')
        line = []
        for dbg in (0, 16, 2):
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
        print("%d x %dx%d plane 0x%x: " % (n, w, h, h * w * 4) + " | ".join(line), flush=True)
