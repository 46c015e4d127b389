"""1080p 64->64 forward conv with padded channel planes (ops.PLANE_PAD_FLOATS): which paddings avoid the plane aliasing?"""
import sys, torch
sys.path.insert(0, '.')
from isosurfacesuperresolution_amd import ops
lib = ops._sr()
h, w = 1080, 1920
with torch.no_grad():
    wt = (torch.rand(64, 64, 3, 3, device='cuda') - 0.5) * 0.1
    b = torch.rand(64, device='cuda')
    ref = None
    for pad_in, pad_out in [(0, 0), (0, 64), (64, 0), (64, 64), (256, 256), (1024, 1024), (1920, 1920), (4096, 4096), (16, 16), (32, 32), (96, 96), (4160, 4160)]:
        ops.plane_pad = (lambda h_, w_, v=pad_in: v)
        x = ops.empty_planes(1, 64, h, w, 'cuda')
        torch.manual_seed(0)
        x.copy_(torch.rand(1, 64, h, w, device='cuda') - 0.5)
        ops.plane_pad = (lambda h_, w_, v=pad_out: v)
        line = []
        for ups in (False,):
            ops.profile_enable(True)
            for _ in range(8): y = ops.conv3x3(x, wt, b, act='relu')
            torch.cuda.synchronize()
            rec = ops.profile_records()[2:]
            ops.profile_enable(False)
            ms = sum(r[2] for r in rec) / len(rec)
            fl = 2.0 * 9 * 64 * 64 * h * w
            line.append("%.1f us (%.1f TF)" % (ms * 1e3, fl / ms / 1e9))
        if ref is None:
            ref = y.clone()
        print("pad in %5d out %5d floats: " % (pad_in, pad_out) + " | ".join(line) + " | equal to packed: %s" % torch.equal(ref, y), flush=True)
        del x, y
