"""Ablations of conv3x3_split_upsp_kernel to localise a corruption: with the MFMAs skipped (dbg 1) the body must be relu-free bias."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from isosurfacesuperresolution_amd import ops
h, w, dbg = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
g = torch.Generator().manual_seed(1)
wt = ((torch.rand(64, 64, 3, 3, generator=g) - 0.5) * 0.12).cuda()
b = ((torch.rand(64, generator=g) - 0.5) * 0.2).cuda()
x = ((torch.rand(1, 64, h, w, generator=g) - 0.3) * 2).cuda()
xp = ops.pack_split(x)
ops._sr().isrDebugSetSplitAblation(dbg)
for it in range(3):
    y = ops.conv3x3_ups_phase(xp, wt, b, act='none')
    torch.cuda.synchronize()
    got = y.to_float()[0, :, 1:-1, 1:-1]
    ref = b.view(64, 1, 1).expand_as(got)
    bad = torch.nonzero((got - ref).abs() > 1e-3)
    print("run", it, "dbg", dbg, "bad elements vs bias:", bad.shape[0], bad[:5].tolist())
