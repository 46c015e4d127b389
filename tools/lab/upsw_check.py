"""Form 8 (csrc/sr_conv_upsw.h) against form 3 on a few shapes: where do the results differ?  PYTHONPATH=. python tools/lab/upsw_check.py"""
import ctypes
import torch
from isosurfacesuperresolution_amd import ops
lib = ops._sr()
lib.isrDebugSetSplitUpsForm.argtypes = [ctypes.c_int]
ops.RANGE_GUARD = False
for (h, w) in ((4, 16), (17, 34), (64, 64), (135, 240), (270, 480)):
    g = torch.Generator().manual_seed(h * 1000 + w)
    x = ((torch.rand(1, 64, h, w, generator=g) - 0.4) * 3).cuda()
    wt = ((torch.rand(64, 64, 3, 3, generator=g) - 0.5) * 0.2).cuda()
    b = ((torch.rand(64, generator=g) - 0.5) * 0.3).cuda()
    with torch.no_grad():
        lib.isrDebugSetSplitUpsForm(3)
        a = ops.conv3x3_split(x, wt, b, act='relu', upsample2x=True)
        lib.isrDebugSetSplitUpsForm(8)
        c = ops.conv3x3_split(x, wt, b, act='relu', upsample2x=True)
        torch.cuda.synchronize()
        d = (a - c).abs()
        bad = (d > 0)
        nz = bad.nonzero()
        tiles = sorted(set(((r // 16), (c // 32)) for r, c in zip(nz[:, 2].tolist(), nz[:, 3].tolist())))
        print("   tiles (ty, tx) with differences (%d): %s" % (len(tiles), tiles[:60]))
        rows_in_tile = sorted(set(r % 16 for r in nz[:, 2].tolist()))
        print("   rows within tile:", rows_in_tile, " channels:", sorted(set(nz[:, 1].tolist()))[:70])
        print("%dx%d: max diff %.3g, %d of %d values differ; channels with differences: %s; rows: %s; cols: %s" % (
            h, w, d.max().item(), int(bad.sum()), bad.numel(), sorted(set(bad.nonzero()[:, 1].tolist()))[:12],
            sorted(set(bad.nonzero()[:, 2].tolist()))[:20], sorted(set(bad.nonzero()[:, 3].tolist()))[:40]), flush=True)
lib.isrDebugSetSplitUpsForm(3)
