"""Per-rank critical path of parallel_sr at 960x540 -> 3840x2160 (BASELINE config #5's SR half) on ONE GPU:
replicated input assembly + the tile of the rank with the largest extended area (a middle strip has both halos), for world = 1, 2, 4, 8
as strips and -- where it differs -- as the best (rows x columns) grid."""
import sys
sys.path.insert(0, '.')
import argparse
import torch
from isosurfacesuperresolution_amd import models, ops, parallel_sr
from isosurfacesuperresolution_amd.inference import LoadedModel
from isosurfacesuperresolution_amd.pipeline import default_shading
opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10)
torch.manual_seed(0)
net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt)
lm = LoadedModel.from_model(net, "cuda", parameters={"initialImage": "zero"})
sr = parallel_sr.StripSuperResolution(lm, default_shading("cuda", 30.0))
_size = [a for a in sys.argv[1:] if "x" in a]
h, w = (int(v) for v in (_size[0] if _size else "540x960").split("x"))
if "f16" in sys.argv:
    ops.FAST_F16 = True          # the opt-in fp16 fast mode, not the parity path
g = torch.rand(h, w, 12, device="cuda")
g[..., 3] = (g[..., 3] > 0.4).float()
g[..., 8:10] = (g[..., 8:10] - 0.5) * 0.02
with torch.no_grad():
    sr.previous = torch.rand(1, 6, 4 * h, 4 * w, device="cuda")
    for world in (1, 2, 4, 8):
        for grid in sorted({(world, 1), parallel_sr.best_grid(world, h, w)}, key=lambda gr: gr[1]):
            rank = max(range(world), key=lambda r: (lambda b: (min(h, b[1] + 24) - max(0, b[0] - 24)) * (min(w, b[3] + 24) - max(0, b[2] - 24)))(
                parallel_sr.tile_bounds(h, w, grid, r)))
            for it in range(3):
                if it == 1:
                    torch.cuda.synchronize()
                    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                    e0.record()
                y0, y1, x0, x1 = parallel_sr.tile_bounds(h, w, grid, rank)
                x = sr.network_input(g, rows=(max(0, y0 - 24), min(h, y1 + 24)), cols=(max(0, x0 - 24), min(w, x1 + 24)))
                if it == 1: e1.record()
                raw, rgb = sr.compute_strip(x, rank, world, grid=grid)
                if it == 1: e2.record()
            torch.cuda.synchronize()
            print("world %d, grid %dx%d: tile %dx%d + halo = %.2fx its share -> assembly %.2f ms + tile %.2f ms = %.2f ms; all-gather payload %.1f MB per rank" % (
                world, grid[0], grid[1], y1 - y0, x1 - x0, parallel_sr.extended_area(h, w, grid) * world / float(h * w), e0.elapsed_time(e1),
                e1.elapsed_time(e2), e0.elapsed_time(e2), 9 * (y1 - y0) * 4 * (x1 - x0) * 4 * 4 / 1e6), flush=True)
