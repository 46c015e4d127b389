"""BASELINE config #5 rehearsed on ONE GPU: a volume split 2x2x2 in object space, each tile ray-marched for the full
960x540 image (what one rank of ``parallel_render.TiledRenderer`` does), nearest-hit composite, then the 4K frame
super-resolved in 8 screen strips one after the other (what the ranks of ``parallel_sr`` do) and checked against
the unsplit pipeline.  The volume is generated TILE-WISE (``parallel_render.generate_tiles``: every tile evaluates only its
own box + halo of the global lattice); the unsplit volume it is compared with is assembled from the tiles.
Usage: python tools/config5_tiled.py [n=512] [out.json]   (n = 1024: BASELINE config #5, ~15 GB of host memory)"""
import argparse, json, sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
from isosurfacesuperresolution_amd import models, parallel_render as PR, parallel_sr, volumes as V
from isosurfacesuperresolution_amd.inference import DirectRenderer, LoadedModel
from isosurfacesuperresolution_amd.pipeline import default_shading

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
W, H = 960, 540
out_json = sys.argv[2] if len(sys.argv) > 2 else None
t0 = time.perf_counter()
tiles = PR.generate_tiles(V.EjectaField(n, seed=1024 if n == 1024 else 272), (2, 2, 2))
t_gen = time.perf_counter() - t0
print("ejecta %d^3 generated tile-wise (8 tiles of %s + halo) in %.1f s" % (n, "x".join(str(d) for d in tiles[0]["data"].shape), t_gen), flush=True)
r = DirectRenderer()
origin = V.orbit_camera(9)

def set_camera():
    for c, v in (("cameraLookAt", "0,0,0"), ("cameraUp", "0,1,0"), ("cameraFoV", "30.000"), ("isovalue", "0.340"), ("aosamples", "0"),
                 ("resolution", "%d,%d" % (W, H)), ("viewport", "0,0,%d,%d" % (W, H)), ("cameraOrigin", V.fmt3(origin))):
        r.send_command(c, v)

set_camera()      # before the first load: a load makes the current camera the flow reference ("last camera")

def render_current(buf):
    r.profile_enable(True)
    r.render_direct(buf)
    ms = r.profile_times_ms()[-1]
    r.profile_enable(False)
    return ms

gb = torch.empty((8, H, W, 12), device="cuda")
times = []
for k, tile in enumerate(tiles):
    t0 = time.perf_counter(); r.load_tile(tile); torch.cuda.synchronize(); t_load = time.perf_counter() - t0
    times.append((t_load, render_current(gb[k])))
print("per tile: load %.2f s, ray-march of the full 960x540 image %.2f ms (max %.2f)" % (
    np.mean([t[0] for t in times]), np.mean([t[1] for t in times]), max(t[1] for t in times)), flush=True)
comp = PR.composite(gb)
torch.cuda.synchronize(); t0 = time.perf_counter()
comp = PR.composite(gb)
torch.cuda.synchronize(); print("composite of 8 G-buffers: %.2f ms (all-gather payload per rank %.1f MB)" % ((time.perf_counter() - t0) * 1e3, H * W * 48 / 1e6), flush=True)
vol = PR.assemble(tiles, (n, n, n))
print("unsplit volume assembled (%.1f GB)" % (vol.nbytes / 1e9), flush=True)
t0 = time.perf_counter(); r.load_dense(vol); torch.cuda.synchronize(); t_load_whole = time.perf_counter() - t0
info = r.volume_info()
whole = torch.empty((H, W, 12), device="cuda")
ms_whole = render_current(whole)
diff_mask = int((comp[..., 3] != whole[..., 3]).sum())
identical = bool(torch.equal(comp, whole))
hits = int((whole[..., 3] == 1).sum())
print("unsplit volume: load %.1f s (%d bricks), ray-march %.2f ms, %d hit pixels; tiled composite vs unsplit: %d silhouette pixels differ, "
      "all 12 channels bit-identical: %s" % (t_load_whole, info["bricks"], ms_whole, hits, diff_mask, identical), flush=True)
del vol

opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10)
torch.manual_seed(0)
net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt)
lm = LoadedModel.from_model(net, "cuda", parameters={"initialImage": "zero"})
sr = parallel_sr.StripSuperResolution(lm, default_shading("cuda", 30.0))
with torch.no_grad():
    x = sr.network_input(comp)
    sr.compute_strip(x, 0, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    full_raw, full_rgb = sr.compute_strip(x, 0, 1)
    torch.cuda.synchronize(); t_full = time.perf_counter() - t0
    parts, t_strip = [], []
    for rank in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        parts.append(sr.compute_strip(x, rank, 8))
        torch.cuda.synchronize(); t_strip.append(time.perf_counter() - t0)
    same = torch.equal(torch.cat([p[0] for p in parts], dim=2), full_raw)
    # the same frame as the (rows x columns) grid of screen tiles with the smallest largest tile + halo (round 5)
    grid = parallel_sr.best_grid(8, H, W)
    t_tile, same_grid = [], True
    for rank in range(8):
        y0, y1, x0, x1 = parallel_sr.tile_bounds(H, W, grid, rank)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        raw_t, rgb_t = sr.compute_strip(x, rank, 8, grid=grid)
        torch.cuda.synchronize(); t_tile.append(time.perf_counter() - t0)
        same_grid = same_grid and torch.equal(raw_t, full_raw[:, :, 4 * y0:4 * y1, 4 * x0:4 * x1]) and torch.equal(rgb_t, full_rgb[:, :, 4 * y0:4 * y1, 4 * x0:4 * x1])
print("4K super-resolution: whole frame %.1f ms; 8 strips %.1f ms each (max %.1f); strips == whole frame bit for bit: %s; %d x %d tiles %.1f ms each (max %.1f), bit for bit: %s" % (
    t_full * 1e3, 1e3 * np.mean(t_strip), 1e3 * max(t_strip), same, grid[0], grid[1], 1e3 * np.mean(t_tile), 1e3 * max(t_tile), same_grid))
result = {"config": "BASELINE #5 rehearsed on one GPU", "volume": "ejecta%d (seed %d), generated tile-wise" % (n, 1024 if n == 1024 else 272),
          "tiles": "2x2x2 + 8-voxel halo", "image": [W, H], "generate_s": round(t_gen, 1),
          "tile_load_s_mean": round(float(np.mean([t[0] for t in times])), 2),
          "tile_raymarch_ms": [round(t[1], 3) for t in times], "unsplit_raymarch_ms": round(ms_whole, 3), "unsplit_bricks": info["bricks"],
          "hit_pixels": hits, "composite_mask_mismatches": diff_mask, "composite_bit_identical_12ch": identical,
          "sr_4k_whole_ms": round(t_full * 1e3, 2), "sr_4k_strip_ms": [round(t * 1e3, 2) for t in t_strip], "sr_strips_bit_identical": bool(same),
          "sr_tile_grid": list(grid), "sr_4k_tile_ms": [round(t * 1e3, 2) for t in t_tile], "sr_tiles_bit_identical": bool(same_grid)}
print(json.dumps(result), flush=True)
if out_json:
    with open(out_json, "w") as f:
        json.dump(result, f, indent=1)
assert same and same_grid and identical and diff_mask == 0
