"""BASELINE config #5 rehearsed on ONE GPU: a volume split 2x2x2 in object space, each tile ray-marched for the full
960x540 image (what one rank of ``parallel_render.TiledRenderer`` does), nearest-hit composite, then the 4K frame
super-resolved in 8 screen strips one after the other (what the ranks of ``parallel_sr`` do) and checked against
the unsplit pipeline.  Usage: python tools/config5_tiled.py [n=512]   (n = 1024 needs ~10 GB of host memory)"""
import argparse, sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
from isosurfacesuperresolution_amd import models, parallel_render as PR, parallel_sr, volumes as V
from isosurfacesuperresolution_amd.inference import DirectRenderer, LoadedModel
from isosurfacesuperresolution_amd.pipeline import default_shading

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
W, H = 960, 540
t0 = time.perf_counter(); vol = V.ejecta(n); print("ejecta %d^3 generated in %.1f s" % (n, time.perf_counter() - t0), flush=True)
tiles = PR.partition_volume(vol, (2, 2, 2))
r = DirectRenderer()
origin = V.orbit_camera(9)

def render_current(buf):
    for c, v in (("cameraLookAt", "0,0,0"), ("cameraUp", "0,1,0"), ("cameraFoV", "30.000"), ("isovalue", "0.340"), ("aosamples", "0"),
                 ("resolution", "%d,%d" % (W, H)), ("viewport", "0,0,%d,%d" % (W, H)), ("cameraOrigin", V.fmt3(origin))):
        r.send_command(c, v)
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
r.load_dense(vol)
whole = torch.empty((H, W, 12), device="cuda")
ms_whole = render_current(whole)
diff_mask = int((comp[..., 3] != whole[..., 3]).sum())
both = (comp[..., 3] == 1) & (whole[..., 3] == 1)
d = (comp - whole)[both][:, [0, 1, 2, 4, 5, 6, 7]].abs()
err = float(d.max())
off = int((d.max(dim=1).values > 1e-4).sum())
print("unsplit volume: %.2f ms; tiled vs unsplit: %d silhouette pixels differ; %d of %d common hits differ by more than 1e-4 "
      "(max %.2g; per channel r,g,b,nx,ny,nz,depth: %s)" % (ms_whole, diff_mask, off, int(both.sum()), err,
      ["%.1g" % v for v in d.max(dim=0).values.tolist()]), flush=True)

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
print("4K super-resolution: whole frame %.1f ms; 8 strips %.1f ms each (max %.1f); strips == whole frame bit for bit: %s" % (
    t_full * 1e3, 1e3 * np.mean(t_strip), 1e3 * max(t_strip), same))
assert same and diff_mask <= 8 and off <= 0.002 * int(both.sum())   # a tile boundary that cuts a voxel-level bracket moves the 5-bisection estimate
