"""The ray-march kernel on its own (one launch per frame) for profiling: python tools/raymarch_only.py <volume> <WxH> [frames] [variant] [tile order mode]
volume: ejecta256 | cloud512 | ejecta512 | ejecta1024 (generated region-wise: 4.3 GB dense, beyond the 256 MiB Infinity Cache by 16x).
Prints the mean kernel time from the dispatch-packet events."""
import sys
sys.path.insert(0, '.')
import torch
from isosurfacesuperresolution_amd import volumes as V
from isosurfacesuperresolution_amd.inference import DirectRenderer

name = sys.argv[1] if len(sys.argv) > 1 else "ejecta256"
w, h = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "480x270").split("x"))
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 12
variant = int(sys.argv[4]) if len(sys.argv) > 4 else 0
order = int(sys.argv[5]) if len(sys.argv) > 5 else 0
n = int(name.replace("cloud", "").replace("ejecta", ""))
iso = 0.30 if name.startswith("cloud") else 0.34
r = DirectRenderer()
if n >= 1024:
    from isosurfacesuperresolution_amd import parallel_render as PR
    tile = PR.generate_tiles(V.EjectaField(n, seed=1024), (1, 1, 1))[0]      # the whole volume as one "tile", built without n^3 temporaries
    r.load_dense(tile["data"])
    del tile
else:
    r.load_dense(V.cloud(n) if name.startswith("cloud") else V.ejecta(n))
if order:
    r.set_tile_order_mode(order)
for c, v in [("cameraLookAt", "0,0,0"), ("cameraUp", "0,1,0"), ("cameraFoV", "30.000"), ("isovalue", "%5.3f" % iso),
             ("aosamples", "0"), ("resolution", "%d,%d" % (w, h)), ("viewport", "0,0,%d,%d" % (w, h))]:
    r.send_command(c, v)
r.set_kernel_variant(variant)
out = torch.empty((h, w, 12), dtype=torch.float32, device="cuda")
for k in range(2):
    r.send_command("cameraOrigin", V.fmt3(V.orbit_camera(k)))
    r.render_direct(out)
r.profile_enable(True)
for k in range(2, 2 + frames):
    r.send_command("cameraOrigin", V.fmt3(V.orbit_camera(k)))
    r.render_async(out, torch.cuda.current_stream())
torch.cuda.synchronize()
ms = r.profile_times_ms()
info = r.volume_info()
print("%s %dx%d variant %d order %d: %.3f ms per frame (min %.3f max %.3f), %d hit pixels in the last frame, %d bricks stored (%.0f MB)" % (
    name, w, h, variant, order, sum(ms) / len(ms), min(ms), max(ms), int((out[..., 3] == 1).sum()), info["bricks"], info["bricks"] * 2944 / 1e6))
