"""Cost-ordered dispatch of the ray-marcher (isoSetTileOrderMode): the default kernel alone at 480x270 on the bench volume, per
mode, interleaved rounds after a long warm-up; every mode's frames must equal mode 0's bit for bit.
usage: PYTHONPATH=. python tools/lab/raymarch_order.py [WxH]"""
import sys

import torch

from isosurfacesuperresolution_amd import volumes as V
from isosurfacesuperresolution_amd.inference import DirectRenderer

w, h = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "480x270").split("x"))
r = DirectRenderer()
r.load_dense(V.ejecta(256))
for c, v in [("cameraLookAt", "0,0,0"), ("cameraUp", "0,1,0"), ("cameraFoV", "30.000"), ("isovalue", "0.340"),
             ("aosamples", "0"), ("resolution", "%d,%d" % (w, h)), ("viewport", "0,0,%d,%d" % (w, h))]:
    r.send_command(c, v)
out = torch.empty((h, w, 12), dtype=torch.float32, device="cuda")


def run(mode, frames=24, keep=False):
    r.set_tile_order_mode(mode)
    r.send_command("cameraOrigin", V.fmt3(V.orbit_camera(-1)))
    r.render_direct(out)
    imgs = []
    r.profile_enable(True)
    for k in range(frames):
        r.send_command("cameraOrigin", V.fmt3(V.orbit_camera(k)))
        r.render_async(out, torch.cuda.current_stream())
        if keep:
            imgs.append(out.clone())
    torch.cuda.synchronize()
    ms = r.profile_times_ms()[2:]
    r.profile_enable(False)
    return sum(ms) / len(ms), imgs


for _ in range(20):
    run(0)
ref = run(0, keep=True)[1]
res = {0: [], 1: [], 2: []}
for rnd in range(3):
    for mode in (0, 1, 2):
        t, imgs = run(mode, keep=(rnd == 0))
        res[mode].append(t)
        if rnd == 0:
            assert all(torch.equal(a, b) for a, b in zip(imgs, ref)), "mode %d changed the G-buffer" % mode
for mode, name in ((0, "XCD-aware scan order"), (1, "heaviest first"), (2, "heaviest first, lightest as second waves")):
    print("mode %d %-42s %s ms per frame" % (mode, name, "  ".join("%.3f" % t for t in res[mode])))
r.set_tile_order_mode(0)
