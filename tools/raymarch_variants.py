"""Kernel variants of the ray-marcher side by side: same frames, results compared bit for bit against the first one listed
(4 = the nested loops of the reference, 5 = flat state machine, 0 = the default: flat, two samples per iteration, 3 = 0 with the
slot table in LDS, 2 = 5 in 128 registers).  python tools/raymarch_variants.py [variants, e.g. 4,5,0] [cases, e.g. ejecta256@480x270,ejecta256@1920x1080,cloud512@1920x1080]"""
import os, sys
sys.path.insert(0, '.')
import torch
from isosurfacesuperresolution_amd import volumes as V
from isosurfacesuperresolution_amd.inference import DirectRenderer

variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "4,5,0,3,2").split(",")]
cases = (sys.argv[2] if len(sys.argv) > 2 else "ejecta256@480x270,ejecta256@1920x1080,cloud512@1920x1080").split(",")
frames = 8
r = DirectRenderer()
for case in cases:
    name, res = case.split("@")
    w, h = (int(v) for v in res.split("x"))
    n = int(name.replace("cloud", "").replace("ejecta", ""))
    vol = V.cloud(n) if name.startswith("cloud") else V.ejecta(n)
    iso = 0.30 if name.startswith("cloud") else 0.34
    r.load_dense(vol)
    for c, v in [("cameraLookAt", "0,0,0"), ("cameraUp", "0,1,0"), ("cameraFoV", "30.000"), ("isovalue", "%5.3f" % iso),
                 ("aosamples", os.environ.get("RM_AO", "0")), ("aoradius", "0.050"), ("resolution", "%d,%d" % (w, h)), ("viewport", "0,0,%d,%d" % (w, h))]:
        r.send_command(c, v)
    ref = None
    for variant in variants:
        assert r.set_kernel_variant(variant) == 0
        outs = [torch.empty((h, w, 12), dtype=torch.float32, device="cuda") for _ in range(frames)]
        for k in range(2):
            r.send_command("cameraOrigin", V.fmt3(V.orbit_camera(k)))
            r.render_direct(outs[0])
        r.profile_enable(True)
        for k in range(frames):
            r.send_command("cameraOrigin", V.fmt3(V.orbit_camera(2 + k)))
            r.render_async(outs[k], torch.cuda.current_stream())
        torch.cuda.synchronize()
        ms = r.profile_times_ms()
        r.profile_enable(False)
        same = "reference"
        if ref is None:
            ref = outs
        else:
            same = "bit-identical" if all(torch.equal(a, b) for a, b in zip(outs, ref)) else "DIFFERENT"
        print("%s %dx%d variant %d: %.3f ms per frame (min %.3f max %.3f)  %s" % (name, w, h, variant, sum(ms) / len(ms), min(ms), max(ms), same), flush=True)
    r.set_kernel_variant(0)
