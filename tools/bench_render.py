"""Times the ray-march kernel variants alone (packet-level events) on the bench frame."""
import sys, torch
sys.path.insert(0, '.')
from isosurfacesuperresolution_amd import volumes as V
from isosurfacesuperresolution_amd.inference import DirectRenderer
res = [(480, 270), (1920, 1080)]
vol = V.ejecta(256)
r = DirectRenderer(); r.load_dense(vol)
for c, v in [("cameraLookAt", "0,0,0"), ("cameraUp", "0,1,0"), ("cameraFoV", "30.000"), ("isovalue", "0.340"), ("aosamples", "0")]:
    r.send_command(c, v)
for W, H in res:
    r.send_command("resolution", "%d,%d" % (W, H)); r.send_command("viewport", "0,0,%d,%d" % (W, H))
    out = torch.empty((H, W, 12), device='cuda')
    for variant in (0, 1):
        r.set_kernel_variant(variant)
        for k in range(3):
            r.send_command("cameraOrigin", V.fmt3(V.orbit_camera(k))); r.render_direct(out)
        r.profile_enable(True)
        for k in range(10):
            r.send_command("cameraOrigin", V.fmt3(V.orbit_camera(5 + k))); r.render_async(out, torch.cuda.current_stream())
        torch.cuda.synchronize()
        ms = r.profile_times_ms(); r.profile_enable(False)
        print("%dx%d variant %d: %.3f ms/frame (min %.3f max %.3f) checksum %.6f" % (W, H, variant, sum(ms) / len(ms), min(ms), max(ms), out[..., 3].sum().item()))
    # semantics=gvdb (the CUDA renderer's arithmetic): camera distance 1.0 because the world is half the size
    r.set_kernel_variant(0)
    r.send_command("semantics", "gvdb"); r.send_command("isovalue", "0.250")
    for k in range(3):
        r.send_command("cameraOrigin", V.fmt3(V.orbit_camera(k, distance=1.0))); r.render_direct(out)
    r.profile_enable(True)
    for k in range(10):
        r.send_command("cameraOrigin", V.fmt3(V.orbit_camera(5 + k, distance=1.0))); r.render_async(out, torch.cuda.current_stream())
    torch.cuda.synchronize()
    ms = r.profile_times_ms(); r.profile_enable(False)
    print("%dx%d semantics=gvdb: %.3f ms/frame (min %.3f max %.3f) hit pixels %.0f" % (W, H, sum(ms) / len(ms), min(ms), max(ms), out[..., 3].sum().item()))
    r.send_command("semantics", "cpu"); r.send_command("isovalue", "0.340")
