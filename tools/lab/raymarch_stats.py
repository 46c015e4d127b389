"""Latency model of the ray-march kernel from its instrumented twin (iso_render_stats): per 8x8 tile the wave's
cycles and the step counts of its busiest ray.  python tools/lab/raymarch_stats.py <volume> <WxH> [out.json]"""
import ctypes, json, sys
import numpy as np
sys.path.insert(0, '.')
import torch
from isosurfacesuperresolution_amd import volumes as V
from isosurfacesuperresolution_amd.inference import DirectRenderer

name = sys.argv[1] if len(sys.argv) > 1 else "ejecta256"
w, h = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "480x270").split("x"))
n = int(name.replace("cloud", "").replace("ejecta", ""))
vol = V.cloud(n) if name.startswith("cloud") else V.ejecta(n)
iso = 0.30 if name.startswith("cloud") else 0.34
r = DirectRenderer()
r.load_dense(vol)
for c, v in [("cameraLookAt", "0,0,0"), ("cameraUp", "0,1,0"), ("cameraFoV", "30.000"), ("isovalue", "%5.3f" % iso),
             ("aosamples", "0"), ("resolution", "%d,%d" % (w, h)), ("viewport", "0,0,%d,%d" % (w, h))]:
    r.send_command(c, v)
r.set_kernel_variant(int(__import__('os').environ.get('RM_VARIANT', '0')))
out = torch.empty((h, w, 12), dtype=torch.float32, device="cuda")
tiles = ((w + 7) // 8) * ((h + 7) // 8)
stats = torch.zeros((tiles, 6), dtype=torch.int64, device="cuda")
r.lib.isoDebugSetStatsBuffer.argtypes = [ctypes.c_ulonglong]
# plain kernel time of the same frames
r.profile_enable(True)
for k in range(2, 10):
    r.send_command("cameraOrigin", V.fmt3(V.orbit_camera(k)))
    r.render_async(out, torch.cuda.current_stream())
torch.cuda.synchronize()
ms = r.profile_times_ms()[1:]
r.profile_enable(False)
ref = out.clone()
r.send_command("cameraOrigin", V.fmt3(V.orbit_camera(8)))
r.render_direct(out)
r.lib.isoDebugSetStatsBuffer(ctypes.c_ulonglong(stats.data_ptr()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
r.send_command("cameraOrigin", V.fmt3(V.orbit_camera(9)))
e0.record(); r.render_async(out, torch.cuda.current_stream()); e1.record()
torch.cuda.synchronize()
r.lib.isoDebugSetStatsBuffer(ctypes.c_ulonglong(0))
assert torch.equal(out, ref), "the instrumented kernel must render the same frame"
s = stats.cpu().numpy().astype(np.float64)
cyc, best, lv, sk, tot, hits = s.T
clock = 100e6                                   # s_memtime ticks at 100 MHz on gfx950
k = int(np.argmax(cyc))
busy = cyc > 0
res = {"volume": name, "resolution": [w, h], "tiles": tiles, "kernel_ms_plain": float(np.mean(ms)), "kernel_ms_instrumented": e0.elapsed_time(e1),
       "wave_us_mean": float(cyc[busy].mean() / clock * 1e6), "wave_us_p50": float(np.median(cyc[busy]) / clock * 1e6),
       "wave_us_p99": float(np.percentile(cyc[busy], 99) / clock * 1e6), "wave_us_max": float(cyc.max() / clock * 1e6),
       "longest_wave": {"busiest_ray_samples": int(best[k]), "leaves_marched": int(lv[k]), "leaves_skipped": int(sk[k]),
                        "samples_all_rays": int(tot[k]), "hits": int(hits[k])},
       "samples_total": int(tot.sum()), "samples_busiest_ray_max": int(best.max()), "samples_per_ray_mean": float(tot.sum() / (w * h)),
       "lane_utilisation": float(tot.sum() / max(1.0, (best * 64).sum())),
       "us_per_sample_of_busiest_ray_p50": float(np.median(cyc[best > 20] / clock * 1e6 / best[best > 20]))}
# least squares: wave time = a + b * samples of its busiest ray + c * (leaves marched + skipped by it)
A = np.stack([np.ones(busy.sum()), best[busy], (lv + sk)[busy]], axis=1)
coef, *_ = np.linalg.lstsq(A, cyc[busy] / clock * 1e6, rcond=None)
res["fit_wave_us"] = {"const": float(coef[0]), "per_sample": float(coef[1]), "per_leaf_step": float(coef[2])}
print(json.dumps(res, indent=1))
if len(sys.argv) > 3:
    json.dump(res, open(sys.argv[3], "w"), indent=1)
