#!/usr/bin/env python
"""Is the default bench frame host bound on this box?  Enqueue K frames WITHOUT waiting and compare the time the host needs to
issue them with the time the GPU needs to run them (tools; prints one JSON line)."""
import argparse, contextlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from isosurfacesuperresolution_amd import models, volumes as V
from isosurfacesuperresolution_amd.inference import DirectRenderer, LoadedModel
from isosurfacesuperresolution_amd.pipeline import SuperResolutionPipeline, default_shading

K = int(sys.argv[1]) if len(sys.argv) > 1 else 60
renderer = DirectRenderer()
renderer.load_dense(V.VOLUMES["ejecta256"][0]())
opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10)
torch.manual_seed(0)
with contextlib.redirect_stdout(sys.stderr):
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt)
model = LoadedModel.from_model(net, "cuda", parameters={"initialImage": "zero"})
pipe = SuperResolutionPipeline(renderer, model, default_shading("cuda", 30.0), (480, 270))
pipe.set_static(fov=30.0, isovalue=0.34)
origins = [V.orbit_camera(k, K=64) for k in range(K + 8)]
for k in range(6):
    pipe.frame(origins[k], origins[k + 1])
torch.cuda.synchronize()
out = {}
for name, overlap in (("overlap", True), ("no_overlap", False)):
    pipe.reset()
    pipe.frame(origins[0], origins[1] if overlap else None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(1, K + 1):
        pipe.frame(origins[k], origins[k + 1] if overlap else None)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    out[name] = {"host_issue_ms_per_frame": (t1 - t0) / K * 1e3, "total_ms_per_frame": (t2 - t0) / K * 1e3,
                 "gpu_tail_after_last_issue_ms": (t2 - t1) * 1e3}
# the same K frames with a synchronisation after every frame: the GPU's time for ONE frame from an idle queue
pipe.reset()
pipe.frame(origins[0]); torch.cuda.synchronize()
ts = []
for k in range(1, 21):
    t0 = time.perf_counter(); pipe.frame(origins[k]); th = time.perf_counter(); torch.cuda.synchronize(); t1 = time.perf_counter()
    ts.append((th - t0, t1 - t0))
out["synced_single_frames"] = {"host_issue_ms": sum(a for a, _ in ts) / len(ts) * 1e3, "frame_ms": sum(b for _, b in ts) / len(ts) * 1e3}
out["cpu_count"] = os.cpu_count()
print(json.dumps(out))
