"""How long does the CPU need to ENQUEUE one frame (no synchronisation inside the loop)?  If this approaches the GPU
frame time the pipeline is launch bound."""
import sys, time, argparse
sys.path.insert(0, '.')
import torch
from isosurfacesuperresolution_amd import models, ops, volumes as V
from isosurfacesuperresolution_amd.inference import DirectRenderer, LoadedModel
from isosurfacesuperresolution_amd.pipeline import SuperResolutionPipeline, default_shading
r = DirectRenderer(); r.load_dense(V.ejecta(256))
opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10)
torch.manual_seed(0)
net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt)
model = LoadedModel.from_model(net, "cuda", parameters={"initialImage": "zero"})
pipe = SuperResolutionPipeline(r, model, default_shading("cuda", 30.0), (480, 270))
pipe.set_static(fov=30.0, isovalue=0.34)
for k in range(5):
    pipe.frame(V.orbit_camera(k), V.orbit_camera(k + 1))
torch.cuda.synchronize()
K = 40
per = []
t_all = time.perf_counter()
for k in range(K):
    t0 = time.perf_counter()
    pipe.frame(V.orbit_camera(5 + k), V.orbit_camera(6 + k))
    per.append(time.perf_counter() - t0)
t_enq = time.perf_counter() - t_all
torch.cuda.synchronize()
t_tot = time.perf_counter() - t_all
per.sort()
print("enqueue per frame: median %.2f ms, min %.2f, max %.2f; whole loop enqueued in %.1f ms, finished in %.1f ms (%.2f ms/frame)" % (
    per[K // 2] * 1e3, per[0] * 1e3, per[-1] * 1e3, t_enq * 1e3, t_tot * 1e3, t_tot / K * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for k in range(10):
    pipe.frame(V.orbit_camera(50 + k), V.orbit_camera(51 + k))
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
