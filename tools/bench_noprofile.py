"""Frame time of the bench loop WITHOUT per-dispatch profiling events (how much do the events cost?)."""
import sys, time, argparse
sys.path.insert(0, '.')
import torch
from isosurfacesuperresolution_amd import models, ops, volumes as V
from isosurfacesuperresolution_amd.inference import DirectRenderer, LoadedModel
from isosurfacesuperresolution_amd.pipeline import SuperResolutionPipeline, default_shading
vol = V.ejecta(256)
r = DirectRenderer(); r.load_dense(vol)
opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10)
torch.manual_seed(0)
net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt)
model = LoadedModel.from_model(net, "cuda", parameters={"initialImage": "zero"})
pipe = SuperResolutionPipeline(r, model, default_shading("cuda", 30.0), (480, 270))
pipe.set_static(fov=30.0, isovalue=0.34)
K = 60
for prof in (False, True, False, True):
    for overlap in (True, False):
        pipe.reset()
        for k in range(5):
            pipe.frame(V.orbit_camera(k), V.orbit_camera(k + 1) if overlap else None)
        torch.cuda.synchronize()
        ops.profile_enable(prof); r.profile_enable(prof)
        t0 = time.perf_counter()
        for k in range(K):
            pipe.frame(V.orbit_camera(5 + k), V.orbit_camera(6 + k) if overlap and k + 1 < K else None)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        ops.profile_enable(False); r.profile_enable(False)
        print("profiling events %-5s overlap %-5s: %.3f ms/frame  %.1f fps" % (prof, overlap, dt * 1e3, 1 / dt), flush=True)
