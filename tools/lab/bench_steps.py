"""Per-launch view of one steady-state frame of the bench loop (diagnostic): every conv dispatch with its time."""
import sys
sys.path.insert(0, '.')
import argparse
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
overlap = "overlap" in sys.argv
K = 12
for k in range(K):
    if k == K - 3:
        torch.cuda.synchronize(); ops.profile_enable(True)
    pipe.frame(V.orbit_camera(k), V.orbit_camera(k + 1) if overlap else None)
torch.cuda.synchronize()
rec = ops.profile_records()
ops.profile_enable(False)
n = len(rec) // 3
for name, fl, ms in rec[-n:]:
    print("%-32s %8.1f GF %8.1f us %6.1f TF" % (name, fl / 1e9, ms * 1e3, fl / ms / 1e9))
print("sum %.3f ms" % sum(ms for _, _, ms in rec[-n:]))
