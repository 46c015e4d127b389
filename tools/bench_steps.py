"""Per-step view of the bench loop (diagnostic): how the conv time evolves over the frames."""
import sys, json, subprocess
sys.path.insert(0, '.')
import torch, argparse
from isosurfacesuperresolution_amd import models, ops, volumes as V
from isosurfacesuperresolution_amd.inference import DirectRenderer, LoadedModel
from isosurfacesuperresolution_amd.pipeline import SuperResolutionPipeline, default_shading
vol = V.ejecta(256)
r = DirectRenderer(); r.load_dense(vol)
opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10)
torch.manual_seed(0)
net = models.createNetwork('EnhanceNet', 4, 101, [0,1,2,3,4], 6, opt)
model = LoadedModel.from_model(net, "cuda", parameters={"initialImage": "zero"})
pipe = SuperResolutionPipeline(r, model, default_shading("cuda", 30.0), (480, 270), temporal=("notemp" not in sys.argv))
pipe.set_static(fov=30.0, isovalue=0.34)
sink = []
ops.set_profile_sink(sink)
stats = []
for k in range(40):
    n0 = len(sink)
    rgb, raw = pipe.frame(V.orbit_camera(k))
    torch.cuda.synchronize()
    per = {}
    for name, fl, g, e0, e1 in sink[n0:]:
        per.setdefault(name, 0.0); per[name] += e0.elapsed_time(e1)
    lowres = [e0.elapsed_time(e1) for name, fl, g, e0, e1 in sink[n0:] if abs(fl - 2*9*64*64*270*480) < 1]
    stats.append((per, sum(lowres)/len(lowres), raw.abs().max().item(), raw.abs().mean().item()))
for k, (per, lr, mx, mean) in enumerate(stats):
    if k % 3 == 0: print(k, {n[-9:]: round(v, 2) for n, v in per.items()}, 'lowres us %.1f' % (lr*1e3), 'raw max %.3g mean %.3g' % (mx, mean))
