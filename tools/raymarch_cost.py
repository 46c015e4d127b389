"""What the ray-march on the side stream costs the MAIN stream: the headline pipeline (render(t+1) beside SR(t)) with the
render launch switched off after the warm-up -- the side stream still fills the flow of the (stale, pre-rendered) G-buffer, everything on the main
stream is unchanged -- against the same frames with the render on; interleaved, five rounds.  PYTHONPATH=. python tools/raymarch_cost.py"""
import argparse
import contextlib
import sys
import time

import torch
from isosurfacesuperresolution_amd import models, ops, volumes as V
from isosurfacesuperresolution_amd.inference import DirectRenderer, LoadedModel
from isosurfacesuperresolution_amd.pipeline import SuperResolutionPipeline, default_shading

renderer = DirectRenderer()
renderer.load_dense(V.VOLUMES["ejecta256"][0]())
opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10)
torch.manual_seed(0)
with contextlib.redirect_stdout(sys.stderr):
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt)
model = LoadedModel.from_model(net, "cuda", parameters={"initialImage": "zero"})
pipe = SuperResolutionPipeline(renderer, model, default_shading("cuda", 30.0), (480, 270))
pipe.set_static(fov=30.0, isovalue=0.34)
K = 200
cams = [V.orbit_camera(k, K=64) for k in range(K + 2)]
real_render = renderer.render_async


def run(render_on):
    renderer.render_async = real_render if render_on else (lambda tensor, stream=None: None)
    pipe.reset()
    renderer.render_async = real_render
    for k in range(10):                                   # (the first frames always render: the G-buffers hold a real frame)
        pipe.frame(cams[k], cams[k + 1])
    renderer.render_async = real_render if render_on else (lambda tensor, stream=None: None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        pipe.frame(cams[k], cams[k + 1] if k + 1 < K else None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    renderer.render_async = real_render
    return dt / K * 1e3


on, off = [], []
for rnd in range(5):
    on.append(run(True))
    off.append(run(False))
    print("round %d: render on the side stream %.4f ms per frame, render switched off %.4f ms" % (rnd, on[-1], off[-1]), flush=True)
a, b = sorted(on)[2], sorted(off)[2]
print("median: %.4f ms with the ray-march beside the network, %.4f ms without: the guest costs the main stream %.4f ms per frame = %.2f %%"
      % (a, b, a - b, 100 * (a - b) / a))
