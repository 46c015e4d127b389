"""BASELINE config #4 end to end on one GPU: 512^3 cloud volume -> clips of 5 frames rendered by this package's
ray-marcher (low 128x72 + ground truth 512x288 with ray-cast AO) -> 32^2 crops -> EnhanceNet training steps with
the temporal loss (temp-l2) and the warped previous-frame recurrence.  Usage: python tools/config4_cloud.py [n]"""
import argparse, sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
from isosurfacesuperresolution_amd import models, losses, train, volumes as V
from isosurfacesuperresolution_amd.dataset_video import render_clip
from isosurfacesuperresolution_amd.inference import DirectRenderer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T, B, steps = 5, 8, 24
t0 = time.perf_counter()
vol = V.cloud(n)
t_vol = time.perf_counter() - t0
r = DirectRenderer()
t0 = time.perf_counter(); r.load_dense(vol); torch.cuda.synchronize(); t_load = time.perf_counter() - t0
info = r.volume_info()
print("cloud %d^3: generated in %.1f s, bricked on the GPU in %.2f s: %d bricks stored, %d leaves" % (n, t_vol, t_load, info["bricks"], info["leaves"]), flush=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
clips = []
for c in range(B):
    origins = [V.orbit_camera(8 * c + k, K=64, distance=1.8, pitch=0.3) for k in range(T)]
    clips.append(render_clip(r, origins, (128, 72), isovalue=0.30, ao_samples=8, ao_radius=0.05))
torch.cuda.synchronize(); t_render = time.perf_counter() - t0
print("rendered %d clips x %d frames (low 128x72 + high 512x288 with 8 AO rays): %.2f s" % (B, T, t_render), flush=True)
# one 32^2 crop per clip around the image centre (where the cloud is)
lo = torch.from_numpy(np.stack([c[1][:, :, 20:52, 48:80] for c in clips])).cuda()
fl = torch.from_numpy(np.stack([c[2][:, :, 20:52, 48:80] for c in clips])).cuda()
hi = torch.from_numpy(np.stack([c[0][:, :, 80:208, 192:320] for c in clips])).cuda()
print("crop coverage (mask > 0): %.2f" % float((lo[:, :, 0] > 0).float().mean()), flush=True)
opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10,
                         losses="l1:mask:1,l1:ao:1,l1:normal:10,l1:depth:10,temp-l2:color:0.1",
                         lossAO=0.0, lossAmbient=0.1, lossDiffuse=0.9, lossSpecular=0.0)
torch.manual_seed(124)
net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt).cuda()
crit = losses.LossNetUnshaded('cuda', 5, 6, 128, 16, opt).cuda()
optim, _ = train.make_optimizer(net)            # Adam, lr 1e-4 as in mainVideoUnshaded.py:287
hist = []
for s in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    l = train.train_step(net, crit, optim, (lo, fl, hi), initial_image="zero")
    torch.cuda.synchronize(); hist.append((l, time.perf_counter() - t0))
print("loss per step:", ["%.3f" % h[0] for h in hist[::3]])
print("step time: %.1f ms (B=%d, T=%d, 32^2 -> 128^2)" % (1e3 * np.median([h[1] for h in hist[1:]]), B, T))
assert min(h[0] for h in hist[-4:]) < hist[0][0], "the loss must go down on a fixed batch"
