"""Which fused training kernel moves the weight gradients of a clip, and by how much: module path vs fused loss vs own
upsampling (relative L2 per parameter against the PyTorch module path; two runs of the module path differ by ~8e-7
because of its atomics)."""
import argparse, sys
sys.path.insert(0, '.')
import torch, torch.nn.functional as F
from isosurfacesuperresolution_amd import models, losses as L, train, ops
RECIPE = "l1:mask:1,l1:ao:1,l1:normal:10,l1:depth:10,temp-l2:color:0.1"
opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=2, losses=sys.argv[1] if len(sys.argv) > 1 else RECIPE,
                         lossAO=0.0, lossAmbient=0.1, lossDiffuse=0.9, lossSpecular=0.0)
g = torch.Generator().manual_seed(3)
B, T = 2, 3
inp = torch.rand(B, T, 5, 16, 16, generator=g).cuda(); inp[:, :, 0] = inp[:, :, 0] * 2 - 1
flow = ((torch.rand(B, T, 2, 16, 16, generator=g) - 0.5) * 0.05).cuda()
tgt = torch.rand(B, T, 6, 64, 64, generator=g).cuda(); tgt[:, :, 0] = tgt[:, :, 0] * 2 - 1
own = ops.bilinear_upsample2x
ref = lambda x: F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)
res = {}
for name, fl, up in (("module/torch", False, ref), ("module/torch again", False, ref), ("fused/torch", True, ref), ("module/own", False, own), ("fused/own", True, own)):
    torch.manual_seed(124)
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt).cuda()
    crit = L.LossNetUnshaded('cuda', 5, 6, 64, 8, opt).cuda()
    crit.fused = fl
    ops.bilinear_upsample2x = up
    loss, loss_sum = train.clip_loss(net, crit, inp, flow, tgt, initial_image="zero")
    loss.backward()
    res[name] = (loss_sum.item(), [p.grad.detach().clone() for p in net.parameters()])
base = res["module/torch"]
for name, (l, gs) in res.items():
    errs = [((a - b).norm() / b.norm()).item() for a, b in zip(gs, base[1])]
    print("%-20s loss %.7f  max rel %.2e  median %.2e" % (name, l, max(errs), sorted(errs)[len(errs) // 2]))
