"""Training-step throughput (BASELINE config #3 shape per GPU): B clips of T frames, 32^2 -> 128^2,
README loss recipe, Adam.  Synthetic clips.  Usage: python tools/bench_train.py [B] [T] [steps]"""
import argparse, sys, time
sys.path.insert(0, '.')
import torch
from isosurfacesuperresolution_amd import models, losses, train, ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
T = int(sys.argv[2]) if len(sys.argv) > 2 else 10
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10,
                         losses="l1:mask:1,l1:ao:1,l1:normal:10,l1:depth:10,temp-l2:color:0.1",
                         lossAO=0.0, lossAmbient=0.1, lossDiffuse=0.9, lossSpecular=0.0)
if "bf16" in sys.argv:
    ops.TRAIN_BF16 = True       # forward + data-gradient convolutions with bf16 MFMA operands (opt-in mixed precision)
torch.manual_seed(124)
net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt).cuda()
crit = losses.LossNetUnshaded('cuda', 5, 6, 128, 16, opt).cuda()
optim, sched = train.make_optimizer(net)
g = torch.Generator(device='cuda').manual_seed(1)
inp = torch.rand(B, T, 5, 32, 32, device='cuda', generator=g); inp[:, :, 0] = inp[:, :, 0] * 2 - 1
flow = (torch.rand(B, T, 2, 32, 32, device='cuda', generator=g) - 0.5) * 0.05
tgt = torch.rand(B, T, 6, 128, 128, device='cuda', generator=g); tgt[:, :, 0] = tgt[:, :, 0] * 2 - 1
for _ in range(2):
    l = train.train_step(net, crit, optim, (inp, flow, tgt), initial_image="zero")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    l = train.train_step(net, crit, optim, (inp, flow, tgt), initial_image="zero")
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
flops = 13.4e9 * B * T
graphed = "graph" in sys.argv
if graphed:
    torch.manual_seed(124)
    net2 = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt).cuda()
    optim2, _ = train.make_optimizer(net2, capturable=True)
    t0 = time.perf_counter()
    step = train.GraphedTrainStep(net2, crit, optim2, (inp, flow, tgt), initial_image="zero")
    torch.cuda.synchronize(); t_cap = time.perf_counter() - t0
    for _ in range(2): lg = step((inp, flow, tgt))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): lg = step((inp, flow, tgt))
    torch.cuda.synchronize(); dtg = (time.perf_counter() - t0) / steps
    print("B=%d T=%d HIP graph: %.1f ms/step, %.2f clips/s, %.1f TFLOP/s, loss %.4f (capture %.1f s)" % (B, T, dtg * 1e3, B / dtg, flops / dtg / 1e12, float(lg), t_cap))
print("B=%d T=%d: %.1f ms/step, %.2f clips/s, %.1f TFLOP/s (conv fwd+dgrad+wgrad algorithmic), loss %.4f" % (B, T, dt * 1e3, B / dt, flops / dt / 1e12, l))
