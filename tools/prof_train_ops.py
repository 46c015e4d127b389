"""Which PyTorch ops are still launched by one eager training step (the HIP kernels of this package show up as their
autograd Function names): python tools/prof_train_ops.py"""
import argparse, contextlib, sys
sys.path.insert(0, '.')
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from isosurfacesuperresolution_amd import losses, models, train

opt = argparse.Namespace(**bench.TRAIN_OPT)
torch.manual_seed(124)
with contextlib.redirect_stdout(sys.stderr):
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt).cuda()
    crit = losses.LossNetUnshaded("cuda", 5, 6, 128, 16, opt).cuda()
optim, _ = train.make_optimizer(net, capturable=True)
batch = bench._clip_batch(torch, 16, 10, 32, 1000, "cuda")
for _ in range(2):
    train.train_step(net, crit, optim, batch, initial_image="zero")
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU]) as prof:
    train.train_step(net, crit, optim, batch, initial_image="zero")
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.count)
for e in rows[:45]:
    print("%6d  %s" % (e.count, e.key))
