"""The warp's backward (isrRecurrentInputBackward: zero fill + atomic scatter + clamp / normalise derivative) alone (the loop is bound by
Python: read the kernel times with rocprofv3 --kernel-trace --stats), at the
training bench's shapes (16 clips x 32^2 -> 128^2)."""
import sys
sys.path.insert(0, '.')
import torch
from isosurfacesuperresolution_amd import ops

g = torch.Generator().manual_seed(0)
B, h, w = 16, 32, 32
prev = (torch.rand(B, 6, 4 * h, 4 * w, generator=g) * 2 - 1).cuda().requires_grad_(True)
inp = torch.rand(B, 5, h, w, generator=g).cuda()
flow = ((torch.rand(B, 2, h, w, generator=g) - 0.5) * 0.05).cuda()
netin, warped = ops.recurrent_input(prev, inp, flow)
gn, gw = torch.rand_like(netin), torch.rand_like(warped)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    torch.autograd.grad((netin, warped), prev, (gn, gw), retain_graph=True)
torch.cuda.synchronize()
e0.record()
for _ in range(20):
    torch.autograd.grad((netin, warped), prev, (gn, gw), retain_graph=True)
e1.record(); torch.cuda.synchronize()
print("recurrent input backward: %.1f us per call" % (e0.elapsed_time(e1) / 20 * 1e3))
