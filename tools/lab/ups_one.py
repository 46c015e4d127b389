"""Ten launches of the 1080p upsampling layer (packed-split output) for profilers.  PYTHONPATH=. python3 tools/lab/ups_one.py"""
import torch
from isosurfacesuperresolution_amd import ops
x = torch.rand(1, 64, 540, 960, device='cuda') - 0.5
wt = (torch.rand(64, 64, 3, 3, device='cuda') - 0.5) * 0.1
b = torch.rand(64, device='cuda')
with torch.no_grad():
    for _ in range(10):
        ops.conv3x3_split_packed(x, wt, b, act='relu', upsample2x=True)
torch.cuda.synchronize()
