import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from isosurfacesuperresolution_amd import ops
torch.manual_seed(0)
x = torch.zeros(1, 64, 16, 32); w = torch.zeros(64, 64, 3, 3); b = torch.zeros(64)
res = torch.arange(64 * 16 * 32, dtype=torch.float32).view(1, 64, 16, 32)
with torch.no_grad():
    y = ops.conv3x3(x.cuda(), w.cuda(), b.cuda(), residual=res.cuda()).cpu()
d = (y - res)
print('max err', d.abs().max().item())
bad = (d.abs() > 0).nonzero()
print('num bad', len(bad), 'of', y.numel())
print(bad[:10].tolist())
print('y[0,0,0,:8]', y[0, 0, 0, :8].tolist(), 'y[0,1,0,:8]', y[0, 1, 0, :8].tolist())
