import sys, torch, numpy as np
sys.path.insert(0, '.')
import torch.nn.functional as F
from isosurfacesuperresolution_amd import ops
g = torch.Generator().manual_seed(1)
for (N, Cin, Cout, h, w) in ((1, 64, 64, 16, 32), (1, 64, 64, 40, 64), (1, 32, 32, 8, 32)):
    x = torch.rand(N, Cin, h, w, generator=g) * 2 - 1
    wt = (torch.rand(Cout, Cin, 3, 3, generator=g) * 2 - 1) / (3.0 * Cin ** 0.5)
    ref = F.conv2d(x.double(), wt.double(), padding=1)
    with torch.no_grad():
        wq = ops._prepare_split(wt.cuda())
        torch.cuda.synchronize()
        hdr = wq[:16].cpu().numpy()
        print("header:", hdr.view(np.float32)[:2], hdr.view(np.int32)[2], "max|w|", wt.abs().max().item())
        h16 = wq[16:16 + 64].cpu().numpy().view(np.float16)
        print("first hi units:", h16[:16])
        y = ops.conv3x3_split(x.cuda(), wt.cuda()).cpu().double()
    d = (y - ref).abs()
    print((N, Cin, Cout, h, w), "max err", d.max().item(), "ref max", ref.abs().max().item(), "y max", y.abs().max().item(),
          "finite", torch.isfinite(y).all().item())
    idx = np.unravel_index(d.argmax().item(), d.shape)
    print(" worst at", idx, y[idx].item(), ref[idx].item())
    print(" err per channel (first 8):", d.amax(dim=(0, 2, 3))[:8].tolist())
    print(" err per row:", d.amax(dim=(0, 1, 3)).tolist()[:16])
    print(" err per col:", d.amax(dim=(0, 1, 2)).tolist()[:34])
