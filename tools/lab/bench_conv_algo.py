"""Compares the two forward conv kernels (one vs two workgroups per CU): max difference and time per shape."""
import sys, torch
sys.path.insert(0, '.')
from isosurfacesuperresolution_amd import ops
lib = ops._sr()
shapes = [(64, 64, 270, 480, False, True), (64, 64, 1080, 1920, False, False), (64, 64, 540, 960, True, False),
          (64, 64, 1080, 1920, True, False), (101, 64, 270, 480, False, False), (5, 32, 37, 50, False, True), (64, 96, 64, 64, True, True)]
with torch.no_grad():
    for cin, cout, h, w, ups, res in shapes:
        hin, win = (h // 2, w // 2) if ups else (h, w)
        x = torch.rand(1, cin, hin, win, device='cuda') - 0.5
        wt = (torch.rand(cout, cin, 3, 3, device='cuda') - 0.5) * 0.1
        b = torch.rand(cout, device='cuda')
        r = torch.rand(1, cout, h, w, device='cuda') if res else None
        outs, line = [], []
        for algo in (0, 1):
            lib.isrDebugSetForwardAlgo(algo)
            for _ in range(3): y = ops.conv3x3(x, wt, b, act='relu', residual=r, upsample2x=ups)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10
            e0.record()
            for _ in range(n): y = ops.conv3x3(x, wt, b, act='relu', residual=r, upsample2x=ups)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            fl = 2.0 * 9 * cin * cout * h * w
            outs.append(y.clone())
            line.append("algo%d %.1f us %.1f TF" % (algo, ms * 1e3, fl / ms / 1e9))
        lib.isrDebugSetForwardAlgo(0)
        print("%dx%d %d->%d%s%s: " % (w, h, cin, cout, " ups" if ups else "", " +res" if res else "") + " | ".join(line)
              + " | max diff %.3g" % (outs[0] - outs[1]).abs().max().item(), flush=True)
