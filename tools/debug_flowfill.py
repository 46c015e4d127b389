import sys
import torch
sys.path.insert(0, "tests")
from test_flowfill_gpu import _gbuffer
from isosurfacesuperresolution_amd import ops
for (h, w, d) in ((270, 480, "blobs"), (64, 64, "blobs"), (37, 53, 0.5), (128, 128, "one"), (65, 129, 0.02)):
    gb = _gbuffer(h, w, 7 * h + w, d)
    three = ops.fill_flow_gbuffer(gb, one_launch=False)
    one = ops.fill_flow_gbuffer(gb, one_launch=True)
    torch.cuda.synchronize()
    diff = (one - three).abs()
    bad = (diff > 0).nonzero()
    print(h, w, d, "mismatches", bad.shape[0], "of", one.numel(), "max", diff.max().item(), "nan", int(torch.isnan(one).sum()))
    if bad.shape[0]:
        ys, xs = bad[:, 2], bad[:, 3]
        print("  y range", ys.min().item(), ys.max().item(), "x range", xs.min().item(), xs.max().item())
        print("  first", bad[:6].tolist(), [(one[tuple(b)].item(), three[tuple(b)].item()) for b in bad[:4]])
        hist = torch.histc(diff[diff > 0].log10().float(), bins=8, min=-9, max=-1)
        print("  log10 diff histogram (-9..-1):", hist.tolist())
        print("  mismatch x mod 64 histogram:", torch.bincount(xs % 64, minlength=64).tolist())
