"""Where a dataflow-trunk launch spends its time: per-tile, per-layer phase durations stamped by the kernel itself
(isrDebugSetTrunkStampBuffer: s_memrealtime, one 100 MHz clock for the whole chip).
usage: PYTHONPATH=. python tools/lab/trunk_timeline.py"""
import ctypes

import numpy as np
import torch

from isosurfacesuperresolution_amd import ops

H, W = 270, 480
g = torch.Generator().manual_seed(0)
x = (torch.rand(1, 101, H, W, generator=g) - 0.3).cuda()
convs = [(((torch.rand(64, 101 if k == 0 else 64, 3, 3, generator=g) - 0.5) * 0.1).cuda(), ((torch.rand(64, generator=g) - 0.5) * 0.1).cuda())
         for k in range(21)]
lib = ops._sr()
tiles = ((H + 15) // 16) * ((W + 31) // 32)
L = 21
stamps = torch.zeros(tiles * L * 8, dtype=torch.int64, device="cuda")
with torch.no_grad():
    for _ in range(10):
        ops.trunk_dataflow(x, convs)
    torch.cuda.synchronize()
    lib.isrDebugSetTrunkStampBuffer(ctypes.c_void_p(stamps.data_ptr()))
    ops.trunk_dataflow(x, convs)
    torch.cuda.synchronize()
    lib.isrDebugSetTrunkStampBuffer(None)
s = stamps.cpu().numpy().reshape(tiles, L, 8).astype(np.float64)
t0 = s[:, :, 0].min()
s[:, 0, 1] = s[:, 0, 0]                             # the first layer waits for nobody
start = (s[:, :, 0] - t0) / 100.0                  # us
ph = np.diff(s[:, :, 0:6], axis=2) / 100.0          # us: wait, stage, mfma, epilogue, drain
end = (s[:, :, 5] - t0) / 100.0
print("launch span %.0f us (first layer start -> last layer end)" % end.max())
names = ["wait", "stage", "mfma", "epilogue", "drain"]
print("mean per tile over the launch: " + ", ".join("%s %.0f us" % (n, ph[:, :, i].sum(axis=1).mean()) for i, n in enumerate(names)),
      "| total %.0f" % ph.sum(axis=(1, 2)).mean())
print("layer:  start(min..max)   " + "  ".join("%8s" % n for n in names) + "   (mean us per tile)")
for l in range(L):
    print("  %2d   %7.1f..%7.1f   " % (l, start[:, l].min(), start[:, l].max()) + "  ".join("%8.2f" % ph[:, l, i].mean() for i in range(5)))
