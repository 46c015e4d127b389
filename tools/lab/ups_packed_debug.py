"""Which form's packed output differs, and where (diagnostics build): ISR_SR_DIAG=1 PYTHONPATH=. python tools/lab/ups_packed_debug.py"""
import ctypes
import torch
import torch.nn.functional as F
from isosurfacesuperresolution_amd import ops
lib = ops._sr()
lib.isrDebugSetSplitUpsForm.argtypes = [ctypes.c_int]
h, w, cin = 4, 16, 64
g = torch.Generator().manual_seed(h * 1000 + w)
x = ((torch.rand(1, cin, h, w, generator=g) - 0.4) * 3).cuda()
wt = ((torch.rand(64, cin, 3, 3, generator=g) - 0.5) * 0.2).cuda()
b = ((torch.rand(64, generator=g) - 0.5) * 0.3).cuda()
ref = F.relu(F.conv2d(F.interpolate(x.double(), scale_factor=2, mode='bilinear', align_corners=False), wt.double(), b.double(), padding=1))
outs = {}
with torch.no_grad():
    for form in (0, 3, 4, 5, 7, 8):
        lib.isrDebugSetSplitUpsForm(form)
        ps = ops.conv3x3_split_packed(x, wt, b, act='relu', upsample2x=True)
        raw = ps.data.view(2 * (ps.channels // 8), ps.plane, 4)[:, :ps.h * ps.w].clone()
        outs[form] = raw
        print("form %d: max err vs fp64 %.3e" % (form, (ps.to_float().double() - ref).abs().max().item()))
lib.isrDebugSetSplitUpsForm(3)
base = outs[3]
for form, raw in outs.items():
    d = (raw != base)
    print("form %d vs 3: %d differing dwords of %d" % (form, int(d.sum()), d.numel()))
    if d.any():
        idx = d.nonzero()[:8]
        for i in idx:
            a, c = int(raw[tuple(i)]) & 0xffffffff, int(base[tuple(i)]) & 0xffffffff
            print("   at", i.tolist(), "%08x vs %08x" % (a, c))
