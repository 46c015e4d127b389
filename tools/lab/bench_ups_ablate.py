"""The 1080p upsampling layer (540 x 960 x 64 -> 1080 x 1920 x 64, packed-split output, conv3x3_split_ups3_kernel) with parts switched off
(isrDebugSetSplitAblation: results wrong, time only): what each part costs where it runs.  PYTHONPATH=. python tools/lab/bench_ups_ablate.py"""
import torch
from isosurfacesuperresolution_amd import ops
lib = ops._sr()
x = torch.rand(1, 64, 540, 960, device='cuda') - 0.5
wt = (torch.rand(64, 64, 3, 3, device='cuda') - 0.5) * 0.1
b = torch.rand(64, device='cuda')


def timed(n=20):
    for _ in range(3):
        ops.conv3x3_split_packed(x, wt, b, act='relu', upsample2x=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.conv3x3_split_packed(x, wt, b, act='relu', upsample2x=True)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


names = {0: "everything", 1: "no MFMAs", 2: "no interpolation (blend + split of the staging)", 8: "no epilogue", 16: "no epilogue stores",
         3: "no MFMAs, no interpolation", 9: "no MFMAs, no epilogue", 10: "no interpolation, no epilogue (loads, parks, barriers, MFMAs)",
         11: "loads, parks and barriers only", 0.5: "everything (again)"}
with torch.no_grad():
    for bits, name in names.items():
        lib.isrDebugSetSplitAblation(int(bits))
        print("%-70s %6.0f us" % (name, timed()))
    lib.isrDebugSetSplitAblation(0)
