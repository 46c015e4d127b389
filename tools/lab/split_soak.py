"""Keeps one split-operand 1080p layer running for a few seconds (for rocm-smi power / clock sampling next to it) and
prints ablation timings per kernel form.  python tools/lab/split_soak.py [algo] [seconds] [dbg bits]"""
import ctypes, sys, time
import torch
sys.path.insert(0, '.')
from isosurfacesuperresolution_amd import ops
algo = int(sys.argv[1]) if len(sys.argv) > 1 else 1
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
lib = ops._sr()
lib.isrDebugSetSplitAlgo.argtypes = [ctypes.c_int]
lib.isrDebugSetSplitAblation.argtypes = [ctypes.c_int]
x = torch.rand(1, 64, 1080, 1920, device='cuda') - 0.5
wt = (torch.rand(64, 64, 3, 3, device='cuda') - 0.5) * 0.1
b = torch.rand(64, device='cuda')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(n=10):
    ops.conv3x3_split(x, wt, b, act='relu'); torch.cuda.synchronize()
    e0.record()
    for _ in range(n): ops.conv3x3_split(x, wt, b, act='relu')
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
with torch.no_grad():
    if secs > 0:
        lib.isrDebugSetSplitAlgo(algo)
        lib.isrDebugSetSplitAblation(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
        t0 = time.time(); n = 0
        while time.time() - t0 < secs:
            for _ in range(50): ops.conv3x3_split(x, wt, b, act='relu')
            torch.cuda.synchronize(); n += 50
        print("algo %d: %.3f ms per launch over %.1f s" % (algo, (time.time() - t0) / n * 1e3, secs))
    else:
        for a, name in ((0, "tile"), (1, "stream"), (2, "wide")):
            lib.isrDebugSetSplitAlgo(a)
            row = []
            for bits, label in ((0, "full"), (1, "no MFMAs"), (2, "no loads"), (4, "no stores"), (6, "MFMAs only"), (5, "loads only"), (3, "stores only")):
                lib.isrDebugSetSplitAblation(bits)
                row.append("%s %.3f" % (label, timed()))
            lib.isrDebugSetSplitAblation(0)
            print("%-6s %s" % (name, " | ".join(row)), flush=True)
