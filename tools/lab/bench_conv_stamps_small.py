"""In-kernel stamps of the forward conv at the training shape (16 images of 32x32, 4x32-pixel tiles): where the 18 us of
a launch go when every CU has exactly one workgroup.  Also times back-to-back launches with HIP events."""
import sys, ctypes, torch
import numpy as np
sys.path.insert(0, '.')
from isosurfacesuperresolution_amd import ops
lib = ops._sr()
N, cin, cout, h, w = 16, 64, 64, 32, 32
if len(sys.argv) > 1:
    N, h, w = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
with torch.no_grad():
    nwg = N * h * ((w + 31) // 32) * 2
    buf = torch.zeros(nwg * 4, dtype=torch.int64, device='cuda')
    x = torch.rand(N, cin, h, w, device='cuda') - 0.5
    wt = (torch.rand(cout, cin, 3, 3, device='cuda') - 0.5) * 0.1
    b = torch.rand(cout, device='cuda')
    for _ in range(3): y = ops.conv3x3(x, wt, b, act='relu')
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for tile, name in ((1, "4x32 tiles"), (3, "row split"), (2, "16x32 tiles")):
        lib.isrDebugSetForwardTile(tile)
        for _ in range(3): y = ops.conv3x3(x, wt, b, act='relu')
        e0.record()
        for _ in range(200): y = ops.conv3x3(y, wt, b, act='relu')
        e1.record(); torch.cuda.synchronize()
        eager = e0.elapsed_time(e1) * 1e3 / 200
        # the same chain as a HIP graph: the eager loop is bound by the ~12 us of Python + ctypes per call
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            y = ops.conv3x3(x, wt, b, act='relu')
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with ops.graph_capture(graph):
            y = x
            for _ in range(200): y = ops.conv3x3(y, wt, b, act='relu')
        graph.replay(); torch.cuda.synchronize()
        e0.record(); graph.replay(); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 200
        print("N=%d %dx%d %-11s: %.2f us per dependent launch eager, %.2f us in a HIP graph (%.1f TFLOP/s)" % (N, w, h, name, eager, us,
              2 * 9 * cin * cout * N * h * w / (us * 1e-6) / 1e12))
    lib.isrDebugSetForwardTile(int(sys.argv[4]) if len(sys.argv) > 4 else 1)
    lib.isrDebugSetStampBuffer(ctypes.c_void_p(buf.data_ptr()))
    lib.isrDebugSetAblation(8)
    for _ in range(2): ops.conv3x3(x, wt, b, act='relu')
    torch.cuda.synchronize()
    st = buf.cpu().numpy().reshape(-1, 4).astype(np.float64)
    st = st[st[:, 0] > 0]
    d = np.diff(st, axis=1)
    print("workgroups stamped %d: ticks (10 ns) prologue %.0f main %.0f epilogue %.0f life %.0f" % (
        len(st), np.median(d[:, 0]), np.median(d[:, 1]), np.median(d[:, 2]), np.median(st[:, 3] - st[:, 0])))
    s8 = st[0::8]
    print("one XCD: start skew pct [50,100] %s, span %.0f ticks" % (np.percentile(s8[:, 0] - s8[:, 0].min(), [50, 100]).round(0), s8[:, 3].max() - s8[:, 0].min()))
    lib.isrDebugSetAblation(0)
    lib.isrDebugSetStampBuffer(None)
    lib.isrDebugSetForwardTile(0)
