"""The 20 block layers at 480x270 back to back: plain stream launches vs one HIP graph replay (does a graph shorten the
gaps between dependent kernels?)."""
import sys
sys.path.insert(0, '.')
import torch
from isosurfacesuperresolution_amd import ops

e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.manual_seed(0)
x = torch.rand(1, 64, 270, 480, device='cuda') - 0.5
ws = [((torch.rand(64, 64, 3, 3, device='cuda') - 0.5) * 0.05) for _ in range(20)]
bs = [torch.rand(64, device='cuda') * 0.1 for _ in range(20)]


def trunk():
    f = x
    for k in range(0, 20, 2):
        f = ops.residual_block(f, ws[k], bs[k], ws[k + 1], bs[k + 1])
    return f


with torch.no_grad():
    for _ in range(3):
        trunk()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        trunk()
    e1.record(); torch.cuda.synchronize()
    t_stream = e0.elapsed_time(e1) / 20
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        trunk()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with ops.graph_capture(g):
        out = trunk()
    g.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    t_graph = e0.elapsed_time(e1) / 20
print("20 layers at 480x270: stream %.3f ms (%.1f us per layer), graph %.3f ms (%.1f us per layer)" % (t_stream, t_stream * 50, t_graph, t_graph * 50))
