"""Per-launch time of the small-image split kernel (conv3x3_split_rows2_kernel) in a HIP graph of 100 dependent layers, for batches of
32 x 32 crops: is a launch bound by the weights every workgroup pulls from L2 (then fewer, larger workgroups help) or by its chain of
latencies?  Measured (round 3): 11.4-12.0 us for 64, 128 or 256 workgroups alike, 22.6 for 512 (two rounds), the same with a residual
tensor -- a chain of latencies.  Requesting the second k-step pair's weights and the residual at the kernel's start (more loads in
flight at once, 320 registers) made a launch 0.75 us SLOWER (12.76 against 11.98 us, same box, interleaved); not kept.
PYTHONPATH=. python tools/lab/bench_rows.py"""
import torch
from isosurfacesuperresolution_amd import ops

wt = (torch.rand(64, 64, 3, 3, device='cuda') - 0.5) * 0.1
b = torch.rand(64, device='cuda')
ops.RANGE_GUARD = False
with torch.no_grad():
    for n, h, w in ((16, 32, 32), (8, 32, 32), (4, 32, 32), (32, 32, 32), (16, 16, 32), (16, 64, 32)):
        x = torch.rand(n, 64, h, w, device='cuda') - 0.5
        for _ in range(3):
            y = ops.conv3x3(x, wt, b, act='relu')
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            g = torch.cuda.CUDAGraph()
            y = ops.conv3x3(x, wt, b, act='relu')
            torch.cuda.synchronize()
            with ops.graph_capture(g, stream=side):
                y = x
                for _ in range(100):
                    y = ops.conv3x3(y, wt, b, act='relu')
        torch.cuda.synchronize()
        ops.profile_enable(True)
        ops.conv3x3(x, wt, b, act='relu')
        torch.cuda.synchronize()
        name = ops.profile_records()[-1][0]
        ops.profile_enable(False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            g.replay()
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        tiles2 = n * (h // 2) * ((w + 31) // 32)
        print("%2d x %dx%d: %s, %4d two-row tiles, %.2f us per layer in a graph" % (n, w, h, name, tiles2, e0.elapsed_time(e1) / 1000 * 1e3))
    # with a residual / gate tensor (the second convolution of a block and every gated data gradient)
    for n, h, w in ((16, 32, 32),):
        x = torch.rand(n, 64, h, w, device='cuda') - 0.5
        r = torch.rand(n, 64, h, w, device='cuda') - 0.5
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            g = torch.cuda.CUDAGraph()
            y = ops.conv3x3(x, wt, b, residual=r)
            torch.cuda.synchronize()
            with ops.graph_capture(g, stream=side):
                y = x
                for _ in range(100):
                    y = ops.conv3x3(y, wt, b, residual=r)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            g.replay()
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        print("%2d x %dx%d with a residual tensor: %.2f us per layer in a graph" % (n, w, h, e0.elapsed_time(e1) / 1000 * 1e3))
