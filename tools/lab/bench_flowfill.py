"""Flow hole filling alone: one launch (isrFlowFillOne) against three (isrFlowFillEx), 480 x 270 and 960 x 540, time per call from events.
usage: PYTHONPATH=. python tools/lab/bench_flowfill.py"""
import sys
import torch
sys.path.insert(0, "tests")
from test_flowfill_gpu import _gbuffer
from isosurfacesuperresolution_amd import ops

for h, w in ((270, 480), (540, 960)):
    gb = _gbuffer(h, w, 1, "blobs")
    for one in (False, True, False, True):
        for _ in range(5):
            ops.fill_flow_gbuffer(gb, one_launch=one)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ops.fill_flow_gbuffer(gb, one_launch=one)
        e1.record()
        torch.cuda.synchronize()
        print("%dx%d %s: %.1f us per fill" % (w, h, "one launch" if one else "three launches", e0.elapsed_time(e1) / 50 * 1e3))
