"""Experiment: the 480 x 270 trunk (pre-block conv + 10 residual blocks) as ONE image vs as S horizontal strips (with the
halo the trunk's receptive field needs) on S HIP streams running concurrently -- do the strips' layers, being out of phase
with each other, overlap their memory and matrix phases?   usage: PYTHONPATH=. python tools/bench_trunk_strips.py"""
import argparse
import sys

import torch

from isosurfacesuperresolution_amd import models, ops

opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10)
torch.manual_seed(0)
net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt).cuda().eval()
x = torch.rand(1, 101, 270, 480, device="cuda")
HALO = 24


def trunk(inp):
    pre = net.preblock[0]
    f = ops.conv3x3(inp, pre.weight, pre.bias, act='relu')
    for b in net.blocks:
        f = ops.residual_block(f, b[0].weight, b[0].bias, b[2].weight, b[2].bias)
    return f


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    for fused in (False, True):
        ops.BLOCK_FUSION = fused
        ops.BLOCK_FUSION_MIN_TILES = 64
        full = trunk(x)
        t_full = timed(lambda: trunk(x))
        print("block fusion %s: whole image %.3f ms" % (fused, t_full))
        for S in (2, 3, 4):
            streams = [torch.cuda.Stream() for _ in range(S)]
            bounds = [(k * 270 // S, (k + 1) * 270 // S) for k in range(S)]
            ext = [(max(0, a - HALO), min(270, b + HALO)) for a, b in bounds]
            parts = [x[:, :, a:b].contiguous() for a, b in ext]
            outs = [None] * S

            def run():
                cur = torch.cuda.current_stream()
                for k, st in enumerate(streams):
                    st.wait_stream(cur)
                    with torch.cuda.stream(st):
                        outs[k] = trunk(parts[k])
                for st in streams:
                    cur.wait_stream(st)
            run()
            torch.cuda.synchronize()
            got = torch.cat([o[:, :, a - ea:o.shape[2] - (eb - b)] for o, (a, b), (ea, eb) in zip(outs, bounds, ext)], dim=2)
            same = torch.equal(got, full)
            t = timed(run)
            # the same strips one after the other on ONE stream (what the extra halo work costs without any overlap)
            t_serial = timed(lambda: [trunk(p) for p in parts])
            print("   %d strips on %d streams: %.3f ms (bit-identical interior: %s); serial on one stream %.3f ms" % (S, S, t, same, t_serial))
