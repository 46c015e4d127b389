"""The fp16 fast mode of the fused convolution (csrc/sr_conv_f16.hip, through the C-ABI).  It is NOT the parity path
(reported separately, judged by PSNR); what is checked here is that it computes the same operator:
exactly, when the inputs are representable in fp16 (every product is then exact in fp32 and only the summation order
differs from the fp64 reference), and to fp16 rounding otherwise."""
import argparse

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # N, Cin, Cout, h, w, act, bias, residual, upsample
    (1, 64, 64, 16, 32, 'relu', True, False, False),
    (1, 64, 64, 37, 45, 'none', True, True, False),       # ragged tile edges + fused skip
    (2, 101, 64, 20, 33, 'relu', True, False, False),     # two staging passes, the second one partly empty
    (1, 64, 64, 18, 22, 'relu', True, False, True),       # x2 upsampling as its own kernel in front (rows not 16-byte aligned)
    (1, 64, 64, 20, 24, 'relu', True, False, True),       # x2 upsampling fused into the staging
    (2, 64, 64, 16, 32, 'none', True, True, True),
    (1, 64, 64, 67, 120, 'relu', True, False, True),      # ragged tiles, several XCD ranges
    (3, 5, 7, 9, 11, 'leaky', False, False, False),
    (1, 64, 96, 20, 40, 'none', True, False, False),      # three 32-channel blocks: the last group has one
    (1, 64, 64, 1, 1, 'relu', True, False, False),
    (1, 16, 32, 270, 480, 'none', False, False, False),   # many tiles
]


def _ref(x, w, b, act, slope, res, ups):
    x = x.double(); w = w.double()
    if ups:
        x = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)
    y = F.conv2d(x, w, b.double() if b is not None else None, padding=1)
    if act == 'relu':
        y = F.relu(y)
    elif act == 'leaky':
        y = F.leaky_relu(y, slope)
    return y + res.double() if res is not None else y


@pytest.mark.parametrize("case", CASES)
def test_f16_conv_is_the_same_operator(case):
    from isosurfacesuperresolution_amd import ops
    n, cin, cout, h, w, act, has_b, has_r, ups = case
    g = torch.Generator().manual_seed(cin * 7 + h)
    x = torch.rand(n, cin, h, w, generator=g) * 2 - 1
    wt = (torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) / (3.0 * cin ** 0.5)
    b = torch.rand(cout, generator=g) - 0.5 if has_b else None
    H, W = (2 * h, 2 * w) if ups else (h, w)
    res = torch.rand(n, cout, H, W, generator=g) - 0.5 if has_r else None
    if not ups:                                  # inputs representable in fp16: products exact, fp32 sums
        xq, wq = x.half().float(), wt.half().float()
        y = ops.conv3x3_f16(xq.cuda(), wq.cuda(), b.cuda() if has_b else None, act, 0.1, res.cuda() if has_r else None)
        ref = _ref(xq, wq, b, act, 0.1, res, False)
        assert y.shape == ref.shape
        assert (y.cpu().double() - ref).abs().max().item() <= 1e-5
    else:                                        # the fused resize against the separate kernel + the same convolution
        up = ops.bilinear_upsample2x(x.cuda())
        y1 = ops.conv3x3_f16(x.cuda(), wt.cuda(), b.cuda() if has_b else None, act, 0.1, res.cuda() if has_r else None, upsample2x=True)
        y2 = ops.conv3x3_f16(up, wt.cuda(), b.cuda() if has_b else None, act, 0.1, res.cuda() if has_r else None)
        # same blend; an FMA contracted differently flips single fp16 roundings (2^-11 relative) of the operands
        assert (y1 - y2).abs().max().item() <= 1e-3
    # arbitrary fp32 inputs: fp16 rounding of both operands, ~2^-12 relative per product, averaging out over the sum
    y = ops.conv3x3_f16(x.cuda(), wt.cuda(), b.cuda() if has_b else None, act, 0.1, res.cuda() if has_r else None, upsample2x=ups)
    ref = _ref(x, wt, b, act, 0.1, res, ups)
    assert y.shape == ref.shape
    assert (y.cpu().double() - ref).abs().max().item() <= 2e-3


def test_f16_network_psnr_against_fp32():
    """EnhanceNet forward with FAST_F16 against the fp32 kernels on the same input: a PSNR, not a parity claim."""
    from isosurfacesuperresolution_amd import models, ops
    opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10)
    torch.manual_seed(0)
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt).cuda().eval()
    x = torch.rand(1, 101, 40, 56, generator=torch.Generator().manual_seed(4)).cuda()
    with torch.no_grad():
        ref, _ = net(x)
        ops.FAST_F16 = True
        try:
            fast, _ = net(x)
        finally:
            ops.FAST_F16 = False
        again, _ = net(x)
    assert torch.equal(again, ref)                                  # the switch leaves the fp32 path untouched
    mse = ((fast - ref) ** 2).mean().item()
    peak = ref.abs().max().item()
    psnr = 10 * torch.log10(torch.tensor(peak * peak / mse)).item()
    assert psnr >= 55.0, psnr
