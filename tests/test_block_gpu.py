"""The fused residual block (csrc/sr_conv_block.hip) against the two split-operand launches it replaces: the same products in
the same order, so the outputs must be EQUAL bit for bit (SuperresolutionNetwork/models/enhancenet.py:18-33,139-141)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(h, w, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = ((torch.rand(1, 64, h, w, generator=g) - 0.3) * scale).cuda()
    w1 = ((torch.rand(64, 64, 3, 3, generator=g) - 0.5) * 0.1).cuda()
    w2 = ((torch.rand(64, 64, 3, 3, generator=g) - 0.5) * 0.1).cuda()
    b1 = ((torch.rand(64, generator=g) - 0.5) * 0.2).cuda()
    b2 = ((torch.rand(64, generator=g) - 0.5) * 0.2).cuda()
    return x, w1, b1, w2, b2


@pytest.mark.parametrize("h,w", [(8, 32), (9, 36), (23, 52), (64, 64), (270, 480), (135, 240)])
def test_fused_block_is_bit_identical_to_two_launches(h, w):
    from isosurfacesuperresolution_amd import ops
    x, w1, b1, w2, b2 = _case(h, w, seed=h * 7 + w)
    with torch.no_grad():
        y_f = ops.residual_block_fused(x, w1, b1, w2, b2)
        y_r = ops.conv3x3_split(ops.conv3x3_split(x, w1, b1, act='relu'), w2, b2, residual=x)
        y_f2 = ops.residual_block_fused(x, w1, b1, w2, b2)
    torch.cuda.synchronize()
    assert y_f.shape == y_r.shape
    assert torch.equal(y_f, y_r), (y_f - y_r).abs().max().item()
    assert torch.equal(y_f, y_f2)


@pytest.mark.parametrize("h,w", [(8, 32), (9, 36), (23, 52), (270, 480)])
def test_packed_split_intermediate_is_bit_identical(h, w):
    """The two launches of a block with the intermediate handed over packed-split (conv1 stores (hi, lo') units from its
    accumulators, conv2 stages them by LDS-DMA) compute exactly what they compute through an fp32 tensor."""
    from isosurfacesuperresolution_amd import ops
    x, w1, b1, w2, b2 = _case(h, w, seed=h * 5 + w)
    with torch.no_grad():
        t = ops.conv3x3_split(x, w1, b1, act='relu')
        tp = ops.conv3x3_split_packed(x, w1, b1, act='relu')
        assert (tp.to_float() - t).abs().max().item() <= 2.0 ** -21 * max(1.0, t.abs().max().item())
        y_r = ops.conv3x3_split(t, w2, b2, residual=x)
        y_p = ops.conv3x3_split_from_packed(tp, w2, b2, residual=x)
        # packed in, packed out (no residual): the units are the split of the fp32 result
        u = ops.conv3x3_split(t, w2, b2, act='relu')
        up = ops.conv3x3_split_from_packed(tp, w2, b2, act='relu', packed_out=True)
    torch.cuda.synchronize()
    assert torch.equal(y_p, y_r), (y_p - y_r).abs().max().item()
    assert (up.to_float() - u).abs().max().item() <= 2.0 ** -21 * max(1.0, u.abs().max().item())


def test_fused_block_without_bias_and_with_padded_planes():
    from isosurfacesuperresolution_amd import ops
    h, w = 40, 96
    x, w1, b1, w2, b2 = _case(h, w, seed=3, scale=30.0)
    xp = torch.zeros(64 * (h * w + 4 * w), device="cuda").as_strided((1, 64, h, w), (64 * (h * w + 4 * w), h * w + 4 * w, w, 1))
    xp.copy_(x)
    with torch.no_grad():
        y_a = ops.residual_block_fused(x, w1, None, w2, None)
        y_b = ops.residual_block_fused(xp, w1, None, w2, None)
        y_r = ops.conv3x3_split(ops.conv3x3_split(x, w1, None, act='relu'), w2, None, residual=x)
    assert torch.equal(y_a, y_r) and torch.equal(y_b, y_r)


def test_trunk_runs_on_the_fused_block_and_matches():
    """``ops.residual_block`` takes the fused kernel for full-size single images (no autograd) and the network's features
    are unchanged."""
    import argparse
    from isosurfacesuperresolution_amd import models, ops
    opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10)
    torch.manual_seed(1)
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt).cuda().eval()
    x = torch.rand(1, 101, 270, 480, device="cuda")
    outs = {}
    for fused in (True, False, "plain", "dataflow"):
        ops.BLOCK_FUSION = fused is True
        ops.BLOCK_PACKED = fused != "plain"
        ops.TRUNK_DATAFLOW = fused == "dataflow"      # (the default: the whole trunk in one launch, tests/test_trunk_gpu.py)
        try:
            ops.profile_enable(True)
            with torch.no_grad():
                outs[fused] = net.forward_features(x, last_layer=False)
            torch.cuda.synchronize()
            names = [n for n, _, _ in ops.profile_records()]
            ops.profile_enable(False)
        finally:
            ops.BLOCK_FUSION = False
            ops.BLOCK_PACKED = True
            ops.TRUNK_DATAFLOW = True
        assert (names.count("resblock_split_kernel") == 10) == (fused is True)
        assert (names.count("trunk_dataflow_kernel") == 1) == (fused == "dataflow")
    assert torch.equal(outs[True], outs[False]) and torch.equal(outs[False], outs["plain"]) and torch.equal(outs["plain"], outs["dataflow"])
