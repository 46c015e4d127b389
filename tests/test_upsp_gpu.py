"""The phase-decomposed upsampling convolution (csrc/sr_conv_upsp.h; SuperresolutionNetwork/models/enhancenet.py:113-124:
nn.Upsample(scale_factor=2, mode='bilinear') + Conv2d(64, 64, 3) + ReLU) against an fp64 convolution of the fp64-upsampled input:
the same distance as the interpolate-then-convolve kernels it replaces -- on every pixel, the one-pixel frame (its own exact kernel)
and the image corners included -- and the packed-split hand-over from the dataflow trunk through both layers to the fused tail."""
import argparse

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _diagnostics_build(diag_lib):
    """Every test of this module is about kernel forms that exist in the diagnostics build only (csrc/sr_diag.h)."""
    yield diag_lib
OPT = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10)


def _layer(seed, scale=0.06):
    g = torch.Generator().manual_seed(seed)
    w = ((torch.rand(64, 64, 3, 3, generator=g) - 0.5) * 2 * scale).cuda()
    b = ((torch.rand(64, generator=g) - 0.5) * 0.2).cuda()
    return w, b


def _ref64(x, w, b, act):
    u = F.interpolate(x.double().cpu(), scale_factor=2, mode='bilinear', align_corners=False)
    y = F.conv2d(u, w.double().cpu(), b.double().cpu(), padding=1)
    return torch.relu(y) if act == 'relu' else y


@pytest.mark.parametrize("h,w", [(270, 480), (135, 240), (17, 37), (8, 32), (9, 33), (1, 5), (40, 64)])
@pytest.mark.parametrize("act", ["relu", "none"])
def test_phase_decomposed_layer_against_fp64(h, w, act):
    from isosurfacesuperresolution_amd import ops
    wt, b = _layer(h * 31 + w)
    g = torch.Generator().manual_seed(h + w)
    x = ((torch.rand(1, 64, h, w, generator=g) - 0.3) * 2).cuda()
    xp = ops.pack_split(x)
    assert (xp.to_float() - x).abs().max().item() <= 2 ** -21 * 2                   # the packed pair carries 22 bits
    assert bool(ops._sr().isrConvUpsPhaseSupported(64, 64, h, w, xp.plane, 4 * h * w + ops.plane_pad(2 * h, 2 * w)))
    y = ops.conv3x3_ups_phase(xp, wt, b, act=act)
    old = ops.conv3x3_split(x, wt, b, act=act, upsample2x=True) if w % 4 == 0 else None
    torch.cuda.synchronize()
    assert (y.h, y.w, y.channels) == (2 * h, 2 * w, 64)
    ref = _ref64(xp.to_float(), wt, b, act)                                         # the fp64 function of the values the kernel was given
    got = y.to_float().double().cpu()
    err = (got - ref).abs()
    scale = ref.abs().max().item()
    assert err.max().item() <= 2e-6 * max(1.0, scale), (err.max().item(), scale, torch.nonzero(err == err.max())[0].tolist())
    # the frame (exact kernel) and the body separately, so that a failure says which
    frame = torch.zeros(2 * h, 2 * w, dtype=torch.bool)
    frame[0] = frame[-1] = True; frame[:, 0] = frame[:, -1] = True
    assert err[0][:, frame].max().item() <= 2e-6 * max(1.0, scale) and (2 * h <= 2 or 2 * w <= 2 or err[0][:, ~frame].max().item() <= 2e-6 * max(1.0, scale))
    if old is not None:
        # ... and no further from fp64 than the interpolating kernel is (twice its error + a floor)
        old_err = (old.double().cpu() - _ref64(x, wt, b, act)).abs().max().item()
        assert err.max().item() <= 2.0 * old_err + 1e-6 * max(1.0, scale), (err.max().item(), old_err)


def test_small_integer_data_come_out_exactly():
    """Products of small integers and sums of them are exact in every arithmetic involved (fp16 pairs, fp32 accumulation, the fp64
    effective weights are multiples of 1/16): a wrong tap, parity or border shows as an integer-sized error."""
    from isosurfacesuperresolution_amd import ops
    g = torch.Generator().manual_seed(3)
    wt = torch.randint(-3, 4, (64, 64, 3, 3), generator=g).float().cuda()
    b = torch.randint(-5, 6, (64,), generator=g).float().cuda()
    x = (torch.randint(-4, 5, (1, 64, 21, 45), generator=g).float() * 16).cuda()     # U2 of multiples of 16 are integers
    y = ops.conv3x3_ups_phase(ops.pack_split(x), wt, b, act='none')
    torch.cuda.synchronize()
    ref = _ref64(x, wt, b, 'none')
    assert torch.equal(y.to_float().double().cpu(), ref)


def test_packed_chain_trunk_to_tail_matches_the_per_layer_route():
    """The frame's network with the dataflow trunk's packed-split result feeding both phase-decomposed upsampling layers and the
    fused tail, against the same network on the interpolating kernels (ISR_UPS_PHASE off): 1e-4 is the parity tolerance; the two
    routes differ by roundings only."""
    from isosurfacesuperresolution_amd import models, ops
    from isosurfacesuperresolution_amd.inference import LoadedModel
    from isosurfacesuperresolution_amd.pipeline import default_shading, run_network
    torch.manual_seed(0)
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, OPT)
    lm = LoadedModel.from_model(net, "cuda", parameters={"initialImage": "zero"})
    sh = default_shading("cuda", 30.0)
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(1, 101, 136, 240, generator=g) * 2 - 0.5).cuda()
    x[:, 0] = (x[:, 0] > 0.3).float() * 2 - 1
    was = ops.UPS_PHASE
    ops.UPS_PHASE = True                                    # (opt-in: ISR_UPS_PHASE=1)
    try:
        ops.profile_enable(True)
        with torch.no_grad():
            raw_a, rgb_a = run_network(lm, sh, x)
        torch.cuda.synchronize()
        names = [n for n, _, _ in ops.profile_records()]
        ops.profile_enable(False)
        assert names.count("conv3x3_split_upsp_kernel") == 2 and "conv3x3_split_ups3_kernel" not in names
        ops.UPS_PHASE = False
        with torch.no_grad():
            raw_b, rgb_b = run_network(lm, sh, x)
        torch.cuda.synchronize()
    finally:
        ops.UPS_PHASE = was
    assert (raw_a - raw_b).abs().max().item() <= 2e-5 and (rgb_a - rgb_b).abs().max().item() <= 2e-5
    # against the CPU fp32 path of the same network
    cnet = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, OPT)
    cnet.load_state_dict({k: v.cpu() for k, v in net.state_dict().items()})
    with torch.no_grad():
        out, _ = cnet.eval()(x.cpu())
    from isosurfacesuperresolution_amd.utils import ScreenSpaceShading
    out = torch.cat([out[:, 0:1].clamp(-1, 1), ScreenSpaceShading.normalize(out[:, 1:4], dim=1), out[:, 4:].clamp(0, 1)], dim=1)
    assert (raw_a.cpu() - out).abs().max().item() <= 1e-4
