"""The range cliff of the split-operand kernels and the guard that removes it.

An activation is split into (hi, lo') fp16 numbers; |x| >= 65520 overflows that split (csrc/sr_conv_split.hip, header).  The
reference's networks never get there (G-buffer inputs in [-1, 5], small weights), but ``inference.LoadedModel`` takes arbitrary
user checkpoints (SuperresolutionNetwork/inference/loadedmodel.py:16-68), so: (1) the overflow is LOUD (inf / NaN, never a
plausible number), (2) every launch records the largest magnitude it stored, and after the first frame of a model (then every
``ops.RANGE_CHECK_EVERY`` frames) the producers that came close are marked hot and their consumers run on the exact fp32
kernels -- a checkpoint with a badly scaled layer still matches the fp64 CPU network."""
import argparse

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mag", [6.0e4, 7.0e4, 1.0e6])
def test_split_overflow_is_loud_and_the_exact_kernel_is_not(mag):
    from isosurfacesuperresolution_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.rand(1, 64, 24, 40, generator=g) - 0.5
    x[0, 5, 10, 17] = mag
    w = (torch.rand(64, 64, 3, 3, generator=g) - 0.5) * 0.05
    ref = F.conv2d(x.double(), w.double(), padding=1)
    with torch.no_grad():
        y_split = ops.conv3x3_split(x.cuda(), w.cuda())
        old = ops.SPLIT_F16
        ops.SPLIT_F16 = False
        try:
            y_exact = ops.conv3x3(x.cuda(), w.cuda())
        finally:
            ops.SPLIT_F16 = old
    assert torch.isfinite(y_exact).all()
    assert (y_exact.double().cpu() - ref).abs().max().item() <= 1e-6 * mag
    if mag < 65520:
        assert torch.isfinite(y_split).all() and (y_split.double().cpu() - ref).abs().max().item() <= 1e-6 * mag
    else:
        # the 3x3 footprint of the overflowing value is non-finite in every output channel it feeds -- loud, not a plausible number
        window = y_split[0, :, 9:12, 16:19]
        assert not torch.isfinite(window).any()
        away = y_split.clone()
        away[0, :, 9:12, 16:19] = 0
        assert torch.isfinite(away).all()


def test_range_guard_routes_the_consumer_of_a_hot_tensor_to_the_exact_kernel():
    from isosurfacesuperresolution_amd import ops
    ops.range_reset()
    g = torch.Generator().manual_seed(4)
    x = (torch.rand(1, 64, 32, 64, generator=g)).cuda()
    w1 = ((torch.rand(64, 64, 3, 3, generator=g) - 0.5) * 0.05).cuda() * 1.0e5       # the badly scaled layer
    w2 = ((torch.rand(64, 64, 3, 3, generator=g) - 0.5) * 0.05).cuda()
    ref = F.conv2d(F.relu(F.conv2d(x.double().cpu(), w1.double().cpu(), padding=1)), w2.double().cpu(), padding=1)
    with torch.no_grad():
        t = ops.conv3x3(x, w1, act='relu')
        assert t.abs().max().item() > 65520 and torch.isfinite(t).all()        # the PRODUCER is fine: fp32 output
        y1 = ops.conv3x3(t, w2)                                                   # first time: split consumer, overflow, loud
        assert not torch.isfinite(y1).all()
        assert not ops.any_hot(x.device)
        new = ops.refresh_range_flags(x.device)
        assert id(w1) in new and ops.any_hot(x.device)
        t = ops.conv3x3(x, w1, act='relu')
        y2 = ops.conv3x3(t, w2)                                                   # now routed to the exact kernel
        assert y2._isr_range_key == ops.HOT
    assert torch.isfinite(y2).all()
    assert (y2.double().cpu() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    ops.range_reset()
    assert not ops.any_hot(x.device)


def test_checkpoint_with_a_badly_scaled_layer_still_matches_the_fp64_network(tmp_path):
    """A model whose block-3 convolution has its weights scaled by 1e5 (activations of 1e5..1e7 from there on): the first frame
    trips the guard, is recomputed with the exact routing and matches the fp64 CPU network to 1e-4 relative; later frames keep
    the routing without another host read.  Through the frame pipeline (fused kernels step aside) and through LoadedModel."""
    from isosurfacesuperresolution_amd import models, ops, volumes as V
    from isosurfacesuperresolution_amd.inference import DirectRenderer, LoadedModel
    from isosurfacesuperresolution_amd.pipeline import SuperResolutionPipeline, default_shading
    opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10)
    torch.manual_seed(5)
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt)
    with torch.no_grad():
        net.blocks[3][0].weight.mul_(1.0e5)
    ref_net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt).double()
    ref_net.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    lm = LoadedModel.from_model(net, "cuda", parameters={"initialImage": "zero"})
    low = torch.rand(1, 12, 24, 40, generator=torch.Generator().manual_seed(6))
    low[:, 3] = (low[:, 3] > 0.3).float()
    with torch.no_grad():
        out = lm.inference(low.cuda(), None)
        inp = torch.cat((low[:, 3:4] * 2 - 1, low[:, 4:8], torch.zeros(1, 96, 24, 40)), dim=1)
        ref, _ = ref_net(inp.double())
    assert ops.any_hot(torch.device("cuda", torch.cuda.current_device())) or ops.any_hot(out.device)
    scale = ref.abs().max().item()
    assert scale > 1e4 and torch.isfinite(out).all()
    assert (out.double().cpu() - ref).abs().max().item() <= 1e-4 * scale
    with torch.no_grad():
        out2 = lm.inference(low.cuda(), None)                  # no further host read, routing kept
    assert torch.equal(out, out2)
    # the frame pipeline: fused tail / packed / block paths step aside while a layer is hot
    renderer = DirectRenderer()
    renderer.load_dense(V.ejecta(64))
    lm2 = LoadedModel.from_model(net, "cuda", parameters={"initialImage": "zero"})      # resets the guard
    assert not ops.any_hot(out.device)
    pipe = SuperResolutionPipeline(renderer, lm2, default_shading("cuda", 30.0), (96, 56))
    pipe.set_static(fov=30.0, isovalue=0.34)
    rgb, raw = pipe.frame(V.orbit_camera(0))
    assert ops.any_hot(raw.device) and torch.isfinite(raw).all() and torch.isfinite(rgb).all()
    rgb2, raw2 = pipe.frame(V.orbit_camera(1))
    assert torch.isfinite(raw2).all()
    ops.range_reset()
