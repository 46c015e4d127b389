"""The range cliff of the split-operand kernels and the guard that removes it.

An activation is split into (hi, lo') fp16 numbers; |x| >= 65520 overflows that split (csrc/sr_conv_split.hip, header).  The
reference's networks never get there (G-buffer inputs in [-1, 5], small weights), but ``inference.LoadedModel`` takes arbitrary
user checkpoints (SuperresolutionNetwork/inference/loadedmodel.py:16-68), so: (1) the overflow is LOUD (inf / NaN, never a
plausible number), (2) every launch records the largest magnitude it stored, and after the first frame of a model the producers that came close are marked hot and their consumers run on the exact fp32
kernels -- a checkpoint with a badly scaled layer still matches the fp64 CPU network; (3) every LATER frame mirrors the words
into pinned host memory and the next frame reads them at its start: a layer that turns hot in mid-sequence is rerouted one frame
later, without a synchronisation.  (``ops.RANGE_CHECK_EVERY`` is gone.)"""
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


def test_a_layer_that_turns_hot_after_the_first_frame_is_rerouted_one_frame_later():
    """the range guard is continuous.  A network whose activations stay small for frames 0-2 and reach ~4.5e4
    (above the 3e4 threshold, below the fp16 overflow at 65520) from frame 3 on: frame 3 still runs on the split kernels -- and is
    right, that is what the factor-two margin is for -- its guard words travel to pinned memory with the frame, frame 4 STARTS by
    reading them (no synchronisation) and runs every layer of the fused trunk on the exact kernels.  Every frame matches the fp64
    network to 1e-4 of the output scale -- frames 5-6, whose activations (1.35e5) would overflow the split, included; nothing reads the
    device between frames."""
    from isosurfacesuperresolution_amd import models, ops
    from isosurfacesuperresolution_amd.inference import LoadedModel
    opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10)
    torch.manual_seed(15)
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt)
    with torch.no_grad():
        net.blocks[2][0].weight.mul_(300.0)                       # one badly scaled layer inside the fused trunk
        net.postblock[1].weight.mul_(0.02)                        # ... and nothing larger downstream: the trunk holds the maximum
    ref_net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt).double()
    ref_net.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    lm = LoadedModel.from_model(net, "cuda", parameters={"initialImage": "zero"})
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator().manual_seed(16)
    base = torch.rand(1, 12, 40, 64, generator=g)
    base[:, 3] = (base[:, 3] > 0.3).float()

    def low(amp):
        v = base.clone()
        v[:, 4:8] *= amp                                          # normals / depth: the network is (nearly) homogeneous in them
        return v

    def reference(v):
        inp = torch.cat((v[:, 3:4] * 2 - 1, v[:, 4:8], torch.zeros(1, 96, 40, 64)), dim=1)
        with torch.no_grad():
            return ref_net(inp.double())[0]

    def largest_stored():
        st = ops._range_state(dev)
        words = st["buf"][:len(st["slots"])].view(torch.float32)
        trunk, = [i for k, i in st["slots"].items() if isinstance(k, tuple) and k[0] == "trunk"]
        assert float(words.max().item()) <= 1.05 * float(words[trunk].item())      # the trunk's word is the network's maximum
        return float(words[trunk].item())
    # calibrate the amplitudes on the device itself: largest |value| any layer stores at amplitude 1
    with torch.no_grad():
        lm.inference(low(1.0).cuda(), None)
    m1 = largest_stored()
    ops.range_reset()
    small, big = 1.0e3 / m1, 4.5e4 / m1
    assert not ops.any_hot(dev)
    hot_after = []
    for t in range(7):
        # frames 0-2 small; 3-4 hot but representable; 5-6 beyond the fp16 range of the split operands: loud NaN unless rerouted
        v = low(small if t < 3 else big if t < 5 else 3.0 * big)
        with torch.no_grad():
            out = lm.inference(v.cuda(), None)
        torch.cuda.synchronize()                                  # (so that "one frame late" is deterministic in this test)
        hot_after.append(ops.any_hot(dev))
        ref = reference(v)
        scale = ref.abs().max().item()
        assert torch.isfinite(out).all(), t
        assert (out.double().cpu() - ref).abs().max().item() <= 1e-4 * scale, (t, scale)
        if t == 3:
            peak = largest_stored()
            assert 3.2e4 < peak < 6.4e4, peak                     # hot, not yet overflowing: the margin the guard relies on
    # frames 0-3 ran unflagged (3 produced the hot values), frame 4 read frame 3's words at its start
    assert hot_after == [False, False, False, False, True, True, True], hot_after
    st = ops._range_state(dev)
    trunk_keys = [k for k in st["hot"] if isinstance(k, tuple) and k[0] == "trunk"]
    assert trunk_keys and all(id(m.weight) in st["hot"] for b in lm.model.blocks for m in (b[0], b[2]))      # the segment, taken apart conservatively
    ops.range_reset()
