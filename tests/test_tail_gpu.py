"""The fused 1080p tail (csrc/sr_conv_tail.hip): postblock.6 + postblock.8 + frame finish without the 64-channel round trip,
against the three-kernel path it replaces and against an fp64 CPU evaluation of
SuperresolutionNetwork/models/enhancenet.py:119-125,51-90 + mainGUI.py:594-603."""
import argparse

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _setup(h, w, seed, wscale=1.0, xscale=1.0):
    from isosurfacesuperresolution_amd.pipeline import default_shading
    g = torch.Generator().manual_seed(seed)
    f4 = (torch.rand(1, 64, 4 * h, 4 * w, generator=g) * xscale).cuda()          # post-ReLU features: non-negative
    w6 = ((torch.rand(64, 64, 3, 3, generator=g) - 0.5) * 0.08 * wscale).cuda()
    b6 = ((torch.rand(64, generator=g) - 0.5) * 0.1).cuda()
    w8 = ((torch.rand(6, 64, 3, 3, generator=g) - 0.5) * 0.08).cuda()
    b8 = ((torch.rand(6, generator=g) - 0.5) * 0.1).cuda()
    x = torch.rand(1, 101, h, w, generator=g).cuda()
    x[:, 0] = x[:, 0] * 2 - 1
    return f4, w6, b6, w8, b8, x, default_shading("cuda", 30.0)


def _reference64(f4, w6, b6, w8, b8, x):
    """fp64 on the CPU: the two convolutions, the residual reconstruction and the clamp / normalise of the viewer."""
    d = lambda t: t.double().cpu()
    y6 = F.relu(F.conv2d(d(f4), d(w6), d(b6), padding=1))
    out = F.conv2d(y6, d(w8), d(b8), padding=1)
    out[:, :5] += F.interpolate(d(x)[:, :5], size=out.shape[2:], mode='bilinear', align_corners=False)
    n = out[:, 1:4]
    n = n / torch.clamp(n.norm(dim=1, keepdim=True), min=1e-7)
    return torch.cat([out[:, 0:1].clamp(-1, 1), n, out[:, 4:].clamp(0, 1)], dim=1), out


@pytest.mark.parametrize("h,w", [(2, 8), (6, 8), (23, 37), (30, 52), (67, 120)])
def test_tail_matches_the_three_kernel_path_and_fp64(h, w):
    from isosurfacesuperresolution_amd import ops
    f4, w6, b6, w8, b8, x, sh = _setup(h, w, seed=h * 100 + w)
    assert ops.tail_supported(f4, w6, w8)
    with torch.no_grad():
        raw_t, rgb_t = ops.tail_conv_finish(f4, w6, b6, w8, b8, x, sh)
        f6 = ops.conv3x3(f4, w6, b6, act='relu')
        raw_s, rgb_s = ops.final_conv_finish(f6, w8, b8, x, sh)
    torch.cuda.synchronize()
    ref, pre = _reference64(f4, w6, b6, w8, b8, x)
    scale = max(1.0, pre.abs().max().item())
    err_t = (raw_t.double().cpu() - ref).abs().max().item()
    err_s = (raw_s.double().cpu() - ref).abs().max().item()
    # the normalised normal amplifies errors of short vectors: compare where the normal is not degenerate
    assert err_s <= 2e-5 * scale, err_s
    assert err_t <= 2e-5 * scale and err_t <= 4 * err_s + 2e-6, (err_t, err_s)
    assert (raw_t - raw_s).abs().max().item() <= 2e-5 * scale
    assert (rgb_t - rgb_s).abs().max().item() <= 2e-5
    assert torch.isfinite(raw_t).all() and torch.isfinite(rgb_t).all()


@pytest.mark.parametrize("h,w", [(2, 8), (6, 8), (23, 37), (30, 52), (67, 120), (270, 480)])
def test_fused_form_equals_the_two_kernel_form_bit_for_bit(h, w, diag_lib):
    """Form 1 finishes every pixel whose nine partials lie in its own tile inside the convolution kernel and
    assembles the others (the tiles' rims) from per-pixel records; form 0 writes all 54 partial planes and adds
    them in a streaming kernel.  Same partials, same order of additions: equal outputs, at ragged and full sizes.
    (Neither is the default any more: see test_s_form_*.)"""
    import ctypes
    from isosurfacesuperresolution_amd import ops
    lib = ops._sr()
    lib.isrDebugSetTailFused.argtypes = [ctypes.c_int]
    f4, w6, b6, w8, b8, x, sh = _setup(h, w, seed=h * 31 + w)
    out = {}
    try:
        for fused in (0, 1, 0, 1):
            lib.isrDebugSetTailFused(fused)
            with torch.no_grad():
                raw, rgb = ops.tail_conv_finish(f4, w6, b6, w8, b8, x, sh)
            torch.cuda.synchronize()
            if fused in out:
                assert torch.equal(out[fused][0], raw) and torch.equal(out[fused][1], rgb)      # and deterministic
            out[fused] = (raw, rgb)
    finally:
        lib.isrDebugSetTailFused(2)                                   # the default form
    assert torch.equal(out[1][0], out[0][0]), (out[1][0] - out[0][0]).abs().max().item()
    # (the shading arithmetic is inlined into different kernels, where the compiler may contract different multiply-adds)
    assert (out[1][1] - out[0][1]).abs().max().item() <= 1e-6


@pytest.mark.parametrize("h,w", [(4, 8), (11, 20), (30, 52), (135, 240)])
def test_packed_split_producer_and_packed_tail(h, w):
    """postblock.4 can write its output PACKED-SPLIT (the (hi, lo') fp16 units the next layer multiplies) and the tail stages that
    by LDS-DMA: the units are the split of exactly the fp32 values the ordinary launch stores, and the tail's output is bit for
    bit the one it computes from the fp32 tensor."""
    from isosurfacesuperresolution_amd import ops
    g = torch.Generator().manual_seed(h * 13 + w)
    f2 = torch.rand(1, 64, 2 * h, 2 * w, generator=g).cuda()
    w4 = ((torch.rand(64, 64, 3, 3, generator=g) - 0.5) * 0.08).cuda()
    b4 = ((torch.rand(64, generator=g) - 0.5) * 0.1).cuda()
    _, w6, b6, w8, b8, x, sh = _setup(h, w, seed=3)
    assert ops.packed_supported(f2, w4, True)
    with torch.no_grad():
        f4 = ops.conv3x3_split(f2, w4, b4, act='relu', upsample2x=True)
        f4p = ops.conv3x3_split_packed(f2, w4, b4, act='relu', upsample2x=True)
        back = f4p.to_float()
        # hi + lo' 2^-11 carries 22 significand bits of the fp32 value (and is exactly it for most values)
        assert (back - f4).abs().max().item() <= 2.0 ** -21 * max(1.0, f4.abs().max().item())
        hi = f4p.data.view(torch.float16).view(2, 8, f4p.plane, 8)[0, :, :16 * h * w].float().permute(0, 2, 1).reshape(1, 64, 4 * h, 4 * w)
        assert torch.equal(hi, f4.half().float())                                      # hi = RN16(value), bit for bit
        raw_a, rgb_a = ops.tail_conv_finish(f4, w6, b6, w8, b8, x, sh)
        raw_b, rgb_b = ops.tail_conv_finish(f4p, w6, b6, w8, b8, x, sh)
        raw_c, _ = ops.tail_conv_finish(f4p, w6, b6, w8, b8, x, sh)
    torch.cuda.synchronize()
    assert torch.equal(raw_a, raw_b) and torch.equal(rgb_a, rgb_b) and torch.equal(raw_b, raw_c)
    # plain (non-upsampling) producer as well
    with torch.no_grad():
        f5 = ops.conv3x3_split(f4, w6, b6, act='relu')
        f5p = ops.conv3x3_split_packed(f4, w6, b6, act='relu')
    assert (f5p.to_float() - f5).abs().max().item() <= 2.0 ** -21 * max(1.0, f5.abs().max().item())


def test_tail_with_padded_planes_and_without_shading_or_bias():
    from isosurfacesuperresolution_amd import ops
    h, w = 12, 20
    f4, w6, b6, w8, b8, x, sh = _setup(h, w, seed=5)
    padded = torch.zeros(64 * (16 * h * w + 4 * w), device="cuda").as_strided((1, 64, 4 * h, 4 * w), (64 * (16 * h * w + 4 * w), 16 * h * w + 4 * w, 4 * w, 1))
    padded.copy_(f4)
    with torch.no_grad():
        raw_a, rgb_a = ops.tail_conv_finish(f4, w6, b6, w8, b8, x, sh)
        raw_b, rgb_b = ops.tail_conv_finish(padded, w6, b6, w8, b8, x, sh)
        raw_c, rgb_c = ops.tail_conv_finish(f4, w6, None, w8, None, x, None)
    assert torch.equal(raw_a, raw_b) and torch.equal(rgb_a, rgb_b)
    assert rgb_c is None
    ref, _ = _reference64(f4, w6, torch.zeros_like(b6), w8, torch.zeros_like(b8), x)
    assert (raw_c.double().cpu() - ref).abs().max().item() <= 2e-5


def test_tail_is_deterministic_and_independent_of_the_tile_origin():
    """Every pixel's nine partials are added in a fixed order and each partial is a fixed-order MFMA chain over the
    pixel's own 3x3 neighbourhood: the same pixels computed as part of a larger image (other tile boundaries, other
    workgroups) come out bit for bit -- what the strip super-resolution of parallel_sr.py relies on."""
    from isosurfacesuperresolution_amd import ops
    h, w = 24, 32
    f4, w6, b6, w8, b8, x, sh = _setup(h, w, seed=11)
    with torch.no_grad():
        raw_a, _ = ops.tail_conv_finish(f4, w6, b6, w8, b8, x, None)
        raw_a2, _ = ops.tail_conv_finish(f4, w6, b6, w8, b8, x, None)
        # a crop whose origin is not a multiple of the 8 x 32 tile: rows 12.., columns 20.. (high-res), interior compared
        f4c = f4[:, :, 12:, 20:].contiguous()
        xc = x[:, :, 3:, 5:].contiguous()
        raw_c, _ = ops.tail_conv_finish(f4c, w6, b6, w8, b8, xc, None)
    assert torch.equal(raw_a, raw_a2)
    # the residual reconstruction resamples the low-res input (different at the crop's border) and the convolutions pad with
    # zeros at the crop's border: compare channel 5 (no reconstruction) two pixels inside
    assert torch.equal(raw_a[:, 5, 12 + 2:-2, 20 + 2:-2], raw_c[:, 5, 2:-2, 2:-2])


def test_pipeline_uses_the_fused_tail_and_matches_the_unfused_frame():
    from isosurfacesuperresolution_amd import models, ops, volumes as V
    from isosurfacesuperresolution_amd.inference import DirectRenderer, LoadedModel
    from isosurfacesuperresolution_amd.pipeline import SuperResolutionPipeline, default_shading
    opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10)
    torch.manual_seed(3)
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt)
    lm = LoadedModel.from_model(net, "cuda", parameters={"initialImage": "zero"})
    renderer = DirectRenderer()
    renderer.load_dense(V.ejecta(64))
    origins = [V.orbit_camera(k) for k in range(3)]
    frames = {}
    for fused in (True, False, "unpacked"):
        ops.TAIL_FUSION = bool(fused)
        ops.TAIL_PACKED = fused is True
        try:
            pipe = SuperResolutionPipeline(renderer, lm, default_shading("cuda", 30.0), (96, 56))
            pipe.set_static(fov=30.0, isovalue=0.34)
            pipe.frame(origins[0])
            pipe.reset()
            ops.profile_enable(True)
            out = [tuple(t.clone() for t in pipe.frame(o)) for o in origins]
            torch.cuda.synchronize()
            names = {n for n, _, _ in ops.profile_records()}
            ops.profile_enable(False)
        finally:
            ops.TAIL_FUSION = True
            ops.TAIL_PACKED = True
        assert ("conv3x3_split_tail_kernel" in names) == bool(fused)
        assert ("conv3x3_small_cout_kernel" in names) == (not fused)
        frames[fused] = out
    for (rgb_a, raw_a), (rgb_b, raw_b) in zip(frames[True], frames["unpacked"]):
        assert torch.equal(raw_a, raw_b) and torch.equal(rgb_a, rgb_b)               # packed-split hand-over: nothing changes
    for (rgb_a, raw_a), (rgb_b, raw_b) in zip(frames[True], frames[False]):
        assert (raw_a - raw_b).abs().max().item() <= 1e-4 and (rgb_a - rgb_b).abs().max().item() <= 1e-4
    assert (frames[True][0][1] - frames[False][0][1]).abs().max().item() <= 5e-5      # first frame: no recurrence yet (normalised normals amplify)


@pytest.mark.parametrize("h,w", [(2, 8), (6, 8), (23, 37), (30, 52), (67, 120), (270, 480)])
def test_s_form_matches_the_54_plane_form_and_fp64(h, w, diag_lib):
    """The default form (2): every wave adds the three horizontal taps of each (dy, c) for its row of 32 pixels and stores 18 S planes
    (+ small records for the rows' end pixels) instead of 54 partial planes.  The additions associate differently from the 54-plane
    form's tap-by-tap sum -- ((z0 + z1) + z2 per row, then the three rows) -- so the two agree to rounding, not bit for bit; both
    are measured against fp64.  Ragged sizes (W % 32 != 0: 32, 148, 208; H % 8 != 0) included."""
    import ctypes
    from isosurfacesuperresolution_amd import ops
    lib = ops._sr()
    lib.isrDebugSetTailFused.argtypes = [ctypes.c_int]
    f4, w6, b6, w8, b8, x, sh = _setup(h, w, seed=h * 17 + w)
    out = {}
    try:
        for form in (2, 0, 2):
            lib.isrDebugSetTailFused(form)
            with torch.no_grad():
                raw, rgb = ops.tail_conv_finish(f4, w6, b6, w8, b8, x, sh)
            torch.cuda.synchronize()
            if form in out:
                assert torch.equal(out[form][0], raw) and torch.equal(out[form][1], rgb)      # deterministic
            out[form] = (raw, rgb)
    finally:
        lib.isrDebugSetTailFused(2)
    ref, pre = _reference64(f4, w6, b6, w8, b8, x)
    scale = max(1.0, pre.abs().max().item())
    err2 = (out[2][0].double().cpu() - ref).abs().max().item()
    err0 = (out[0][0].double().cpu() - ref).abs().max().item()
    assert err2 <= 2e-5 * scale and err2 <= 2 * err0 + 2e-6, (err2, err0)
    assert (out[2][0] - out[0][0]).abs().max().item() <= 1e-5 * scale
    assert (out[2][1] - out[0][1]).abs().max().item() <= 1e-5
    assert torch.isfinite(out[2][0]).all() and torch.isfinite(out[2][1]).all()


@pytest.mark.parametrize("h,w", [(2, 8), (6, 8), (23, 37), (30, 52), (67, 120), (135, 240), (270, 480)])
def test_v_form_equals_the_s_form_bit_for_bit(h, w, diag_lib):
    """Form 4: the vertical sums of a tile's inner rows are made in the convolution kernel -- 6 planes per
    row + the rows 0 / 7 exchanges instead of 18 S planes -- per pixel the same additions in the same order as the S form
    (((bias + S0) + S1) + S2): EQUAL bit for bit, at ragged sizes (H % 8 != 0, W % 32 != 0) and with the packed-split input the frame uses."""
    import ctypes
    from isosurfacesuperresolution_amd import ops
    lib = diag_lib
    lib.isrDebugSetTailFused.argtypes = [ctypes.c_int]
    f4, w6, b6, w8, b8, x, sh = _setup(h, w, seed=h * 13 + w)
    out = {}
    try:
        for form in (2, 4, 4):
            lib.isrDebugSetTailFused(form)
            with torch.no_grad():
                raw, rgb = ops.tail_conv_finish(f4, w6, b6, w8, b8, x, sh)
                packed = None
                if h >= 4 and ops.packed_supported(f4, w6, False):
                    ps = ops.pack_split(f4)
                    packed = ops.tail_conv_finish(ps, w6, b6, w8, b8, x, sh)
            torch.cuda.synchronize()
            if form in out:
                assert torch.equal(out[form][0], raw) and torch.equal(out[form][1], rgb)      # deterministic
            out[form] = (raw, rgb, packed)
    finally:
        lib.isrDebugSetTailFused(2)
    assert torch.equal(out[4][0], out[2][0]), (out[4][0] - out[2][0]).abs().max().item()
    assert (out[4][1] - out[2][1]).abs().max().item() <= 1e-6             # (shading inlined into two kernels: contraction may differ)
    if out[2][2] is not None:
        assert torch.equal(out[4][2][0], out[2][2][0])


def test_s_form_does_not_depend_on_where_tile_borders_fall():
    """The same pixels as part of images cropped at every horizontal offset 0 .. 33 (every position of a pixel relative to the 32-pixel
    tiles: interior, first, last, next to an end) and two vertical offsets: bit for bit equal two pixels inside the crops."""
    from isosurfacesuperresolution_amd import ops
    h, w = 24, 40
    f4, w6, b6, w8, b8, x, sh = _setup(h, w, seed=21)
    with torch.no_grad():
        full, _ = ops.tail_conv_finish(f4, w6, b6, w8, b8, x, None)
        for oy in (0, 4):
            for ox in range(0, 36, 4):
                f4c = f4[:, :, oy:, ox:].contiguous()
                xc = x[:, :, oy // 4:, ox // 4:].contiguous()
                part, _ = ops.tail_conv_finish(f4c, w6, b6, w8, b8, xc, None)
                assert torch.equal(full[:, 5, oy + 2:-2, ox + 2:-2], part[:, 5, 2:-2, 2:-2]), (oy, ox)
