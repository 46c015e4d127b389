"""GPU parity of the fused HIP conv3x3 (libisr_sr.so, through the C-ABI) against a plain PyTorch
fp32 CPU reference of the same op.  Tolerance: 1e-4 (BASELINE.json north_star), on O(1) data."""
import argparse
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "sr_reference.npz"))


def _ref(x, w, b, act, slope, res, ups):
    x = x.double(); w = w.double()
    if ups:
        x = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)
    y = F.conv2d(x, w, b.double() if b is not None else None, padding=1)
    if act == 'relu':
        y = F.relu(y)
    elif act == 'leaky':
        y = F.leaky_relu(y, slope)
    if res is not None:
        y = y + res.double()
    return y


CASES = [
    # N, Cin, Cout, h, w, act, bias, residual, upsample
    (1, 64, 64, 16, 32, 'relu', True, False, False),
    (1, 64, 64, 37, 45, 'none', True, True, False),       # ragged tile edges + fused skip
    (2, 101, 64, 20, 33, 'relu', True, False, False),     # pre-block: Cin padded 101 -> 112
    (1, 64, 6, 24, 40, 'none', True, False, False),       # final layer: Cout padded 6 -> 32
    (1, 64, 64, 18, 22, 'relu', True, False, True),       # fused bilinear x2 -> 36x44
    (3, 5, 7, 9, 11, 'leaky', False, False, False),
    (1, 6, 64, 8, 8, 'none', False, False, False),        # shape of the data gradient of the final layer
    (1, 64, 64, 1, 1, 'relu', True, False, False),
    (1, 64, 64, 24, 36, 'none', True, True, False),       # fused skip on the dwordx4 (W % 4 == 0) epilogue
    (2, 64, 6, 20, 64, 'relu', True, True, False),
    (1, 64, 64, 8, 8, 'relu', True, False, True),         # tiny upsampled tile
    (1, 101, 8, 33, 70, 'leaky', True, True, False),      # small-Cout kernel: ragged tiles, Cin % 4 != 0, all 8 channels
    (2, 3, 1, 17, 130, 'none', False, False, False),      # small-Cout kernel: one channel, one partial chunk
    (1, 64, 64, 34, 38, 'relu', True, True, True),        # upsampled, ragged, fused skip
]


@pytest.fixture(params=["split", "exact"])
def conv_mode(request):
    """Inference convolutions run on the split-operand kernel (ops.SPLIT_F16, the default) or on the exact k-ordered
    fmaf-chain kernels; both must meet the same bar on every case."""
    from isosurfacesuperresolution_amd import ops
    old = ops.SPLIT_F16
    ops.SPLIT_F16 = request.param == "split"
    yield request.param
    ops.SPLIT_F16 = old


@pytest.fixture
def exact_mode():
    from isosurfacesuperresolution_amd import ops
    old = ops.SPLIT_F16
    ops.SPLIT_F16 = False
    yield
    ops.SPLIT_F16 = old


@pytest.mark.parametrize("case", CASES)
def test_conv3x3_forward(case, conv_mode):
    _forward_case(case)


def _forward_case(case):
    from isosurfacesuperresolution_amd import ops
    N, Cin, Cout, h, w, act, has_b, has_r, ups = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.rand(N, Cin, h, w, generator=g) * 2 - 1
    wt = (torch.rand(Cout, Cin, 3, 3, generator=g) * 2 - 1) / (3.0 * Cin ** 0.5)
    b = torch.rand(Cout, generator=g) - 0.5 if has_b else None
    H, W = (2 * h, 2 * w) if ups else (h, w)
    res = torch.rand(N, Cout, H, W, generator=g) if has_r else None
    ref = _ref(x, wt, b, act, 0.1, res, ups)
    with torch.no_grad():
        y = ops.conv3x3(x.cuda(), wt.cuda(), b.cuda() if has_b else None, act=act, slope=0.1,
                        residual=res.cuda() if has_r else None, upsample2x=ups)
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    err = (y.cpu().double() - ref).abs().max().item()
    assert err <= 1e-4, err


@pytest.mark.parametrize("tile", [1, 2, 3])
@pytest.mark.parametrize("case", [c for c in CASES if c[2] > 8 or c[8]])
def test_conv3x3_forward_both_tilings(case, tile, exact_mode, diag_lib):
    """The forward kernel has a 16x32 and a 4x32 tiling and a one-row-per-workgroup form with K split over the waves
    (picked by problem size; 3 = the latter, which has no upsampling variant and falls back to the 4x32 tiling
    there); all must give the reference result on every case, whatever the heuristic would choose."""
    from isosurfacesuperresolution_amd import ops
    lib = ops._sr()
    lib.isrDebugSetForwardTile(tile)
    try:
        _forward_case(case)
    finally:
        lib.isrDebugSetForwardTile(0)


def test_conv3x3_padded_channel_planes(conv_mode):
    """Tensors with padded channel planes (ops.empty_planes) go through the strided C entry points unchanged."""
    from isosurfacesuperresolution_amd import ops
    g = torch.Generator().manual_seed(11)
    x = torch.rand(2, 64, 20, 36, generator=g) * 2 - 1
    w1 = (torch.rand(64, 64, 3, 3, generator=g) * 2 - 1) / 24.0
    w2 = (torch.rand(6, 64, 3, 3, generator=g) * 2 - 1) / 24.0
    b1, b2 = torch.rand(64, generator=g), torch.rand(6, generator=g)
    res = torch.rand(2, 64, 40, 72, generator=g)
    ref1 = _ref(x, w1, b1, 'relu', 0.1, res, True)
    ref2 = _ref(ref1.float(), w2, b2, 'none', 0.1, None, False)
    old = ops.plane_pad
    ops.plane_pad = lambda h, w: 52        # force padding at every size (a multiple of 4 floats)
    try:
        with torch.no_grad():
            xp = ops.empty_planes(2, 64, 20, 36, 'cuda'); xp.copy_(x)
            rp = ops.empty_planes(2, 64, 40, 72, 'cuda'); rp.copy_(res)
            y1 = ops.conv3x3(xp, w1.cuda(), b1.cuda(), act='relu', residual=rp, upsample2x=True)
            assert y1.stride(1) == 40 * 72 + 52 and not y1.is_contiguous()
            y2 = ops.conv3x3(y1, w2.cuda(), b2.cuda())
    finally:
        ops.plane_pad = old
    assert (y1.cpu().double() - ref1).abs().max().item() <= 1e-4
    assert (y2.cpu().double() - ref2).abs().max().item() <= 1e-4


@pytest.mark.parametrize("form", ["tile", "stream", "wide", "rows2"])
@pytest.mark.parametrize("slots", [8, 24])
def test_split_kernel_forms_agree_when_workgroups_walk_many_tiles(form, slots, diag_lib):
    """The plain split-operand layer has four kernel forms (one workgroup per 8x32 tile; persistent streaming; wide
    512-thread; 2-row tiles for small images).  The persistent ones walk a list of tiles per workgroup -- with the grid
    capped to a few workgroups even a small image exercises the tile-to-tile hand-over (next tile's operands in flight
    under the last k-step, epilogue scratch vs parked data); all of them see ragged last tile rows and an odd number of
    k-steps (Cin = 101: four 64-channel-chunk / k-step-pair boundaries), a 32-channel output and a fused skip."""
    import ctypes
    from isosurfacesuperresolution_amd import ops
    lib = ops._sr()
    for fn in (lib.isrDebugSetSplitAlgo, lib.isrDebugSetSplitSlots, lib.isrDebugSetSplitSmall):
        fn.argtypes = [ctypes.c_int]
    g = torch.Generator().manual_seed(31)
    old = ops.SPLIT_F16
    ops.SPLIT_F16 = True
    lib.isrDebugSetSplitAlgo({"tile": 0, "stream": 1, "wide": 2, "rows2": 1}[form])
    lib.isrDebugSetSplitSmall(1 if form == "rows2" else 0)
    lib.isrDebugSetSplitSlots(slots)
    try:
        for N, Cin, Cout, h, w, act, has_r in ((1, 64, 64, 70, 96, 'relu', False), (2, 101, 64, 45, 64, 'none', True), (1, 64, 32, 31, 128, 'relu', True),
                                               (16, 64, 64, 32, 32, 'relu', True)):
            x = torch.rand(N, Cin, h, w, generator=g) * 2 - 1
            wt = (torch.rand(Cout, Cin, 3, 3, generator=g) * 2 - 1) / (3.0 * Cin ** 0.5)
            b = torch.rand(Cout, generator=g) - 0.5
            res = torch.rand(N, Cout, h, w, generator=g) if has_r else None
            ref = _ref(x, wt, b, act, 0.1, res, False)
            with torch.no_grad():
                y = ops.conv3x3(x.cuda(), wt.cuda(), b.cuda(), act=act, residual=res.cuda() if has_r else None)
            err = (y.cpu().double() - ref).abs().max().item()
            assert err <= 1e-4, (form, slots, (N, Cin, Cout, h, w), err)
    finally:
        lib.isrDebugSetSplitAlgo(1)
        lib.isrDebugSetSplitSlots(0)
        lib.isrDebugSetSplitSmall(1)
        ops.SPLIT_F16 = old


def test_split_operand_accuracy_matches_the_exact_kernel():
    """The split-operand kernel (three fp16 MFMAs per product, fp32 accumulation) against an fp64 convolution, next to
    the exact fp32 kernel on the same data: 64 -> 64 channels (K = 576), O(1) activations, weights of the network's
    scale, plus the corners of the fp16 range -- tiny activations (x_lo subnormal), large ones (|x| up to 2000),
    tiny and large weights (the per-layer power-of-two scale)."""
    from isosurfacesuperresolution_amd import ops
    g = torch.Generator().manual_seed(21)
    old = ops.SPLIT_F16
    try:
        for xs, ws in ((1.0, 0.06), (1e-3, 0.06), (2000.0, 0.06), (1.0, 1e-5), (1.0, 300.0), (30.0, 0.5)):
            x = (torch.rand(1, 64, 40, 64, generator=g) * 2 - 1) * xs
            x[:, ::3] = torch.relu(x[:, ::3])                               # exact zeros as after a ReLU
            w = (torch.rand(64, 64, 3, 3, generator=g) * 2 - 1) * ws
            b = (torch.rand(64, generator=g) - 0.5) * xs * ws
            ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
            scale = ref.abs().max().item()
            err = {}
            for mode in (True, False):
                ops.SPLIT_F16 = mode
                with torch.no_grad():
                    y = ops.conv3x3(x.cuda(), w.cuda(), b.cuda())
                err[mode] = (y.cpu().double() - ref).abs().max().item() / scale
            # both are at the level of fp32 rounding of a 576-term sum; the split kernel must not be worse than 2x the exact one
            assert err[True] <= 2.0 * err[False] + 2e-7, (xs, ws, err)
            assert err[True] <= 2e-6, (xs, ws, err)
    finally:
        ops.SPLIT_F16 = old


def test_split_operand_network_matches_the_exact_kernels():
    """The whole EnhanceNet forward on the split-operand kernels and on the exact fp32 kernels against the same network
    in fp64 on the CPU: both sit at the same distance from the fp64 answer (the split path at most 2x the exact one),
    far inside the 1e-4 bar."""
    from isosurfacesuperresolution_amd import models, ops
    opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10)
    torch.manual_seed(0)
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt).eval()
    x = torch.rand(1, 101, 24, 32, generator=torch.Generator().manual_seed(2))
    net64 = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt).eval().double()
    net64.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    with torch.no_grad():
        ref = net64(x.double())[0]
    net = net.cuda()
    old = ops.SPLIT_F16
    out = {}
    try:
        for mode in (True, False):
            ops.SPLIT_F16 = mode
            with torch.no_grad():
                out[mode] = net(x.cuda())[0].cpu().double()
    finally:
        ops.SPLIT_F16 = old
    assert not torch.equal(out[True], out[False])                           # two different kernels did run
    e_split, e_exact = (out[True] - ref).abs().max().item(), (out[False] - ref).abs().max().item()
    assert e_exact <= 1e-4 and e_split <= 1e-4, (e_split, e_exact)
    assert e_split <= 2.0 * e_exact + 1e-6, (e_split, e_exact)


def test_conv3x3_is_exact_fma_chain_on_integers(conv_mode):
    """fp32 MFMA is an exact fmaf chain: small-integer data must come out exactly (and so it must on the split-operand
    kernel: small integers are their own fp16 'hi' part, 'lo' is zero, products and fp32 sums are exact)."""
    from isosurfacesuperresolution_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randint(-3, 4, (1, 64, 19, 35), generator=g).float()
    w = torch.randint(-2, 3, (64, 64, 3, 3), generator=g).float()
    ref = F.conv2d(x.double(), w.double(), padding=1)
    with torch.no_grad():
        y = ops.conv3x3(x.cuda(), w.cuda())
    assert torch.equal(y.cpu().double(), ref)


@pytest.mark.parametrize("shape", [(2, 64, 64, 32, 32, 'relu'), (1, 101, 64, 17, 23, 'relu'),
                                   (2, 64, 6, 40, 24, 'none'), (1, 64, 64, 64, 96, 'none')])
def test_conv3x3_backward(shape):
    from isosurfacesuperresolution_amd import ops
    N, Cin, Cout, h, w, act = shape
    g = torch.Generator().manual_seed(11)
    x = (torch.rand(N, Cin, h, w, generator=g) * 2 - 1)
    wt = ((torch.rand(Cout, Cin, 3, 3, generator=g) * 2 - 1) / (3.0 * Cin ** 0.5))
    b = torch.rand(Cout, generator=g) - 0.5
    gy = torch.rand(N, Cout, h, w, generator=g) * 2 - 1
    xr, wr, br = (t.double().requires_grad_() for t in (x, wt, b))
    yr = F.conv2d(xr, wr, br, padding=1)
    if act == 'relu':
        yr = F.relu(yr)
    yr.backward(gy.double())
    xg, wg, bg = (t.cuda().requires_grad_() for t in (x, wt, b))
    y = ops.conv3x3(xg, wg, bg, act=act)
    y.backward(gy.cuda())
    torch.cuda.synchronize()
    assert (y.detach().cpu().double() - yr.detach()).abs().max() <= 1e-4
    assert (xg.grad.cpu().double() - xr.grad).abs().max() <= 1e-4
    scale = max(1.0, wr.grad.abs().max().item())
    assert (wg.grad.cpu().double() - wr.grad).abs().max() / scale <= 1e-4
    assert (bg.grad.cpu().double() - br.grad).abs().max() / max(1.0, br.grad.abs().max().item()) <= 1e-4


def test_enhancenet_gpu_matches_reference_fixture():
    from isosurfacesuperresolution_amd import models
    opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10)
    torch.manual_seed(0)
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt).eval().cuda()
    torch.manual_seed(1)
    x = torch.rand(1, 101, 8, 8)
    with torch.no_grad():
        y, raw = net(x.cuda())
    assert np.abs(y.cpu().numpy() - G["net_y"]).max() <= 1e-4
    assert np.abs(raw.cpu().numpy() - G["net_raw"]).max() <= 1e-4


def test_enhancenet_gpu_train_step_matches_cpu():
    """forward + backward through the HIP kernels vs. the same network in fp64 on CPU PyTorch.
    Gradients pass through 24 ReLU layers, so they are compared in relative L2 norm (the fp32 CPU
    path of PyTorch itself sits at the same distance from the fp64 answer)."""
    from isosurfacesuperresolution_amd import models
    opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10)
    torch.manual_seed(3)
    cpu = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt)
    gpu = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt)
    gpu.load_state_dict(cpu.state_dict())
    gpu = gpu.cuda()
    cpu = cpu.double()
    x = torch.rand(2, 101, 12, 12)
    tgt = torch.rand(2, 6, 48, 48)
    lc = F.mse_loss(cpu(x.double())[0], tgt.double())
    lc.backward()
    lg = F.mse_loss(gpu(x.cuda())[0], tgt.cuda())
    lg.backward()
    assert abs(lc.item() - lg.item()) <= 1e-4 * max(1.0, abs(lc.item()))
    # A ReLU whose pre-activation is within fp32 rounding of zero may switch on one path and not
    # on the other; one such flip moves the gradients upstream of it by O(1e-3) in relative L2
    # (observed: exactly one, at blocks.1.0, for this seed).  So: most tensors must agree to fp32
    # rounding, all of them to 1e-2.
    errs = []
    for (n, pc), (_, pg) in zip(cpu.named_parameters(), gpu.named_parameters()):
        errs.append((pc.grad - pg.grad.cpu().double()).norm().item() / pc.grad.norm().item())
    assert max(errs) <= 1e-2, max(errs)
    assert sum(e <= 1e-5 for e in errs) >= 0.8 * len(errs), sorted(errs)[-12:]


def test_fused_frame_kernels_match_module_path():
    """isrAssembleInput / isrFinishFrame (one launch each) vs. the module-level PyTorch path
    (LoadedModel.inference + clamp/normalise + ScreenSpaceShading), over a short temporal sequence."""
    from isosurfacesuperresolution_amd import models, ops
    from isosurfacesuperresolution_amd.inference import LoadedModel, fill_flow
    from isosurfacesuperresolution_amd.pipeline import default_shading
    from isosurfacesuperresolution_amd.utils import ScreenSpaceShading
    opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10)
    torch.manual_seed(7)
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt).cuda().eval()
    for mode in ("zero", "input", "unshaded"):
        lm = LoadedModel.from_model(net, "cuda", parameters={"initialImage": mode})
        sh = default_shading("cuda", 30.0)
        h, w = 23, 37
        prev_a = prev_b = None
        for step in range(3):
            g = torch.rand(h, w, 12, device="cuda")
            g[..., 3] = (g[..., 3] > 0.4).float()
            g[..., 8:10] = (g[..., 8:10] - 0.5) * 0.05
            low = g.permute(2, 0, 1).unsqueeze(0)
            with torch.no_grad():
                raw_a = lm.inference(low, prev_a)
                raw_a = torch.cat([raw_a[:, 0:1].clamp(-1, 1), ScreenSpaceShading.normalize(raw_a[:, 1:4], dim=1),
                                   raw_a[:, 4:].clamp(0, 1)], dim=1)
                rgb_a = sh(raw_a)
                flow = ops.fill_flow_gbuffer(g) if prev_b is not None else None
                if flow is not None:
                    assert (flow - fill_flow(low[:, 8:10], low[:, 3:4] != 0)).abs().max().item() <= 1e-5
                x = ops.assemble_input(g, flow, prev_b, mode, False)
                feat = net.forward_features(x)
                raw_b, rgb_b = ops.finish_frame(feat, x, sh)
            assert (raw_a - raw_b).abs().max().item() <= 1e-4, (mode, step)
            assert (rgb_a - rgb_b).abs().max().item() <= 1e-4, (mode, step)
            # ... and the last layer fused with the finishing (isrConvSmallFinishFrame) gives the same frame (the shading
            # arithmetic is inlined into another kernel, where the compiler may contract different multiply-adds)
            with torch.no_grad():
                last = net.postblock[8]
                raw_c, rgb_c = ops.final_conv_finish(net.forward_features(x, last_layer=False), last.weight, last.bias, x, sh)
            assert torch.equal(raw_c, raw_b) and (rgb_c - rgb_b).abs().max().item() <= 1e-6, (mode, step)
            prev_a, prev_b = raw_a, raw_b


@pytest.mark.parametrize("h,w", [(270, 480), (135, 240), (23, 37)])
def test_assembled_input_is_bit_identical_to_the_module_path(h, w):
    """The temporal input path -- hole-filled flow -> x4 resize -> pixel grid -> bilinear sample of the previous frame (special mask)
    -> space-to-depth -> cat -- is defined operation by operation in models/videotools.py / inference/loadedmodel.py; the fused
    kernel (isrAssembleInput) computes the SAME bits, on a previous frame with silhouette edges (where a differently rounded
    coordinate shows as 1e-4 in the value) and flows of a few pixels, against the definition evaluated on the CPU and on the device."""
    from isosurfacesuperresolution_amd import ops
    from isosurfacesuperresolution_amd.inference.flowfill import fill_flow
    from isosurfacesuperresolution_amd.models import VideoTools
    g = torch.Generator().manual_seed(h * 7 + w)
    gb = torch.rand((h, w, 12), generator=g)
    yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    gb[..., 3] = (((yy - 0.45 * h) ** 2 + (xx - 0.55 * w) ** 2) < (0.3 * min(h, w)) ** 2).float()
    gb[..., 8:10] = (gb[..., 8:10] - 0.5) * 0.02
    H, W = 4 * h, 4 * w
    YY, XX = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    inside = (((YY - 0.5 * H) ** 2 + (XX - 0.5 * W) ** 2) < (0.33 * min(H, W)) ** 2).float()
    prev = torch.rand((1, 6, H, W), generator=g)
    prev[:, 0] = inside * 2 - 1                                   # mask -1 / +1 with a hard edge
    prev[:, 1:4] = torch.nn.functional.normalize(prev[:, 1:4] - 0.5, dim=1) * inside
    prev[:, 4:] = prev[:, 4:] * inside
    gb_d, prev_d = gb.cuda(), prev.cuda()
    flow_d = ops.fill_flow_gbuffer(gb_d)
    x = ops.assemble_input(gb_d, flow_d, prev_d, "zero", False)
    torch.cuda.synchronize()
    for dev in ("cpu", "cuda"):
        low = gb.permute(2, 0, 1).unsqueeze(0).to(dev)
        flow = fill_flow(low[:, 8:10], low[:, 3:4] != 0)
        assert torch.equal(flow.cpu(), flow_d.cpu())
        warped = VideoTools.warp_upscale(prev.to(dev), flow, 4, special_mask=True)
        ref = torch.cat((low[:, 3:4] * 2 - 1, low[:, 4:8], VideoTools.flatten_high(warped, 4)), dim=1)
        assert torch.equal(x.cpu(), ref.cpu()), (dev, (x.cpu() - ref.cpu()).abs().max().item())
    # ... while the library-call form of the same warp (F.interpolate + F.grid_sample) is a differently rounded fp32 evaluation: at this
    # size it differs from the definition by the conditioning of the normalised grid, which is why the definition is spelled out
    lib = VideoTools.warp_upscale_library(prev, fill_flow(gb.permute(2, 0, 1).unsqueeze(0)[:, 8:10], gb.permute(2, 0, 1).unsqueeze(0)[:, 3:4] != 0), 4, special_mask=True)
    w64 = VideoTools.warp_upscale(prev.double(), flow_d.cpu().double(), 4, special_mask=True)
    own = VideoTools.flatten_high(w64, 4).float()
    assert (VideoTools.flatten_high(lib, 4) - own).abs().max().item() < 2e-3 and (x.cpu()[:, 5:] - own).abs().max().item() < 2e-3


@pytest.mark.parametrize("h,w,mode,with_prev", [(270, 480, "zero", True), (270, 480, "zero", False), (135, 240, "input", False), (135, 240, "unshaded", False),
                                                 (23, 37, "zero", True), (64, 70, "zero", True)])
def test_packed_assembly_feeds_the_dataflow_trunk_the_same_bits(h, w, mode, with_prev):
    """ops.assemble_input_packed (isrAssembleInputPacked: the input written packed-split into the dataflow trunk's workspace, the trunk's
    own packing pass skipped -- isrTrunkDataflowPrepacked) against ops.assemble_input + the packing pass: channels 0 .. 4 of the fp32
    tensor and the trunk's result are bit-identical (same values through the same split16x), for every initial-image mode, with and
    without a previous frame, at sizes with partial 64-pixel runs; twice in a row (the housekeeping the packing pass did -- progress
    counters, zero units -- is done by the assembly)."""
    import argparse
    from isosurfacesuperresolution_amd import models, ops
    torch.manual_seed(3)
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6,
                               argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10)).cuda().eval()
    convs = net.trunk_convs()
    g = torch.Generator().manual_seed(h + w)
    for trial in range(2):
        gb = torch.rand((h, w, 12), generator=g)
        gb[..., 3] = (gb[..., 3] > 0.4).float()
        gb[..., 8:10] = (gb[..., 8:10] - 0.5) * 0.02
        gb = gb.cuda()
        prev = (torch.rand((1, 6, 4 * h, 4 * w), generator=g) * 2 - 1).cuda() if with_prev else None
        flow = ops.fill_flow_gbuffer(gb) if with_prev else None
        with torch.no_grad():
            x = ops.assemble_input(gb, flow, prev, mode, False)
            assert ops.trunk_supported(x, convs)
            f_ref = ops.trunk_dataflow(x, convs).clone()
            xp = ops.assemble_input_packed(gb, flow, prev, convs, mode, False)
            assert xp is not None and getattr(xp, '_isr_prepacked', None) is not None
            f = ops.trunk_dataflow(xp, convs).clone()
            torch.cuda.synchronize()
            ops.trunk_check()
        assert torch.equal(xp[:, :5], x[:, :5])
        assert torch.equal(f, f_ref), (trial, (f - f_ref).abs().max().item())


def test_a_prepacked_input_refuses_any_route_but_the_dataflow_trunk():
    import argparse
    from isosurfacesuperresolution_amd import models, ops
    torch.manual_seed(3)
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6,
                               argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10)).cuda().eval()
    gb = torch.rand((32, 48, 12)).cuda()
    with torch.no_grad():
        xp = ops.assemble_input_packed(gb, None, None, net.trunk_convs(), "zero", False)
        assert xp is not None
        ops.TRUNK_DATAFLOW = False
        try:
            with pytest.raises(RuntimeError, match="packed-split"):
                net.forward_features(xp)
            assert ops.assemble_input_packed(gb, None, None, net.trunk_convs(), "zero", False) is None     # ... and is not offered then
        finally:
            ops.TRUNK_DATAFLOW = True


def test_pipeline_overlap_matches_back_to_back():
    """frame(origin, next_origin) renders frame t+1 on a side stream under the network of frame t
    (pipeline.py); the frames must be the ones the single-stream sequence produces, bit for bit."""
    from isosurfacesuperresolution_amd import models, volumes as V
    from isosurfacesuperresolution_amd.inference import DirectRenderer, LoadedModel
    from isosurfacesuperresolution_amd.pipeline import SuperResolutionPipeline, default_shading
    opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10)
    torch.manual_seed(3)
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt)
    lm = LoadedModel.from_model(net, "cuda", parameters={"initialImage": "zero"})
    renderer = DirectRenderer()
    renderer.load_dense(V.ejecta(64))
    origins = [V.orbit_camera(k) for k in range(5)]
    outs = []
    for overlap in (False, True):
        pipe = SuperResolutionPipeline(renderer, lm, default_shading("cuda", 30.0), (96, 56))
        pipe.set_static(fov=30.0, isovalue=0.34)
        pipe.frame(origins[0])          # settle the renderer's "last camera" identically for both passes
        pipe.reset()
        seq = []
        for k, o in enumerate(origins):
            nxt = origins[k + 1] if overlap and k + 1 < len(origins) else None
            rgb, raw = pipe.frame(o, nxt)
            seq.append((rgb.clone(), raw.clone(), pipe.gbuffer.clone()))
        torch.cuda.synchronize()
        outs.append(seq)
    for (rgb_a, raw_a, g_a), (rgb_b, raw_b, g_b) in zip(*outs):
        assert torch.equal(g_a, g_b)
        assert torch.equal(raw_a, raw_b)
        assert torch.equal(rgb_a, rgb_b)
    assert outs[0][-1][2][..., 3].sum().item() > 0


def test_strip_super_resolution_is_bit_identical_on_gpu():
    """parallel_sr: strips (with their 24-px halo) computed one after the other reproduce the full-frame network
    output bit for bit on the HIP kernels -- the per-pixel accumulation order does not depend on the tiling."""
    from isosurfacesuperresolution_amd import models, parallel_sr
    from isosurfacesuperresolution_amd.inference import LoadedModel
    from isosurfacesuperresolution_amd.pipeline import default_shading
    opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10)
    torch.manual_seed(5)
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt)
    lm = LoadedModel.from_model(net, "cuda", parameters={"initialImage": "zero"})
    sr = parallel_sr.StripSuperResolution(lm, default_shading("cuda", 30.0))
    g = torch.Generator().manual_seed(8)
    prev = None
    for step in range(2):
        gb = torch.rand(90, 56, 12, generator=g)
        gb[..., 3] = (gb[..., 3] > 0.4).float()
        gb[..., 8:10] = (gb[..., 8:10] - 0.5) * 0.05
        gb = gb.cuda()
        sr.previous = prev
        with torch.no_grad():
            x = sr.network_input(gb)
            full_raw, full_rgb = sr.compute_strip(x, 0, 1)
            for world in (2, 4):
                parts = [sr.compute_strip(x, r, world) for r in range(world)]
                assert torch.equal(torch.cat([p[0] for p in parts], dim=2), full_raw), (step, world)
                assert torch.equal(torch.cat([p[1] for p in parts], dim=2), full_rgb), (step, world)
        prev = full_raw


def test_tile_grid_super_resolution_is_bit_identical_on_gpu():
    """parallel_sr with a (rows x columns) grid of screen tiles: every tile of a 4 x 2 and a 2 x 4 grid -- halo in y
    AND x, column cuts on multiples of 8 -- reproduces its part of the full-frame network output bit for bit on the HIP kernels."""
    from isosurfacesuperresolution_amd import models, parallel_sr
    from isosurfacesuperresolution_amd.inference import LoadedModel
    from isosurfacesuperresolution_amd.pipeline import default_shading
    opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10)
    torch.manual_seed(5)
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt)
    lm = LoadedModel.from_model(net, "cuda", parameters={"initialImage": "zero"})
    sr = parallel_sr.StripSuperResolution(lm, default_shading("cuda", 30.0))
    g = torch.Generator().manual_seed(9)
    prev = None
    h, w = 136, 240
    for step in range(2):
        gb = torch.rand(h, w, 12, generator=g)
        gb[..., 3] = (gb[..., 3] > 0.4).float()
        gb[..., 8:10] = (gb[..., 8:10] - 0.5) * 0.05
        gb = gb.cuda()
        sr.previous = prev
        with torch.no_grad():
            x = sr.network_input(gb)
            full_raw, full_rgb = sr.compute_strip(x, 0, 1)
            for grid in ((4, 2), (2, 4), (1, 3)):
                world = grid[0] * grid[1]
                for r in range(world):
                    raw_t, rgb_t = sr.compute_strip(x, r, world, grid=grid)
                    y0, y1, x0, x1 = parallel_sr.tile_bounds(h, w, grid, r)
                    assert torch.equal(raw_t, full_raw[:, :, 4 * y0:4 * y1, 4 * x0:4 * x1]), (step, grid, r)
                    assert torch.equal(rgb_t, full_rgb[:, :, 4 * y0:4 * y1, 4 * x0:4 * x1]), (step, grid, r)
        prev = full_raw


def test_config4_clips_rendered_here_train_with_temporal_loss():
    """BASELINE config #4 in miniature (tools/config4_cloud.py runs the 512^3 version): a cloud volume, clips of 3
    frames rendered by this package's ray-marcher (low + 4x ground truth with ray-cast AO), 32^2 crops, EnhanceNet
    training steps with the temp-l2 loss and the warped previous-frame recurrence; the loss must fall."""
    from isosurfacesuperresolution_amd import models, losses, train, volumes as V
    from isosurfacesuperresolution_amd.dataset_video import render_clip
    from isosurfacesuperresolution_amd.inference import DirectRenderer
    r = DirectRenderer()
    r.load_dense(V.cloud(64))
    clips = [render_clip(r, [V.orbit_camera(8 * c + k, K=64, distance=1.8, pitch=0.3) for k in range(3)], (64, 40),
                         isovalue=0.30, ao_samples=4, ao_radius=0.05) for c in range(2)]
    lo = torch.from_numpy(np.stack([c[1][:, :, 4:36, 16:48] for c in clips])).cuda()
    fl = torch.from_numpy(np.stack([c[2][:, :, 4:36, 16:48] for c in clips])).cuda()
    hi = torch.from_numpy(np.stack([c[0][:, :, 16:144, 64:192] for c in clips])).cuda()
    assert lo.shape == (2, 3, 5, 32, 32) and hi.shape == (2, 3, 6, 128, 128) and fl.shape == (2, 3, 2, 32, 32)
    assert float((lo[:, :, 0] > 0).float().mean()) > 0.3                       # the crops see the cloud
    assert float(hi[:, :, 5].min()) >= 0 and float(hi[:, :, 5].max()) <= 1      # AO target channel
    opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10,
                             losses="l1:mask:1,l1:ao:1,l1:normal:10,l1:depth:10,temp-l2:color:0.1",
                             lossAO=0.0, lossAmbient=0.1, lossDiffuse=0.9, lossSpecular=0.0)
    torch.manual_seed(124)
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt).cuda()
    crit = losses.LossNetUnshaded('cuda', 5, 6, 128, 16, opt).cuda()
    optim, _ = train.make_optimizer(net)
    hist = [train.train_step(net, crit, optim, (lo, fl, hi), initial_image="zero") for _ in range(5)]
    assert all(np.isfinite(hist)) and hist[-1] < hist[0], hist


@pytest.mark.parametrize("low", [(61, 35), (97, 3), (33, 64)])
def test_pipeline_ragged_sizes_fused_vs_module_path(low):
    """The fused frame path (side-stream render + gate, fused assembly, HIP convs with both tilings, last layer +
    finishing in one launch) against the module-level path at sizes that are multiples of nothing."""
    from isosurfacesuperresolution_amd import models, volumes as V
    from isosurfacesuperresolution_amd.inference import DirectRenderer, LoadedModel
    from isosurfacesuperresolution_amd.pipeline import SuperResolutionPipeline, default_shading
    opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10)
    torch.manual_seed(3)
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt)
    lm = LoadedModel.from_model(net, "cuda", parameters={"initialImage": "zero"})
    r = DirectRenderer()
    r.load_dense(V.ejecta(64))
    outs = []
    for fused in (True, False):
        pipe = SuperResolutionPipeline(r, lm, default_shading("cuda", 30.0), low, fused=fused)
        pipe.set_static(fov=30.0, isovalue=0.34)
        seq = []
        for k in range(3):
            rgb, raw = pipe.frame(V.orbit_camera(k), V.orbit_camera(k + 1) if fused else None)
            seq.append((rgb.clone(), raw.clone()))
        torch.cuda.synchronize()
        outs.append(seq)
    for (rgb_a, raw_a), (rgb_b, raw_b) in zip(*outs):
        assert rgb_a.shape == (1, 3, 4 * low[1], 4 * low[0])
        assert (raw_a - raw_b).abs().max().item() <= 1e-4 and (rgb_a - rgb_b).abs().max().item() <= 1e-4


def test_reference_checkpoint_on_the_hip_kernels(tmp_path):
    """The reference-format checkpoint fixture (tests/golden/make_checkpoint_fixture.py) loaded onto the GPU: the HIP
    convolutions reproduce the reference network's own CPU output within 1e-4."""
    import zipfile
    from isosurfacesuperresolution_amd import inference
    here = os.path.join(os.path.dirname(__file__), "golden")
    with zipfile.ZipFile(os.path.join(here, "ref_checkpoint.zip")) as z:
        z.extractall(tmp_path)
    io = np.load(os.path.join(here, "ref_checkpoint_io.npz"))
    lm = inference.LoadedModel(str(tmp_path / "model_epoch_12.pth"), "cuda", 4)
    with torch.no_grad():
        y, raw = lm.model(torch.from_numpy(io["x"]).cuda())
    assert np.abs(y.cpu().numpy() - io["y"]).max() <= 1e-4 and np.abs(raw.cpu().numpy() - io["raw"]).max() <= 1e-4


def test_graphed_train_step_matches_eager():
    """train.GraphedTrainStep (the whole T-frame step captured in a HIP graph) against the eager train_step: same
    start, same clip batch, 2 steps -> the same weights (up to individual ReLUs that sit within rounding of
    zero, cf. test_enhancenet_gpu_train_step_matches_cpu) and the same loss.  The capture's warm-up steps leave no
    trace: weights and optimizer state after construction are those before it, so n replays = n eager steps."""
    from isosurfacesuperresolution_amd import models, losses, train
    opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10,
                             losses="l1:mask:1,l1:ao:1,l1:normal:10,l1:depth:10,temp-l2:color:0.1",
                             lossAO=0.0, lossAmbient=0.1, lossDiffuse=0.9, lossSpecular=0.0)
    g = torch.Generator(device='cuda').manual_seed(5)
    inp = torch.rand(2, 3, 5, 32, 32, device='cuda', generator=g); inp[:, :, 0] = inp[:, :, 0] * 2 - 1
    flow = (torch.rand(2, 3, 2, 32, 32, device='cuda', generator=g) - 0.5) * 0.05
    tgt = torch.rand(2, 3, 6, 128, 128, device='cuda', generator=g); tgt[:, :, 0] = tgt[:, :, 0] * 2 - 1
    crit = losses.LossNetUnshaded('cuda', 5, 6, 128, 16, opt).cuda()
    nets, finals, last = [], [], []
    for graphed in (False, True):
        torch.manual_seed(11)
        net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt).cuda()
        init = torch.cat([p.detach().reshape(-1).clone() for p in net.parameters()])
        optim, _ = train.make_optimizer(net, lr=1e-4, capturable=graphed)
        if graphed:
            step = train.GraphedTrainStep(net, crit, optim, (inp, flow, tgt), warmup=3, initial_image="zero")
            torch.cuda.synchronize()
            assert torch.equal(torch.cat([p.detach().reshape(-1) for p in net.parameters()]), init)      # warm-up undone
            for st in optim.state.values():
                assert float(st['step'].item()) == 0.0 and st['exp_avg'].abs().max().item() == 0.0
            for _ in range(2):
                l = float(step((inp, flow, tgt)))
            assert all(float(st['step'].item()) == 2.0 for st in optim.state.values())
        else:
            for _ in range(2):
                l = train.train_step(net, crit, optim, (inp, flow, tgt), initial_image="zero")
        torch.cuda.synchronize()
        finals.append(torch.cat([p.detach().reshape(-1) for p in net.parameters()]) - init)
        last.append(l)
    assert finals[0].abs().max().item() > 1e-4
    rel = ((finals[0] - finals[1]).norm() / finals[0].norm()).item()
    assert rel < 2e-2, rel
    assert abs(last[0] - last[1]) <= 2e-2 * abs(last[0]), last


def test_input_assembly_of_a_row_range_equals_those_rows_of_the_whole():
    """ops.assemble_input(rows=(a, b)) (isrAssembleInputRows: a strip rank's rows + halo) writes exactly what the whole-frame
    assembly holds in those rows and leaves the others alone -- with and without a previous frame."""
    import torch
    from isosurfacesuperresolution_amd import ops
    g = torch.Generator().manual_seed(4)
    h, w = 61, 100
    gb = (torch.rand((h, w, 12), generator=g) * 2 - 1).cuda()
    gb[..., 3] = (gb[..., 3] > 0).float()
    flow = ((torch.rand((1, 2, h, w), generator=g) - 0.5) * 0.05).cuda()
    prev = (torch.rand((1, 6, 4 * h, 4 * w), generator=g) * 2 - 1).cuda()
    for fl, pv, mode in ((flow, prev, "zero"), (None, None, "input"), (None, None, "unshaded")):
        whole = ops.assemble_input(gb, fl, pv, mode, False)
        for a, b in ((0, 17), (13, 48), (40, 61), (0, 61)):
            out = torch.full((1, 101, h, w), 7.0, device="cuda")
            part = ops.assemble_input(gb, fl, pv, mode, False, out=out, rows=(a, b))
            assert part is out and torch.equal(part[:, :, a:b], whole[:, :, a:b])
            assert bool((part[:, :, :a] == 7.0).all()) and bool((part[:, :, b:] == 7.0).all())
