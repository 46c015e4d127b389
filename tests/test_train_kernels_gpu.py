"""GPU parity of the training-side elementwise kernels (csrc/sr_train.hip, through the C-ABI of libisr_sr.so):
the x2 bilinear upsampling and the fused LossNetUnshaded, forward and backward, against the PyTorch module path
(the restatement of SuperresolutionNetwork/losses/lossnet_unshaded.py and models/enhancenet.py:116,119)."""
import argparse

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(2, 3, 5, 6), (1, 64, 32, 32), (3, 1, 1, 2), (1, 2, 7, 10), (16, 64, 16, 16), (2, 5, 3, 4), (3, 2, 1, 8), (2, 3, 9, 12),
                                   (2, 3, 5, 7), (1, 2, 4, 1), (1, 1, 3, 5), (2, 64, 17, 33)])   # odd widths: the pair-granular launch
def test_upsample2x_forward_backward_match_torch(shape):
    from isosurfacesuperresolution_amd import ops
    g = torch.Generator().manual_seed(7)
    x = torch.randn(shape, generator=g).cuda().requires_grad_(True)
    xr = x.detach().clone().requires_grad_(True)
    y = ops.bilinear_upsample2x(x)
    yr = F.interpolate(xr, scale_factor=2, mode='bilinear', align_corners=False)
    assert y.shape == yr.shape
    assert (y - yr).abs().max().item() <= 1e-6          # tolerance: fp32 rounding of a 4-term blend on O(1) data
    gy = torch.randn(y.shape, generator=g).cuda()
    y.backward(gy)
    yr.backward(gy)
    assert (x.grad - xr.grad).abs().max().item() <= 1e-5   # 16-term sums, different association than the atomics
    # the adjoint is a gather: bitwise reproducible
    x2 = x.detach().clone().requires_grad_(True)
    ops.bilinear_upsample2x(x2).backward(gy)
    assert torch.equal(x2.grad, x.grad)


def test_upsample2x_refuses_other_dtypes_on_the_gpu():
    """No framework fallback on the GPU: fp32 is what the HIP kernel computes, anything else is an error (as ops.conv3x3)."""
    from isosurfacesuperresolution_amd import ops
    with pytest.raises(TypeError):
        ops.bilinear_upsample2x(torch.zeros(1, 1, 4, 4, dtype=torch.float16, device="cuda"))


def _opt(losses, ao=0.0):
    return argparse.Namespace(upsample='bilinear', losses=losses, lossAO=ao, lossAmbient=0.1, lossDiffuse=0.9, lossSpecular=0.0)


def _clip(n, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    def img():
        t = torch.randn(n, 6, h, w, generator=g)
        t[:, 0] = t[:, 0] * 1.5                      # masks beyond [-1, 1]: exercises the clamps of gate and shading
        t[:, 5] = t[:, 5] * 0.8 + 0.4                # ao below 0 and above 1
        t[:, 4] = t[:, 4] * 0.5 + 0.5
        return t
    gt, pred, prev = img(), img(), img()
    pred[:, 1:4, 5, 7] = 0.0                         # zero normals: the 1e-7 floor of normalize()
    prev[:, 1:4, 6, 9] = 0.0
    return gt.cuda(), pred.cuda(), prev.cuda()


RECIPE = "l1:mask:1,l1:ao:1,l1:normal:10,l1:depth:10,temp-l2:color:0.1"
EVERYTHING = ("mse:mask:0.3,mse:normal:2,mse:ao:0.7,mse:depth:1.5,mse:color:0.9,l1:mask:1,l1:normal:3,l1:ao:0.4,l1:depth:2,"
              "l1:color:0.6,temp-l2:mask:0.2,temp-l2:normal:0.8,temp-l2:ao:0.5,temp-l2:depth:1.1,temp-l2:color:0.1")


@pytest.mark.parametrize("losses,ao,pad,size", [(RECIPE, 0.0, 4, (32, 32)), (EVERYTHING, 0.35, 3, (20, 24)), (EVERYTHING, 1.0, 0, (9, 12)),
                                                ("l1:normal:1", 0.0, 16, (128, 128))])
def test_fused_unshaded_loss_matches_module_path(losses, ao, pad, size):
    from isosurfacesuperresolution_amd import losses as L
    crit = L.LossNetUnshaded('cuda', 5, 6, size[0], pad, _opt(losses, ao)).cuda()
    gt, pred, prev = _clip(3, size[0], size[1], 11)
    out = {}
    for fused in (True, False):
        crit.fused = fused
        p = pred.clone().requires_grad_(True)
        v = prev.clone().requires_grad_(True)
        total, values = crit(gt, p, None, None, v)
        (total * 1.7).backward()                          # an upstream factor other than one
        out[fused] = (total.detach(), values, p.grad, v.grad)
    tf, vf, gpf, gvf = out[True]
    tm, vm, gpm, gvm = out[False]
    assert set(vf) == set(vm)
    for k in vm:
        assert abs(vf[k] - vm[k]) <= 1e-5 * max(1.0, abs(vm[k])), k
    assert abs(tf.item() - tm.item()) <= 1e-5 * max(1.0, abs(tm.item()))
    # gradients: entries are weight / count sized, compare relative to the largest one
    scale = gpm.abs().max().item()
    assert (gpf - gpm).abs().max().item() <= 2e-5 * scale
    if crit.has_temporal_l2_loss:
        assert (gvf - gvm).abs().max().item() <= 2e-5 * max(gvm.abs().max().item(), 1e-12)
    else:
        assert gvf is None or gvf.abs().max().item() == 0.0
    # the border the module zeroes receives no gradient
    if pad:
        assert gpf[:, :, :pad].abs().max().item() == 0.0 and gpf[:, :, :, -pad:].abs().max().item() == 0.0


def test_fused_loss_first_frame_has_no_previous_gradient():
    """Frame 0 compares against the ground truth of frame 0 (mainVideoUnshaded.py:431): prev needs no gradient."""
    from isosurfacesuperresolution_amd import losses as L
    crit = L.LossNetUnshaded('cuda', 5, 6, 32, 4, _opt(RECIPE)).cuda()
    crit.lazy_values = True
    gt, pred, prev = _clip(2, 32, 32, 5)
    p = pred.clone().requires_grad_(True)
    total, values = crit(gt, p, None, None, gt)
    total.backward()
    assert torch.isfinite(p.grad).all()
    assert all(torch.is_tensor(v) and not v.requires_grad for v in values.values())


def test_clip_gradients_with_fused_kernels_match_module_path():
    """Loss and weight gradients of a clip (T = 3 frames, recurrence, BPTT) with the fused loss + hand-written
    upsampling against the PyTorch module path of both.  The upsampled features differ from PyTorch's in the last bit
    (another FMA contraction), so of the ~3.5 M ReLU inputs behind them a handful within 1e-6 of zero switch, each
    moving the weight gradients by O(1e-4) in relative L2 (tools/dbg_train_parity.py: the fused loss alone stays at
    8e-7, the distance between two runs of the module path with its atomics; cf.
    test_conv_gpu.py::test_enhancenet_gpu_train_step_matches_cpu)."""
    from isosurfacesuperresolution_amd import models, losses as L, train, ops
    opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=2, losses=RECIPE,
                             lossAO=0.0, lossAmbient=0.1, lossDiffuse=0.9, lossSpecular=0.0)
    g = torch.Generator().manual_seed(3)
    B, T = 2, 3
    inp = torch.rand(B, T, 5, 16, 16, generator=g).cuda(); inp[:, :, 0] = inp[:, :, 0] * 2 - 1
    flow = ((torch.rand(B, T, 2, 16, 16, generator=g) - 0.5) * 0.05).cuda()
    tgt = torch.rand(B, T, 6, 64, 64, generator=g).cuda(); tgt[:, :, 0] = tgt[:, :, 0] * 2 - 1
    results = []
    for fused in (True, False):
        torch.manual_seed(124)
        net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt).cuda()
        crit = L.LossNetUnshaded('cuda', 5, 6, 64, 8, opt).cuda()
        crit.fused = fused
        saved = ops.bilinear_upsample2x
        if not fused:
            ops.bilinear_upsample2x = lambda x: F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)
        try:
            loss, loss_sum = train.clip_loss(net, crit, inp, flow, tgt, initial_image="zero")
            loss.backward()
        finally:
            ops.bilinear_upsample2x = saved
        results.append((loss_sum.item(), [p.grad.detach().clone() for p in net.parameters()]))
    (lf, gf), (lm, gm) = results
    assert abs(lf - lm) <= 1e-5 * max(1.0, abs(lm))
    errs = [((a - b).norm() / b.norm()).item() for a, b in zip(gf, gm)]
    assert max(errs) <= 1e-2, max(errs)
    assert sorted(errs)[len(errs) // 2] <= 1e-3, sorted(errs)[-8:]


@pytest.mark.parametrize("case", [(3, 2, 64, 64, 12, 20), (5, 1, 101, 64, 9, 7), (2, 3, 64, 6, 8, 40), (33, 1, 8, 8, 4, 4),
                                  (2, 4, 16, 6, 64, 64), (2, 4, 16, 20, 62, 61)])     # the last two: K-split variant
def test_weight_gradient_over_segments_matches_fp64(case):
    """isrConv3x3WeightGradSegments: dw / db summed over several (x, gz) pairs in one pass (33 pairs: two passes)."""
    from isosurfacesuperresolution_amd import ops
    segs, n, cin, cout, h, w = case
    g = torch.Generator().manual_seed(segs * 100 + cin)
    xs = [torch.randn(n, cin, h, w, generator=g) for _ in range(segs)]
    gzs = [torch.randn(n, cout, h, w, generator=g) for _ in range(segs)]
    weight = torch.zeros(cout, cin, 3, 3).cuda()
    gw, gb = ops._weight_grad([t.cuda() for t in xs], [t.cuda() for t in gzs], weight, True)
    wref = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    bref = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    for x, gz in zip(xs, gzs):
        (F.conv2d(x.double(), wref, bref, padding=1) * gz.double()).sum().backward()
    scale = wref.grad.abs().max().item()
    assert (gw.cpu().double() - wref.grad).abs().max().item() <= 1e-5 * scale     # fp32 sums of n*h*w*segs products
    assert (gb.cpu().double() - bref.grad).abs().max().item() <= 1e-5 * bref.grad.abs().max().item()


def test_deferred_weight_gradients_equal_per_frame_accumulation():
    """ops.deferred_weight_gradients (one weight-gradient pass per layer over the clip's frames) against autograd's
    per-frame weight gradients + accumulation: the forward passes are identical, only the summation order differs."""
    from isosurfacesuperresolution_amd import models, losses as L, train, ops
    opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=2, losses=RECIPE,
                             lossAO=0.0, lossAmbient=0.1, lossDiffuse=0.9, lossSpecular=0.0)
    g = torch.Generator().manual_seed(9)
    B, T = 2, 4
    inp = torch.rand(B, T, 5, 16, 16, generator=g).cuda(); inp[:, :, 0] = inp[:, :, 0] * 2 - 1
    flow = ((torch.rand(B, T, 2, 16, 16, generator=g) - 0.5) * 0.05).cuda()
    tgt = torch.rand(B, T, 6, 64, 64, generator=g).cuda(); tgt[:, :, 0] = tgt[:, :, 0] * 2 - 1
    torch.manual_seed(124)
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt).cuda()
    crit = L.LossNetUnshaded('cuda', 5, 6, 64, 8, opt).cuda()
    grads = []
    for deferred in (False, True):
        net.zero_grad(set_to_none=True)
        loss, _ = train.clip_loss(net, crit, inp, flow, tgt, initial_image="zero")
        if deferred:
            train.backward(loss)
        else:
            loss.backward()
        grads.append([p.grad.detach().clone() for p in net.parameters()])
    assert all(a is not None for a in grads[1])
    for (name, _), a, b in zip(net.named_parameters(), grads[0], grads[1]):
        assert ((a - b).norm() / a.norm()).item() <= 1e-5, name
    # accumulation into existing gradients: a second deferred backward doubles them
    loss, _ = train.clip_loss(net, crit, inp, flow, tgt, initial_image="zero")
    train.backward(loss)
    for p, b in zip(net.parameters(), grads[1]):
        assert ((p.grad - 2 * b).norm() / b.norm()).item() <= 1e-5
    with pytest.raises(RuntimeError):
        with ops.deferred_weight_gradients():
            with ops.deferred_weight_gradients():
                pass


@pytest.mark.parametrize("shape", [(2, 64, 12, 20), (1, 64, 9, 7), (3, 64, 32, 32)])
def test_residual_block_function_matches_fp64(shape):
    """ops.residual_block (one autograd node, gated data gradient + fused skip gradient) vs fp64 PyTorch."""
    from isosurfacesuperresolution_amd import ops
    g = torch.Generator().manual_seed(21)
    n, c, h, w = shape
    x = torch.randn(shape, generator=g)
    ws = [torch.randn(c, c, 3, 3, generator=g) * 0.05 for _ in range(2)]
    bs = [torch.randn(c, generator=g) * 0.1 for _ in range(2)]
    gy = torch.randn(shape, generator=g)
    dev = [t.cuda().requires_grad_(True) for t in (x, ws[0], bs[0], ws[1], bs[1])]
    y = ops.residual_block(*dev)
    y.backward(gy.cuda())
    ref = [t.double().requires_grad_(True) for t in (x, ws[0], bs[0], ws[1], bs[1])]
    yr = ref[0] + F.conv2d(F.relu(F.conv2d(ref[0], ref[1], ref[2], padding=1)), ref[3], ref[4], padding=1)
    yr.backward(gy.double())
    assert (y.detach().cpu().double() - yr.detach()).abs().max().item() <= 1e-4
    for a, b in zip(dev, ref):
        assert (a.grad.cpu().double() - b.grad).abs().max().item() <= 1e-4 * max(1.0, b.grad.abs().max().item())
    # and the same numbers as the two separate conv3x3 nodes it replaces
    sep = [t.detach().clone().requires_grad_(True) for t in dev]
    ys = ops.conv3x3(ops.conv3x3(sep[0], sep[1], sep[2], act='relu'), sep[3], sep[4], residual=sep[0])
    ys.backward(gy.cuda())
    assert torch.equal(ys, y)
    for a, b in zip(dev, sep):
        assert (a.grad - b.grad).abs().max().item() <= 1e-5 * max(1.0, b.grad.abs().max().item())


@pytest.mark.parametrize("b,h,w", [(2, 8, 12), (3, 16, 16), (1, 5, 9)])
def test_recurrent_input_matches_module_path(b, h, w):
    """ops.recurrent_input (clamp / normalize + warp_upscale + flatten_high + cat in one launch, scatter backward)
    against the PyTorch module path of train.clip_loss, on strided frame views of a [B, T, ..] clip."""
    from isosurfacesuperresolution_amd import ops
    from isosurfacesuperresolution_amd.models import VideoTools
    from isosurfacesuperresolution_amd.utils import ScreenSpaceShading
    g = torch.Generator().manual_seed(b * 31 + h)
    raw = torch.randn(b, 6, 4 * h, 4 * w, generator=g)
    raw[:, 0] *= 1.5
    raw[:, 4:6] = raw[:, 4:6] * 0.6 + 0.5
    raw[:, 1:4, 3, 5] = 0.0                                     # a zero normal
    clip_in = torch.randn(b, 3, 5, h, w, generator=g).cuda()
    clip_flow = ((torch.rand(b, 3, 2, h, w, generator=g) - 0.5) * 0.4).cuda()     # large enough to leave the image
    inp, flow = clip_in[:, 1], clip_flow[:, 2]
    gn = torch.randn(b, 101, h, w, generator=g).cuda()
    gw = torch.randn(b, 6, 4 * h, 4 * w, generator=g).cuda()

    r1 = raw.cuda().requires_grad_(True)
    netin, warped = ops.recurrent_input(r1, inp, flow)
    ((netin * gn).sum() + (warped * gw).sum()).backward()

    r2 = raw.cuda().requires_grad_(True)
    prev = torch.cat([torch.clamp(r2[:, 0:1], -1, +1), ScreenSpaceShading.normalize(r2[:, 1:4], dim=1),
                      torch.clamp(r2[:, 4:5], 0, +1), torch.clamp(r2[:, 5:6], 0, +1)], dim=1)
    warped_ref = VideoTools.warp_upscale(prev, flow, 4, special_mask=True)
    netin_ref = torch.cat((inp, VideoTools.flatten_high(warped_ref, 4)), dim=1)
    ((netin_ref * gn).sum() + (warped_ref * gw).sum()).backward()

    assert (warped - warped_ref).abs().max().item() <= 2e-5     # sample positions differ in the last bit of the grid
    assert (netin - netin_ref).abs().max().item() <= 2e-5
    assert torch.equal(netin[:, 0:5], inp)
    scale = r2.grad.abs().max().item()
    assert (r1.grad - r2.grad).abs().max().item() <= 1e-4 * scale
    # only one of the two gradients present
    r3 = raw.cuda().requires_grad_(True)
    netin3, _ = ops.recurrent_input(r3, inp, flow)
    (netin3 * gn).sum().backward()
    assert torch.isfinite(r3.grad).all()


def test_clip_gradients_inside_a_hip_graph_are_reproduced_by_every_replay():
    """Forward + backward of a clip captured in a HIP graph: every replay (not only the first) must give the eager
    gradients.  Guards state carried between replays -- a captured hipMemsetAsync that did not take effect on replay
    once left the scatter buffer of the recurrent input's backward un-zeroed from the second replay on."""
    from isosurfacesuperresolution_amd import models, losses as L, ops, train
    opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=3, losses=RECIPE,
                             lossAO=0.0, lossAmbient=0.1, lossDiffuse=0.9, lossSpecular=0.0)
    B, T = 2, 5                                           # 5 frames of 128^2: the K-split weight gradient of the output layer
    g = torch.Generator(device='cuda').manual_seed(1)
    inp = torch.rand(B, T, 5, 32, 32, device='cuda', generator=g); inp[:, :, 0] = inp[:, :, 0] * 2 - 1
    flow = (torch.rand(B, T, 2, 32, 32, device='cuda', generator=g) - 0.5) * 0.05
    tgt = torch.rand(B, T, 6, 128, 128, device='cuda', generator=g); tgt[:, :, 0] = tgt[:, :, 0] * 2 - 1
    torch.manual_seed(124)
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt).cuda()
    crit = L.LossNetUnshaded('cuda', 5, 6, 128, 16, opt).cuda()

    def grads():
        net.zero_grad(set_to_none=True)
        loss, total = train.clip_loss(net, crit, inp, flow, tgt, initial_image="zero")
        train.backward(loss)
        return total

    ref_loss = grads().item()
    ref = [p.grad.detach().clone() for p in net.parameters()]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        grads()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    net.zero_grad(set_to_none=True)
    with ops.graph_capture(graph):
        total = grads()
    for replay in range(3):
        graph.replay()
        torch.cuda.synchronize()
        assert abs(total.item() - ref_loss) <= 1e-5 * abs(ref_loss)
        for (name, p), r in zip(net.named_parameters(), ref):
            assert ((p.grad - r).norm() / r.norm()).item() <= 1e-3, (replay, name)


def test_mixed_precision_training_mode_tracks_the_fp32_step():
    """ops.TRAIN_BF16 (forward / data-gradient convolutions with bf16 MFMA operands, everything else fp32) is opt-in
    and not a parity mode: its loss and weight gradients must stay close to the fp32 step's (bf16 operand rounding,
    2^-9 relative per product), and switching it off must give the fp32 numbers back."""
    from isosurfacesuperresolution_amd import models, losses as L, train, ops
    opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=3, losses=RECIPE,
                             lossAO=0.0, lossAmbient=0.1, lossDiffuse=0.9, lossSpecular=0.0)
    g = torch.Generator().manual_seed(13)
    B, T = 2, 3
    inp = torch.rand(B, T, 5, 32, 32, generator=g).cuda(); inp[:, :, 0] = inp[:, :, 0] * 2 - 1
    flow = ((torch.rand(B, T, 2, 32, 32, generator=g) - 0.5) * 0.05).cuda()
    tgt = torch.rand(B, T, 6, 128, 128, generator=g).cuda(); tgt[:, :, 0] = tgt[:, :, 0] * 2 - 1
    torch.manual_seed(124)
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt).cuda()
    crit = L.LossNetUnshaded('cuda', 5, 6, 128, 16, opt).cuda()

    def grads():
        net.zero_grad(set_to_none=True)
        loss, total = train.clip_loss(net, crit, inp, flow, tgt, initial_image="zero")
        train.backward(loss)
        return total.item(), [p.grad.detach().clone() for p in net.parameters()]

    l32, g32 = grads()
    ops.TRAIN_BF16 = True
    try:
        l16, g16 = grads()
    finally:
        ops.TRAIN_BF16 = False
    l32b, g32b = grads()
    # (not bitwise: the scatter of the warp's backward uses float atomics)
    assert l32b == l32 and all(((a - b).norm() / b.norm()).item() <= 1e-5 for a, b in zip(g32b, g32))
    assert abs(l16 - l32) <= 2e-2 * abs(l32), (l16, l32)
    errs = [((a - b).norm() / b.norm()).item() for a, b in zip(g16, g32)]
    assert max(errs) <= 0.15, max(errs)
    assert sorted(errs)[len(errs) // 2] <= 5e-2, sorted(errs)[-6:]


@pytest.mark.parametrize("case", [(3, 4, 64, 64, 64, 64), (2, 8, 101, 64, 32, 64), (1, 9, 64, 96, 62, 60)])
def test_bf16_weight_gradient_is_the_same_sum(case):
    """isrConv3x3WeightGradSegmentsBf16 (mixed-precision mode): exactly the fp32 sums when the operands are
    representable in bf16 (products exact, fp32 accumulation), bf16 rounding otherwise; bias gradient from fp32 values."""
    from isosurfacesuperresolution_amd import ops
    segs, n, cin, cout, h, w = case
    g = torch.Generator().manual_seed(segs * 17 + cin)
    xs = [torch.randn(n, cin, h, w, generator=g) for _ in range(segs)]
    gzs = [torch.randn(n, cout, h, w, generator=g) for _ in range(segs)]
    weight = torch.zeros(cout, cin, 3, 3).cuda()

    def reference(xl, gl):
        wref = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
        bref = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
        for x, gz in zip(xl, gl):
            (F.conv2d(x.double(), wref, bref, padding=1) * gz.double()).sum().backward()
        return wref.grad, bref.grad

    ops.TRAIN_BF16 = True
    try:
        xq, gq = [t.bfloat16().float() for t in xs], [t.bfloat16().float() for t in gzs]
        gw, gb = ops._weight_grad([t.cuda() for t in xq], [t.cuda() for t in gq], weight, True)
        wr, br = reference(xq, gq)
        assert (gw.cpu().double() - wr).abs().max().item() <= 1e-5 * wr.abs().max().item()
        assert (gb.cpu().double() - br).abs().max().item() <= 1e-5 * br.abs().max().item()
        gw, gb = ops._weight_grad([t.cuda() for t in xs], [t.cuda() for t in gzs], weight, True)
    finally:
        ops.TRAIN_BF16 = False
    wr, br = reference(xs, gzs)
    assert ((gw.cpu().double() - wr).norm() / wr.norm()).item() <= 1e-2
    assert (gb.cpu().double() - br).abs().max().item() <= 1e-5 * br.abs().max().item()       # fp32 path


def test_backward_is_correct_when_large_planes_would_be_padded():
    """Inference outputs get padded channel planes once a plane reaches 4 MiB (ops.empty_planes); the autograd paths
    hand raw pointers of saved outputs to kernels that index [N, C, H, W] flat, so THEIR outputs must stay packed.
    With the padding forced at every size: gradients of a conv + ReLU, a residual block and a leaky conv vs fp64."""
    from isosurfacesuperresolution_amd import ops
    g = torch.Generator().manual_seed(4)
    x = torch.rand(2, 64, 24, 40, generator=g) * 2 - 1
    w1 = (torch.rand(64, 64, 3, 3, generator=g) * 2 - 1) / 24.0
    w2 = (torch.rand(64, 64, 3, 3, generator=g) * 2 - 1) / 24.0
    b1, b2 = torch.rand(64, generator=g) - 0.5, torch.rand(64, generator=g) - 0.5
    gy = torch.rand(2, 64, 24, 40, generator=g) * 2 - 1

    def run(dev, dt):
        t = [v.to(dev).to(dt).requires_grad_() for v in (x, w1, b1, w2, b2)]
        xx, a1, c1, a2, c2 = t
        if dev == "cpu":
            y = F.relu(F.conv2d(xx, a1, c1, padding=1))
            y = y + F.conv2d(F.relu(F.conv2d(y, a2, c2, padding=1)), a1, c1, padding=1)
            y = F.leaky_relu(F.conv2d(y, a2, c2, padding=1), 0.1)
        else:
            y = ops.conv3x3(xx, a1, c1, act='relu')
            y = ops.residual_block(y, a2, c2, a1, c1)
            y = ops.conv3x3(y, a2, c2, act='leaky', slope=0.1)
            assert y.is_contiguous()
        y.backward(gy.to(dev).to(dt))
        return [v.grad.detach().cpu().double() for v in t]

    old = ops.plane_pad
    ops.plane_pad = lambda h, w: 52
    try:
        got = run("cuda", torch.float32)
    finally:
        ops.plane_pad = old
    ref = run("cpu", torch.float64)
    for a, b in zip(ref, got):
        assert ((a - b).norm() / a.norm()).item() <= 1e-5


def test_deferred_weight_gradients_respect_parameter_hooks():
    """Deferred weight gradients bypass autograd's accumulation; a parameter with a tensor hook keeps the ordinary
    path, so the hook fires with the gradient and the result is unchanged."""
    from isosurfacesuperresolution_amd import ops
    g = torch.Generator().manual_seed(8)
    x = (torch.rand(2, 64, 16, 32, generator=g) * 2 - 1).cuda()
    w = ((torch.rand(64, 64, 3, 3, generator=g) * 2 - 1) / 24.0).cuda().requires_grad_()
    b = (torch.rand(64, generator=g) - 0.5).cuda().requires_grad_()
    seen = []
    with ops.deferred_weight_gradients():
        ops.conv3x3(x, w, b, act='relu').square().sum().backward()
    deferred = w.grad.clone()
    w.grad = None; b.grad = None
    h = w.register_hook(lambda grad: seen.append(grad.clone()))
    with ops.deferred_weight_gradients():
        ops.conv3x3(x, w, b, act='relu').square().sum().backward()
        assert len(seen) == 1                      # fired inside backward, not skipped
    h.remove()
    assert torch.allclose(seen[0], deferred, rtol=1e-5, atol=1e-6) and torch.allclose(w.grad, deferred, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("gz_scale", [1.0, 3e-6, 2e3])
@pytest.mark.parametrize("case", [(10, 16, 64, 64, 32, 32), (3, 12, 101, 64, 64, 64), (2, 16, 64, 64, 128, 128), (2, 16, 64, 6, 128, 128)])
def test_split_operand_weight_gradient_matches_fp64(case, gz_scale):
    """isrConv3x3WeightGradSegmentsSplit (three fp16 MFMAs per product on (hi, lo) operand pairs, gz scaled by a power of
    two from its maximum) against fp64, next to the exact fp32 kernel on the same data: as close to fp64 as the fp32
    kernel (within 2x), whatever the magnitude of the gradients -- 3e-6 would be flushed by a plain fp16 operand."""
    from isosurfacesuperresolution_amd import ops
    segs, n, cin, cout, h, w = case
    g = torch.Generator().manual_seed(segs * 100 + cin)
    xs = [torch.relu(torch.randn(n, cin, h, w, generator=g)) for _ in range(segs)]          # activations: half of them exact zeros
    gzs = [torch.randn(n, cout, h, w, generator=g) * gz_scale * (0.02 + torch.rand(1, cout, 1, 1, generator=g)) for _ in range(segs)]
    weight = torch.zeros(cout, cin, 3, 3).cuda()
    assert n * segs * ((h + 3) // 4) * ((w + 31) // 32) >= 1024                             # the route that picks the split kernel
    # fp64 reference through the equivalent correlation: dw = conv(x^T, gz^T) summed over the pairs
    wref = torch.zeros(cout, cin, 3, 3, dtype=torch.float64)
    bref = torch.zeros(cout, dtype=torch.float64)
    for x, gz in zip(xs, gzs):
        xd, gd = x.double().cuda(), gz.double().cuda()
        # dw[co][ci][ky][kx] = sum_{n,y,x} gz[n][co][y][x] * xpad[n][ci][y+ky][x+kx]
        dw = F.conv2d(F.pad(xd, (1, 1, 1, 1)).transpose(0, 1), gd.transpose(0, 1)).transpose(0, 1)
        wref += dw.cpu()
        bref += gd.sum(dim=(0, 2, 3)).cpu()
    err = {}
    old = ops.TRAIN_SPLIT
    try:
        for mode in (True, False):
            ops.TRAIN_SPLIT = mode
            gw, gb = ops._weight_grad([t.cuda() for t in xs], [t.cuda() for t in gzs], weight, True)
            err[mode] = ((gw.cpu().double() - wref).abs().max() / wref.abs().max()).item()
            assert (gb.cpu().double() - bref).abs().max().item() <= 1e-5 * bref.abs().max().item()
    finally:
        ops.TRAIN_SPLIT = old
    assert err[True] != err[False]                                                          # two different kernels did run
    assert err[True] <= 2.0 * err[False] + 5e-7, err
    assert err[True] <= 1e-5, err


@pytest.mark.parametrize("case", [(10, 16, 64, 64, 32, 32), (3, 12, 101, 64, 64, 64), (2, 16, 64, 64, 128, 128), (2, 16, 64, 6, 128, 128),
                                  (5, 7, 64, 64, 36, 40), (1, 3, 64, 64, 130, 68)])
def test_staged_weight_gradient_kernel_is_bit_identical_to_the_one_wave_form(case, diag_lib):
    """conv3x3_wgrad_split2_kernel (staging on its own waves, half-tile pipeline) walks the tiles of conv3x3_wgrad_split_kernel in
    the same order: the weight gradients are EQUAL bit for bit (odd tile counts, ragged rows and columns, partial channel groups
    included); the bias gradient groups its fp32 sums differently and is equal to rounding."""
    import ctypes
    from isosurfacesuperresolution_amd import ops
    lib = ops._sr()
    lib.isrDebugSetWgradSplitForm.argtypes = [ctypes.c_int]
    lib.isrDebugWgradSplitForm.restype = ctypes.c_int
    segs, n, cin, cout, h, w = case
    g = torch.Generator().manual_seed(segs * 7 + h)
    xs = [torch.relu(torch.randn(n, cin, h, w, generator=g)).cuda() for _ in range(segs)]
    gzs = [(torch.randn(n, cout, h, w, generator=g) * 1e-3).cuda() for _ in range(segs)]
    weight = torch.zeros(cout, cin, 3, 3).cuda()
    default = lib.isrDebugWgradSplitForm()
    old = ops.TRAIN_SPLIT
    out = {}
    try:
        ops.TRAIN_SPLIT = True
        for form in (1, 2):
            lib.isrDebugSetWgradSplitForm(form)
            fn = lib.isrConv3x3WeightGradSegmentsSplit
            dw = torch.full_like(weight, 7.0)
            db = torch.full((cout,), 7.0, device="cuda")
            ws = ops._wgrad_workspace(weight.device, lib.isrConvWeightGradWorkspace(n, cin, h, w, cout))
            px = (ctypes.c_void_p * segs)(*[t.data_ptr() for t in xs])
            pg = (ctypes.c_void_p * segs)(*[t.data_ptr() for t in gzs])
            assert fn(px, pg, segs, ops._ptr(dw), ops._ptr(db), ops._ptr(ws), n, cin, h, w, cout, ops._stream()) == 0
            torch.cuda.synchronize()
            out[form] = (dw.clone(), db.clone())
    finally:
        lib.isrDebugSetWgradSplitForm(default)
        ops.TRAIN_SPLIT = old
    assert default == 2
    assert torch.equal(out[1][0], out[2][0]), (out[1][0] - out[2][0]).abs().max().item()
    assert out[1][0].abs().max().item() > 0
    assert (out[1][1] - out[2][1]).abs().max().item() <= 1e-5 * out[1][1].abs().max().item()


@pytest.mark.parametrize("shape", [(16, 32, 32), (16, 31, 32), (40, 9, 28), (130, 2, 4), (8, 32, 32)])
def test_fused_small_image_residual_block_is_bit_identical_to_two_launches(shape):
    """csrc/sr_conv_block2.h (one launch for both convolutions of a residual block of a batch of small images, forward and data
    gradient; the intermediate's halo rows recomputed) against the two-launch path: output, input gradient, and -- through the
    tensors handed to the deferred weight gradients -- every parameter gradient EQUAL bit for bit; and close to a float64 block."""
    from isosurfacesuperresolution_amd import ops
    n, h, w = shape
    g = torch.Generator().manual_seed(n * 100 + h)
    x0 = ((torch.rand(n, 64, h, w, generator=g) - 0.3) * 2).cuda()
    ws = [((torch.rand(64, 64, 3, 3, generator=g) - 0.5) * 0.1).cuda().requires_grad_() for _ in range(2)]
    bs = [((torch.rand(64, generator=g) - 0.5) * 0.2).cuda().requires_grad_() for _ in range(2)]
    gy = ((torch.rand(n, 64, h, w, generator=g) - 0.5) * 1e-2).cuda()
    assert ops._block2_supported(x0, ws[0], ws[1])
    out = {}
    old = ops.TRAIN_BLOCK2
    try:
        for mode in (True, False):
            ops.TRAIN_BLOCK2 = mode
            x = x0.clone().requires_grad_()
            y = ops.residual_block(x, ws[0], bs[0], ws[1], bs[1])
            grads = torch.autograd.grad(y, [x] + ws + bs, gy)
            out[mode] = [y.detach()] + [t.detach() for t in grads]
    finally:
        ops.TRAIN_BLOCK2 = old
    names = ["y", "gx", "gw1", "gw2", "gb1", "gb2"]
    for name, a, b in zip(names, out[True], out[False]):
        assert torch.equal(a, b), "%s: %g" % (name, (a - b).abs().max().item())
    xd = x0.double().requires_grad_()
    wd = [t.detach().double().requires_grad_() for t in ws]
    bd = [t.detach().double().requires_grad_() for t in bs]
    yd = xd + F.conv2d(F.relu(F.conv2d(xd, wd[0], bd[0], padding=1)), wd[1], bd[1], padding=1)
    gd = torch.autograd.grad(yd, [xd] + wd + bd, gy.double())
    for name, a, ref in zip(names, out[True], [yd.detach()] + list(gd)):
        assert (a.double() - ref).abs().max().item() <= 2e-5 * max(1e-3, ref.abs().max().item()), name


def test_weight_gradient_scale_from_the_producers_maxima_is_bit_identical():
    """Inside deferred_weight_gradients() the fused small-image block leaves max |output| of its two data gradients in pool words
    (one per wave) and the split-operand weight gradient takes its gz scale from those words instead of a pass over gz
    (ops.GMAX_FROM_PRODUCERS): every parameter gradient of a ten-frame clip through conv(relu) -> three residual blocks ->
    conv(relu) -> conv EQUAL bit for bit with the switch off, and the tagged route did run."""
    from isosurfacesuperresolution_amd import ops
    g = torch.Generator().manual_seed(5)
    mk = lambda *s, k=0.1: ((torch.rand(*s, generator=g) - 0.5) * k).cuda().requires_grad_()
    w0, b0 = mk(64, 12, 3, 3), mk(64, k=0.2)
    blocks = [(mk(64, 64, 3, 3), mk(64, k=0.2), mk(64, 64, 3, 3), mk(64, k=0.2)) for _ in range(3)]
    w3, b3, w4, b4 = mk(64, 64, 3, 3), mk(64, k=0.2), mk(64, 64, 3, 3), mk(64, k=0.2)
    params = [w0, b0] + [t for blk in blocks for t in blk] + [w3, b3, w4, b4]
    frames = [((torch.rand(16, 12, 32, 32, generator=g) - 0.4)).cuda() for _ in range(10)]
    calls = {"max": 0}
    lib = ops._sr()
    real = lib.isrConv3x3WeightGradSegmentsSplitMax

    def counted(*a):
        calls["max"] += 1 if a[2] is not None else 0        # a[2]: the maxima's addresses
        return real(*a)

    out = {}
    old = ops.GMAX_FROM_PRODUCERS
    try:
        for mode in (True, False):
            ops.GMAX_FROM_PRODUCERS = mode
            for t in params:
                t.grad = None
            lib.isrConv3x3WeightGradSegmentsSplitMax = counted
            try:
                with ops.deferred_weight_gradients():
                    total = 0.0
                    for x in frames:
                        f = ops.conv3x3(x, w0, b0, act='relu')
                        for blk in blocks:
                            f = ops.residual_block(f, *blk)
                        f = ops.conv3x3(ops.conv3x3(f, w3, b3, act='relu'), w4, b4)
                        total = total + (f * f).sum() * 1e-4
                    total.backward()
            finally:
                lib.isrConv3x3WeightGradSegmentsSplitMax = real
            torch.cuda.synchronize()
            out[mode] = [t.grad.clone() for t in params]
            if mode:
                assert calls["max"] == 5, calls          # conv1 of every block, conv2 of all but the last (whose gz is a plain layer's data gradient)
                calls["max"] = 0
            else:
                assert calls["max"] == 0
    finally:
        ops.GMAX_FROM_PRODUCERS = old
    for k, (a, b) in enumerate(zip(out[True], out[False])):
        assert torch.equal(a, b), "parameter %d: %g" % (k, (a - b).abs().max().item())
        assert a.abs().max().item() > 0


def test_slab_reduction_into_grad_is_bit_identical():
    """ops.WGRAD_INTO_GRAD: the deferred layers' slab reductions add straight into .grad (isrSetWeightGradAccumulate), against
    `grad += dw` per parameter -- with gradients already present (second clip accumulates onto the first) and absent."""
    from isosurfacesuperresolution_amd import ops
    g = torch.Generator().manual_seed(9)
    mk = lambda *s, k=0.1: ((torch.rand(*s, generator=g) - 0.5) * k).cuda().requires_grad_()
    w0, b0 = mk(64, 12, 3, 3), mk(64, k=0.2)
    blocks = [(mk(64, 64, 3, 3), mk(64, k=0.2), mk(64, 64, 3, 3), None) for _ in range(2)]
    w3 = mk(6, 64, 3, 3)
    params = [w0, b0] + [t for blk in blocks for t in blk if t is not None] + [w3]
    clips = [[((torch.rand(16, 12, 32, 32, generator=g) - 0.4)).cuda() for _ in range(10)] for _ in range(2)]
    out = {}
    old = ops.WGRAD_INTO_GRAD
    try:
        for mode in (True, False):
            ops.WGRAD_INTO_GRAD = mode
            for t in params:
                t.grad = None
            for frames in clips:                                        # no zero_grad in between: the second pass accumulates
                with ops.deferred_weight_gradients():
                    total = 0.0
                    for x in frames:
                        f = ops.conv3x3(x, w0, b0, act='relu')
                        for blk in blocks:
                            f = ops.residual_block(f, *blk)
                        f = ops.conv3x3(f, w3, None)
                        total = total + (f * f).sum() * 1e-4
                    total.backward()
            torch.cuda.synchronize()
            out[mode] = [t.grad.clone() for t in params]
    finally:
        ops.WGRAD_INTO_GRAD = old
    for k, (a, b) in enumerate(zip(out[True], out[False])):
        assert torch.equal(a, b), "parameter %d: %g" % (k, (a - b).abs().max().item())
        assert a.abs().max().item() > 0


def test_relu_backward_folded_into_the_consumers_data_gradient():
    """conv(relu) -> conv(relu) -> conv: with ops.GATE_FUSION the data gradient of a layer whose input is a ReLU conv3x3's
    output applies that ReLU's backward in its epilogue and the producer skips its isrActBackward; every gradient must be
    the one of the unfused graph (bit for bit: a select instead of a multiplication by 0 / 1), also when the ReLU output
    has a second consumer (autograd then sums two gradients and the producer gates the sum as before)."""
    from isosurfacesuperresolution_amd import ops
    g = torch.Generator().manual_seed(77)
    x0 = (torch.rand(2, 64, 64, 64, generator=g) - 0.5).cuda()
    ws = [((torch.rand(co, ci, 3, 3, generator=g) - 0.5) * 0.1).cuda() for co, ci in ((64, 64), (64, 64), (6, 64))]
    bs = [(torch.rand(w.shape[0], generator=g) - 0.5).cuda() for w in ws]
    tgt = torch.rand(2, 6, 64, 64, generator=g).cuda()
    lib = ops._sr()
    calls = {"n": 0}
    real = lib.isrActBackward

    def run(fuse, second_consumer):
        ops.GATE_FUSION = fuse
        x = x0.clone().requires_grad_(True)
        w = [t.clone().requires_grad_(True) for t in ws]
        b = [t.clone().requires_grad_(True) for t in bs]
        y1 = ops.conv3x3(x, w[0], b[0], act='relu')
        y2 = ops.conv3x3(y1, w[1], b[1], act='relu')
        y3 = ops.conv3x3(y2, w[2], b[2])
        loss = ((y3 - tgt) ** 2).mean()
        if second_consumer:
            loss = loss + (y2 * 0.001).sum()
        loss.backward()
        return [x.grad] + [t.grad for t in w] + [t.grad for t in b]

    old = ops.GATE_FUSION
    try:
        for second in (False, True):
            ref = run(False, second)
            got = run(True, second)
            for a, c in zip(ref, got):
                assert torch.equal(a, c), second
        # the fused graph really launches fewer ReLU-backward kernels: count them through the profile-free path
        import ctypes

        class Counter:
            def __init__(self, fn):
                self.fn, self.n = fn, 0
                self.argtypes, self.restype = fn.argtypes, fn.restype

            def __call__(self, *a):
                self.n += 1
                return self.fn(*a)
        for fuse, expect in ((False, 2), (True, 0)):
            c = Counter(real)
            lib.isrActBackward = c
            try:
                run(fuse, False)
            finally:
                lib.isrActBackward = real
            assert c.n == expect, (fuse, c.n)
    finally:
        ops.GATE_FUSION = old


def test_gate_fusion_steps_aside_when_the_intermediate_gradient_is_observed():
    """A tensor hook or retain_grad() on the ReLU output between two fused convolutions must see dL/dy itself -- the gradient of
    the unfused graph -- not dL/dy already multiplied by (y > 0): the fusion is skipped for that pair, every other gradient stays
    bit-identical."""
    from isosurfacesuperresolution_amd import ops
    g = torch.Generator().manual_seed(78)
    x0 = (torch.rand(1, 64, 32, 64, generator=g) - 0.5).cuda()
    ws = [((torch.rand(64, 64, 3, 3, generator=g) - 0.5) * 0.1).cuda() for _ in range(2)]
    tgt = torch.rand(1, 64, 32, 64, generator=g).cuda()

    def run(fuse, how):
        ops.GATE_FUSION = fuse
        x = x0.clone().requires_grad_(True)
        w = [t.clone().requires_grad_(True) for t in ws]
        y1 = ops.conv3x3(x, w[0], None, act='relu')
        seen = {}
        if how == "hook":
            y1.register_hook(lambda gr: seen.setdefault("g", gr.clone()))
        elif how == "retain":
            y1.retain_grad()
        y2 = ops.conv3x3(y1, w[1], None)
        ((y2 - tgt) ** 2).mean().backward()
        if how == "retain":
            seen["g"] = y1.grad.clone()
        return seen.get("g"), [x.grad] + [t.grad for t in w], y1.detach()

    old = ops.GATE_FUSION
    try:
        for how in ("hook", "retain"):
            gref, ref, y1 = run(False, how)
            gfus, got, _ = run(True, how)
            assert torch.equal(gref, gfus), how
            assert (gref[y1 <= 0] != 0).any(), "the true gradient is non-zero where the ReLU is closed: the comparison means something"
            for a, c in zip(ref, got):
                assert torch.equal(a, c), how
    finally:
        ops.GATE_FUSION = old


@pytest.mark.parametrize("shape", [(2, 6, 101, 5, 32, 32), (1, 6, 5, 5, 9, 7), (3, 6, 8, 6, 5, 12), (1, 4, 7, 2, 1, 1)])
def test_residual_reconstruction_kernel_matches_interpolate_add_cat(shape):
    """isrReconResidualForward / Backward against the reference's slice + F.interpolate(x4, bilinear) + add + cat
    (enhancenet.py:65-78), values and both gradients."""
    import torch.nn.functional as F
    from isosurfacesuperresolution_amd import ops
    n, cout, cin, k, h, w = shape
    g = torch.Generator().manual_seed(3)
    y0 = torch.rand(n, cout, 4 * h, 4 * w, generator=g).cuda()
    x0 = torch.rand(n, cin, h, w, generator=g).cuda()
    gout = torch.rand(n, cout, 4 * h, 4 * w, generator=g).cuda()

    def ref(y, x):
        r = F.interpolate(x[:, 0:k], size=[4 * h, 4 * w], mode='bilinear', align_corners=False)
        return r + y if k == cout else torch.cat([r + y[:, 0:k], y[:, k:]], dim=1)

    res = {}
    for name, fn in (("ref", ref), ("hip", lambda y, x: ops.recon_residual(y, x, k))):
        y, x = y0.clone().requires_grad_(True), x0.clone().requires_grad_(True)
        assert ops.recon_residual_supported(y, x, k)
        out = fn(y, x)
        out.backward(gout)
        res[name] = (out.detach(), y.grad, x.grad)
    for a, b, tol in zip(res["ref"], res["hip"], (1e-6, 0.0, 2e-5)):
        assert (a - b).abs().max().item() <= tol
    assert res["hip"][2][:, k:].abs().max().item() == 0.0 if k < cin else True


def test_batched_weight_preparation_equals_the_per_layer_one():
    """isrConvSplitPrepareMany (all layers, forward + data-gradient images, two launches) must write the very bytes that
    isrConvSplitPrepare writes layer by layer (the data-gradient image there from a flipped / transposed copy)."""
    from isosurfacesuperresolution_amd import ops
    g = torch.Generator().manual_seed(9)
    shapes = [(64, 64), (64, 101), (32, 40), (64, 64), (16, 9)]
    ws = [((torch.rand(co, ci, 3, 3, generator=g) - 0.5) * (10.0 ** (k - 2))).cuda() for k, (co, ci) in enumerate(shapes)]
    ref = [(ops._prepare_split(w, False).clone(), ops._prepare_split(w, True).clone()) for w in ws]
    ops._split_cache.clear()
    ops.prepare_split_many(ws)
    for w, (f, b) in zip(ws, ref):
        assert ops._split_cached(w, False) and ops._split_cached(w, True)
        assert torch.equal(ops._prepare_split(w, False), f) and torch.equal(ops._prepare_split(w, True), b)
    # a weight that changes in place is stale again
    ws[1].add_(0.25)
    assert not ops._split_cached(ws[1], False)
    ops.prepare_split_many(ws)
    assert ops._split_cached(ws[1], False) and not torch.equal(ops._prepare_split(ws[1], False), ref[1][0])
