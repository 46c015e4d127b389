"""The upsampling layer's kernel forms against each other and against torch (SuperresolutionNetwork/models/enhancenet.py:113-124).

Diagnostics build (``diag_lib``: the forms beside the default exist there only, csrc/sr_diag.h): the three-workgroups-per-CU kernel
(csrc/sr_conv_ups3.h, the default), the role-split one (sr_conv_ups4.h), the software-pipelined one (sr_conv_ups5.h), the four-rows-per-wave
one (sr_conv_ups4r.h) and the one-stream persistent one (sr_conv_upsw.h) against the tile kernel (conv3x3_split_kernel<true>) -- the
same interpolation and products in the same order: EQUAL bit for bit.

Product build: the default kernel against an fp64 reference, and EQUAL to what the diagnostics build's default computes."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu



def _forms(fn):
    from isosurfacesuperresolution_amd import ops
    lib = ops._sr()
    lib.isrDebugSetSplitUpsForm.argtypes = [ctypes.c_int]
    lib.isrDebugSplitUpsForm.restype = ctypes.c_int
    default = lib.isrDebugSplitUpsForm()
    try:
        lib.isrDebugSetSplitUpsForm(0)
        tile = fn()
        lib.isrDebugSetSplitUpsForm(3)
        three = fn()
        lib.isrDebugSetSplitUpsForm(4)
        four = fn()
        lib.isrDebugSetSplitUpsForm(5)                       # the software-pipelined form (sr_conv_ups5.h): must equal the others too
        five = fn()
        lib.isrDebugSetSplitUpsForm(7)                       # the four-rows-per-wave form (sr_conv_ups4r.h): must equal the others too
        seven = fn()
        lib.isrDebugSetSplitUpsForm(8)                       # the one-stream persistent form (sr_conv_upsw.h): must equal the others too
        eight = fn()
    finally:
        lib.isrDebugSetSplitUpsForm(default)
    torch.cuda.synchronize()
    assert torch.equal(five, three), (five - three).abs().max().item()
    assert torch.equal(seven, three), (seven - three).abs().max().item()
    assert torch.equal(eight, three), (eight - three).abs().max().item()
    return tile, three, four


@pytest.mark.parametrize("h,w,cin", [(4, 16, 64), (5, 18, 64), (17, 34, 64), (135, 240, 64), (270, 480, 64), (540, 960, 64), (30, 50, 32), (9, 10, 16)])
def test_three_per_cu_and_role_split_upsampling_kernels_are_bit_identical_to_the_tile_kernel(h, w, cin, diag_lib):
    from isosurfacesuperresolution_amd import ops
    g = torch.Generator().manual_seed(h * 1000 + w)
    x = ((torch.rand(1, cin, h, w, generator=g) - 0.4) * 3).cuda()
    wt = ((torch.rand(64, cin, 3, 3, generator=g) - 0.5) * 0.2).cuda()
    b = ((torch.rand(64, generator=g) - 0.5) * 0.3).cuda()
    with torch.no_grad():
        tile, three, four = _forms(lambda: ops.conv3x3_split(x, wt, b, act='relu', upsample2x=True))
        assert torch.equal(tile, three), (tile - three).abs().max().item()
        assert torch.equal(tile, four), (tile - four).abs().max().item()
        ref = F.relu(F.conv2d(F.interpolate(x.double(), scale_factor=2, mode='bilinear', align_corners=False), wt.double(), b.double(), padding=1))
        err = (three.double() - ref).abs().max().item()
        assert err <= 2e-6 * max(1.0, ref.abs().max().item()), err
        if ops.packed_supported(x, wt, True):
            def packed_pixels():                                 # (the planes' padding behind the last pixel is never written: not part of the result)
                ps = ops.conv3x3_split_packed(x, wt, b, act='relu', upsample2x=True)
                return ps.data.view(2 * (ps.channels // 8), ps.plane, 4)[:, :ps.h * ps.w].clone()
            pt, p3, p4 = _forms(packed_pixels)
            assert torch.equal(pt, p3)
            assert torch.equal(pt, p4)


def test_three_per_cu_upsampling_kernel_takes_batches_and_residuals(diag_lib):
    from isosurfacesuperresolution_amd import ops
    g = torch.Generator().manual_seed(11)
    x = (torch.rand(3, 64, 20, 36, generator=g) - 0.5).cuda()
    wt = ((torch.rand(64, 64, 3, 3, generator=g) - 0.5) * 0.2).cuda()
    with torch.no_grad():
        tile, three, four = _forms(lambda: ops.conv3x3_split(x, wt, None, act='none', upsample2x=True))
    assert torch.equal(tile, three) and torch.equal(tile, four)


@pytest.mark.parametrize("h,w", [(5, 16), (135, 240), (270, 480)])
def test_product_build_upsampling_layer_against_fp64_and_the_diagnostics_build(h, w):
    """The library a deployment ships (no diagnostics fixture: lib/libisr_sr.so) runs the default form; the diagnostics build of the same
    sources gives the same bits (the diagnostic fields are compile-time constants in one and run-time zeros in the other)."""
    from isosurfacesuperresolution_amd import ops
    assert not ops.is_diagnostics_library() and ops.debug_switches() == 0
    g = torch.Generator().manual_seed(h * 7 + w)
    x = ((torch.rand(1, 64, h, w, generator=g) - 0.4) * 3).cuda()
    wt = ((torch.rand(64, 64, 3, 3, generator=g) - 0.5) * 0.2).cuda()
    b = ((torch.rand(64, generator=g) - 0.5) * 0.3).cuda()
    with torch.no_grad():
        y = ops.conv3x3_split(x, wt, b, act='relu', upsample2x=True)
        ps = ops.conv3x3_split_packed(x, wt, b, act='relu', upsample2x=True)
        ref = F.relu(F.conv2d(F.interpolate(x.double(), scale_factor=2, mode='bilinear', align_corners=False), wt.double(), b.double(), padding=1))
        err = (y.double() - ref).abs().max().item()
        assert err <= 2e-6 * max(1.0, ref.abs().max().item()), err
        assert (ps.to_float().double() - ref).abs().max().item() <= 2e-6 * max(1.0, ref.abs().max().item())
        with ops.diagnostics_library():
            assert ops.is_diagnostics_library()
            yd = ops.conv3x3_split(x, wt, b, act='relu', upsample2x=True)
            pd = ops.conv3x3_split_packed(x, wt, b, act='relu', upsample2x=True)
        assert torch.equal(y, yd)
        assert torch.equal(ps.to_float(), pd.to_float())
