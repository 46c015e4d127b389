"""``VideoTools``: space-to-depth of the previous high-res frame and flow-driven warping.

Public surface of ``SuperresolutionNetwork/models/videotools.py`` (``flatten_high`` ``:8-25``,
``warp_upscale`` ``:51-87``), written for explicit semantics instead of library defaults:

* ``flatten_high``: output channel ``c*r*r + dy*r + dx`` <- input ``(c, r*y+dy, r*x+dx)``
  (the inverse of ``nn.PixelShuffle``).
* ``warp_upscale``: flow ``(x, y)`` is scaled by ``(-2, +2)`` (``:65-68``), bilinearly upsampled
  with ``align_corners=False`` (``:70``), added to a ``linspace(-1, 1)`` pixel grid (``:37-38``) and
  fed to a bilinear, zero-padded sampler with **align_corners=True** -- the torch 1.0.1 default the
  reference was written for (``Requirements.txt``).  With ``special_mask`` the
  first channel is mapped [-1,1]->[0,1] before and back after, so that padding means "mask = -1".

Since round 5 the warp is spelled out in elementwise operations (one IEEE rounding each) instead of
``F.interpolate`` + ``F.grid_sample``: the reference's formulation goes through NORMALISED coordinates,
``sx = (grid + 1) (W - 1) / 2`` with ``grid`` in [-1, 1], so an fp32 rounding of ``grid`` (6e-8) is
6e-5 pixels at W = 1920, and across a silhouette edge of the previous frame (mask -1 -> +1) that is
1e-4 in the warped value -- any two fp32 evaluations that do not round identically differ by that much
(library kernels do not promise an operation order: torch's CPU ``linspace`` depends on the SIMD width,
its ``grid_sample`` on FMA contraction), and the temporal recurrence multiplies the difference by the
network's gain every frame.  With every operation, its order and its rounding fixed HERE, the HIP kernels
(``csrc/sr_frame.hip: assemble_input_kernel``, ``csrc/sr_train.hip``) compute the SAME bits on the same
inputs (``tests/test_conv_gpu.py::test_assembled_input_is_bit_identical_to_the_module_path``).  This default is a deliberate
RE-ROUNDING of the reference's warp: against the reference-generated fixture (``tests/test_sr_golden_cpu.py``) it is within 2e-5
(1e-5 with zero flow) -- the conditioning described above -- and no further from an fp64 evaluation than the reference's own fp32
output is; ``warp_upscale_library`` keeps the library-call form and is the one that meets 1e-6 against the fixture.  Upscale factors
whose reciprocal is not exact in binary (3, 5, 6, ...) take the library form: the explicit form's source indices assume ``1 / r`` exact.
"""
import torch
import torch.nn.functional as F


def bilinear_source_index(n_out, scale, n_in, dtype, device):
    """Source taps of a bilinear resize with ``align_corners=False`` (ATen ``area_pixel_compute_source_index``):
    ``s = max(0, (dst + 0.5) * scale - 0.5)``; -> (i0, i1 = min(i0 + 1, n_in - 1), weight of i1).  ``scale``: a python
    float that is exact in ``dtype`` (1 / upscale factor) or a 0-dim tensor of ``dtype`` (in / out sizes divided IN that dtype)."""
    d = torch.arange(n_out, dtype=dtype, device=device)
    s = ((d + 0.5) * scale - 0.5).clamp_min(0)
    f0 = s.floor()
    i0 = f0.long()
    i1 = i0 + (i0 < n_in - 1).long()
    return i0, i1, s - f0


def bilinear_taps(t, y0, y1, ly, x0, x1, lx):
    """``hy (hx a + lx b) + ly (hx c + lx d)`` on the last two dimensions of ``t`` -- seven roundings, in this order."""
    hy, hx = (1 - ly).unsqueeze(-1), (1 - lx).unsqueeze(0)
    ly, lx = ly.unsqueeze(-1), lx.unsqueeze(0)
    r0, r1 = t.index_select(-2, y0), t.index_select(-2, y1)
    a, b = r0.index_select(-1, x0), r0.index_select(-1, x1)
    c, d = r1.index_select(-1, x0), r1.index_select(-1, x1)
    return hy * (hx * a + lx * b) + ly * (hx * c + lx * d)


def pixel_grid(n, dtype, device):
    """``linspace(-1, 1, n)`` as the correctly rounded value of ``2 i / (n - 1) - 1``: evaluated in double (multiply, divide,
    subtract -- one rounding each), then rounded once to ``dtype``.  (``torch.linspace`` in fp32 is within one ulp of it but its
    CPU kernel's result depends on the vector width it was compiled for.)"""
    i = torch.arange(n, dtype=torch.float64, device=device)
    return ((i * 2.0) / float(max(n - 1, 1)) - 1.0).to(dtype)


class VideoTools:

    @staticmethod
    def flatten_high(image_high, upscale_factor):
        b, c, h, w = image_high.shape
        r = upscale_factor
        oh, ow = h // r, w // r
        tiles = image_high.contiguous().view(b, c, oh, r, ow, r)
        return tiles.permute(0, 1, 3, 5, 2, 4).reshape(b, c * r * r, oh, ow)

    # keyed by size, dtype AND device (the reference's cache ignores the last two, videotools.py:33-34)
    _offset_cache = dict()

    @staticmethod
    def _grid_offsets(H, W, dtype, device):
        key = (H, W, dtype, str(device))
        grid = VideoTools._offset_cache.get(key)
        if grid is None:
            ys = torch.linspace(-1, +1, H, dtype=dtype, device=device).view(H, 1).expand(H, W)
            xs = torch.linspace(-1, +1, W, dtype=dtype, device=device).view(1, W).expand(H, W)
            grid = torch.stack((xs, ys), dim=2).unsqueeze(0).detach()
            VideoTools._offset_cache[key] = grid
        return grid

    @staticmethod
    def _warp_plan(h, w, r, dtype, device):
        """Everything of the warp that depends on the sizes only (cached): resize taps of the flow, pixel grid."""
        key = ('plan', h, w, r, dtype, str(device))
        plan = VideoTools._offset_cache.get(key)
        if plan is None:
            H, W = r * h, r * w
            plan = (bilinear_source_index(H, 1.0 / r, h, dtype, device), bilinear_source_index(W, 1.0 / r, w, dtype, device),
                    pixel_grid(H, dtype, device).view(H, 1), pixel_grid(W, dtype, device).view(1, W))
            VideoTools._offset_cache[key] = plan
        return plan

    @staticmethod
    def warp_upscale(image_high, flow_low, upscale_factor, special_mask=False):
        B, C, h, w = flow_low.shape
        assert C == 2
        r = int(upscale_factor)
        H, W = r * h, r * w
        if tuple(image_high.shape[-2:]) != (H, W):
            raise ValueError("warp_upscale: previous frame %s is not %d x the flow's %s" % (tuple(image_high.shape[-2:]), r, (H // r, W // r)))
        if (1.0 / r) * r != 1.0:             # (the reference accepts any factor: the library form does too)
            return VideoTools.warp_upscale_library(image_high, flow_low, upscale_factor, special_mask)
        dtype, device = flow_low.dtype, flow_low.device
        (y0, y1, ly), (x0, x1, lx), lin_y, lin_x = VideoTools._warp_plan(h, w, r, dtype, device)
        # flow (x, y) scaled by (-2, +2) (exact), resized bilinearly, added to the pixel grid
        gx = lin_x + bilinear_taps(flow_low[:, 0] * -2.0, y0, y1, ly, x0, x1, lx)
        gy = lin_y + bilinear_taps(flow_low[:, 1] * 2.0, y0, y1, ly, x0, x1, lx)
        # sampler, align_corners=True: pixel position = (grid + 1) (size - 1) / 2; zero padding outside the image
        sx, sy = (gx + 1.0) * (0.5 * (W - 1)), (gy + 1.0) * (0.5 * (H - 1))
        fx0, fy0 = sx.floor(), sy.floor()
        wx1, wy1 = sx - fx0, sy - fy0
        wx0, wy0 = 1.0 - wx1, 1.0 - wy1
        # (clamped before the integer conversion: positions far outside the image are invalid either way)
        ix0, iy0 = fx0.clamp(-2, W).long(), fy0.clamp(-2, H).long()
        if special_mask:
            image_high = torch.cat([image_high[:, 0:1] * 0.5 + 0.5, image_high[:, 1:]], dim=1)
        flat = image_high.reshape(B, image_high.shape[1], H * W)
        out = None
        for dyy, dxx, wgt in ((0, 0, wx0 * wy0), (0, 1, wx1 * wy0), (1, 0, wx0 * wy1), (1, 1, wx1 * wy1)):
            ix, iy = ix0 + dxx, iy0 + dyy
            ok = ((ix >= 0) & (ix < W) & (iy >= 0) & (iy < H)).unsqueeze(1)
            idx = (iy.clamp(0, H - 1) * W + ix.clamp(0, W - 1)).view(B, 1, H * W).expand(B, flat.shape[1], H * W)
            v = torch.where(ok, flat.gather(2, idx).view(B, -1, H, W), torch.zeros((), dtype=dtype, device=device))
            term = v * wgt.unsqueeze(1)
            out = term if out is None else out + term                 # ((v00 w00 + v01 w01) + v10 w10) + v11 w11
        if special_mask:
            out = torch.cat([out[:, 0:1] * 2.0 - 1.0, out[:, 1:]], dim=1)
        return out

    @staticmethod
    def warp_upscale_library(image_high, flow_low, upscale_factor, special_mask=False):
        """The same warp through ``F.interpolate`` + ``F.grid_sample`` (the reference's calls, videotools.py:51-87): equal to
        ``warp_upscale`` up to the roundings those kernels choose."""
        B, C, H, W = flow_low.shape
        assert C == 2
        key = ('scale', flow_low.dtype, str(flow_low.device))
        scale = VideoTools._offset_cache.get(key)        # cached: building it is a host->device copy per call
        if scale is None:
            scale = torch.tensor([-2.0, 2.0], dtype=flow_low.dtype, device=flow_low.device).view(1, 2, 1, 1)
            VideoTools._offset_cache[key] = scale
        flow_high = F.interpolate(flow_low * scale, scale_factor=upscale_factor, mode='bilinear',
                                  align_corners=False)
        flow_high = flow_high.permute(0, 2, 3, 1)
        _, Hh, Wh, _ = flow_high.shape
        grid = VideoTools._grid_offsets(Hh, Wh, flow_high.dtype, flow_high.device) + flow_high
        if special_mask:
            image_high = torch.cat([image_high[:, 0:1] * 0.5 + 0.5, image_high[:, 1:]], dim=1)
        warped = F.grid_sample(image_high, grid, mode='bilinear', padding_mode='zeros', align_corners=True)
        if special_mask:
            warped = torch.cat([warped[:, 0:1] * 2 - 1, warped[:, 1:]], dim=1)
        return warped
