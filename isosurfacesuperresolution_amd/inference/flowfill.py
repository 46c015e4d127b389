"""Hole filling of the low-resolution flow field on the device.

The reference inpaints flow channels 8-9 where mask==0 on the CPU with OpenCV's Navier-Stokes
``cv.inpaint(..., 3, cv.INPAINT_NS)`` (``inference/loadedmodel.py:77-82``), a GPU->CPU->GPU round
trip inside every frame.  OpenCV is a third-party dependency that is not part of the reference
tree (opencv-python 4.0.1, Requirements.txt) and is absent here, so bit parity with it is
unpinned; this module replaces it with a push-pull (mask-weighted pyramid) fill that never leaves
the GPU: known pixels are kept exactly, holes receive the mask-weighted average of the nearest
coarser level that has data.  

This file is the DEFINITION of that fill, spelled out in elementwise operations with one IEEE rounding
each (sums of the 2 x 2 block in the order (0,0), (0,1), (1,0), (1,1); the bilinear push as
``hy (hx a + lx b) + ly (hx c + lx d)``, seven roundings): the HIP kernels (``csrc/sr_frame.hip``:
``flow_fill_kernel``, ``flow_fill_one_kernel``) compute the same bits
(``tests/test_flowfill_gpu.py::test_fill_is_bit_identical_to_the_module_definition``).  That matters
more than its size suggests: the filled flow positions the warp of the previous frame, whose fp32
conditioning at silhouette edges is ~1e-4 per 6e-8 of grid coordinate (``models/videotools.py``).
"""
import torch

from ..models.videotools import bilinear_source_index, bilinear_taps


def fill_flow(flow, valid):
    """flow [B,2,h,w]; valid [B,1,h,w] (1 where the renderer produced a hit). Returns filled flow."""
    dtype, device = flow.dtype, flow.device
    valid = valid.to(dtype)
    levels = []
    v, m = flow * valid, valid
    # pull: mask-weighted 2x2 averages (zero-padded to even sizes) until the level is 1x1
    while True:
        levels.append((v, m))
        h, w = v.shape[-2], v.shape[-1]
        if (h <= 1 and w <= 1) or len(levels) > 18:
            break
        ph, pw = h % 2, w % 2
        if ph or pw:
            v = torch.nn.functional.pad(v, (0, pw, 0, ph))
            m = torch.nn.functional.pad(m, (0, pw, 0, ph))

        def block_sum(t):
            return ((t[..., 0::2, 0::2] + t[..., 0::2, 1::2]) + t[..., 1::2, 0::2]) + t[..., 1::2, 1::2]
        ms, vs = block_sum(m) * 0.25, block_sum(v) * 0.25
        v = torch.where(ms > 0, vs / ms.clamp_min(1e-12), torch.zeros((), dtype=dtype, device=device))
        m = (ms > 0).to(dtype)
    # push: fill holes of each finer level from the (already complete) coarser one; bilinear, align_corners=False, with the
    # resize scale = coarse size / fine size divided IN ``dtype``
    filled = levels[-1][0]
    for v, m in reversed(levels[:-1]):
        fh, fw, ch, cw = v.shape[-2], v.shape[-1], filled.shape[-2], filled.shape[-1]
        one = torch.ones((), dtype=dtype, device=device)
        y0, y1, ly = bilinear_source_index(fh, (one * ch) / (one * fh), ch, dtype, device)
        x0, x1, lx = bilinear_source_index(fw, (one * cw) / (one * fw), cw, dtype, device)
        up = bilinear_taps(filled, y0, y1, ly, x0, x1, lx)
        filled = torch.where(m > 0, v, up)
    return filled
