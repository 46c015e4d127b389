"""Hole filling of the low-resolution flow field on the device.

The reference inpaints flow channels 8-9 where mask==0 on the CPU with OpenCV's Navier-Stokes
``cv.inpaint(..., 3, cv.INPAINT_NS)`` (``inference/loadedmodel.py:77-82``), a GPU->CPU->GPU round
trip inside every frame.  OpenCV is a third-party dependency that is not part of the reference
tree (opencv-python 4.0.1, Requirements.txt) and is absent here, so bit parity with it is
unpinned; this module replaces it with a push-pull (mask-weighted pyramid) fill that never leaves
the GPU: known pixels are kept exactly, holes receive the mask-weighted average of the nearest
coarser level that has data.  
"""
import torch
import torch.nn.functional as F


def fill_flow(flow, valid):
    """flow [B,2,h,w]; valid [B,1,h,w] (1 where the renderer produced a hit). Returns filled flow."""
    valid = valid.to(flow.dtype)
    levels = []
    v, m = flow * valid, valid
    # pull: mask-weighted 2x2 averages until a level has no holes left (or is 1x1)
    while True:
        levels.append((v, m))
        if v.shape[-1] <= 1 and v.shape[-2] <= 1:
            break
        ph, pw = v.shape[-2] % 2, v.shape[-1] % 2
        vp = F.pad(v, (0, pw, 0, ph))
        mp = F.pad(m, (0, pw, 0, ph))
        ms = F.avg_pool2d(mp, 2)
        vs = F.avg_pool2d(vp, 2)
        v = torch.where(ms > 0, vs / ms.clamp_min(1e-12), torch.zeros_like(vs))
        m = (ms > 0).to(flow.dtype)
        if len(levels) > 16:
            break
    # push: fill holes of each finer level from the (already complete) coarser one
    filled = levels[-1][0]
    for v, m in reversed(levels[:-1]):
        up = F.interpolate(filled, size=v.shape[-2:], mode='bilinear', align_corners=False)
        filled = torch.where(m > 0, v, up)
    return filled
