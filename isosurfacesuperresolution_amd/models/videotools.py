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
"""
import torch
import torch.nn.functional as F


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
    def warp_upscale(image_high, flow_low, upscale_factor, special_mask=False):
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
