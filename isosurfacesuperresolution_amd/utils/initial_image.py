"""``initialImage``: the "previous frame" fed to the network at the head of a sequence
(``SuperresolutionNetwork/utils/initial_image.py:5-54``)."""
import torch
import torch.nn.functional as F


def initialImage(current_input, channels, mode, aoInverted, upscaling=4):
    """current_input [B,Cin,h,w] -> [B,channels,h*upscaling,w*upscaling].
    mode "zero": zeros; "unshaded": mask=-1, normal=(0,0,1), depth=.5, ao=1 (0 if aoInverted);
    "input": bilinear upsampling (align_corners=False) of the input, missing channels = 1."""
    B, Cin, H, W = current_input.shape
    Hh, Wh = H * upscaling, W * upscaling
    kw = dict(dtype=current_input.dtype, device=current_input.device)
    if mode == "zero":
        return torch.zeros(B, channels, Hh, Wh, **kw)
    if mode == "unshaded":
        if channels == 5:
            defaults = [-1.0, 0.0, 0.0, 1.0, 0.5]
        elif channels == 6:
            defaults = [-1.0, 0.0, 0.0, 1.0, 0.5, 0.0 if aoInverted else 1.0]
        else:
            raise ValueError("for mode='unshaded', channels is expected to be 5 or 6")
        return torch.tensor(defaults, **kw).view(1, channels, 1, 1).expand(B, channels, Hh, Wh)
    if mode == "input":
        high = F.interpolate(current_input, scale_factor=upscaling, mode='bilinear', align_corners=False)
        if channels <= Cin:
            return high if channels == Cin else high[:, 0:channels]
        return torch.cat([high, torch.ones(B, channels - Cin, Hh, Wh, **kw)], dim=1)
    raise ValueError("unknown input mode: " + mode)
