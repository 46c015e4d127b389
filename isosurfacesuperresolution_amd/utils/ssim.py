"""SSIM / MS-SSIM as the reference's statistics scripts use them
(``SuperresolutionNetwork/utils/ssim.py:10-135``; callers ``mainPSNR3_AllStats.py:216-228``):
gaussian window (sigma 1.5) of size min(11, H, W), *valid* convolution (no padding), dynamic range
guessed from the data (max > 128 -> 255 else 1; min < -0.5 -> -1 else 0), C1 = (0.01 L)^2, C2 = (0.03 L)^2.
"""
import math

import torch
import torch.nn.functional as F

_MS_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def create_window(window_size, channel=1):
    g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(window_size)])
    g = (g / g.sum()).unsqueeze(1)
    w2d = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)
    return w2d.expand(channel, 1, window_size, window_size).contiguous()


def _dynamic_range(img, val_range):
    if val_range is not None:
        return val_range
    max_val = 255 if torch.max(img) > 128 else 1
    min_val = -1 if torch.min(img) < -0.5 else 0
    return max_val - min_val


def ssim(img1, img2, window_size=11, window=None, size_average=True, full=False, val_range=None):
    L = _dynamic_range(img1, val_range)
    _, channel, height, width = img1.size()
    if window is None:
        window = create_window(min(window_size, height, width), channel=channel).to(device=img1.device, dtype=img1.dtype)
    conv = lambda t: F.conv2d(t, window, padding=0, groups=channel)
    mu1, mu2 = conv(img1), conv(img2)
    mu1_sq, mu2_sq, mu12 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1 = conv(img1 * img1) - mu1_sq
    s2 = conv(img2 * img2) - mu2_sq
    s12 = conv(img1 * img2) - mu12
    C1, C2 = (0.01 * L) ** 2, (0.03 * L) ** 2
    v1 = 2.0 * s12 + C2
    v2 = s1 + s2 + C2
    cs = torch.mean(v1 / v2)
    ssim_map = ((2 * mu12 + C1) * v1) / ((mu1_sq + mu2_sq + C1) * v2)
    ret = ssim_map.mean() if size_average else ssim_map.mean(1).mean(1).mean(1)
    return (ret, cs) if full else ret


def msssim(img1, img2, window_size=11, size_average=True, val_range=None, normalize=False):
    weights = torch.tensor(_MS_WEIGHTS, dtype=img1.dtype, device=img1.device)
    sims, css = [], []
    for _ in range(len(_MS_WEIGHTS)):
        sim, cs = ssim(img1, img2, window_size=window_size, size_average=size_average, full=True, val_range=val_range)
        sims.append(sim)
        css.append(cs)
        img1 = F.avg_pool2d(img1, (2, 2))
        img2 = F.avg_pool2d(img2, (2, 2))
    sims, css = torch.stack(sims), torch.stack(css)
    if normalize:
        sims, css = (sims + 1) / 2, (css + 1) / 2
    # as in the reference (which follows the Matlab implementation): prod(cs_i^w_i, i<last) * ssim_last^w
    return torch.prod((css ** weights)[:-1] * (sims ** weights)[-1])


class SSIM(torch.nn.Module):
    def __init__(self, window_size=11, size_average=True, val_range=None):
        super().__init__()
        self.window_size, self.size_average, self.val_range = window_size, size_average, val_range
        self.channel = 1
        self.window = create_window(window_size)

    def forward(self, img1, img2):
        channel = img1.size(1)
        if channel != self.channel or self.window.dtype != img1.dtype or self.window.device != img1.device:
            self.window = create_window(self.window_size, channel).to(img1.device).type(img1.dtype)
            self.channel = channel
        return ssim(img1, img2, window=self.window, window_size=self.window_size, size_average=self.size_average)


class MSSSIM(torch.nn.Module):
    def __init__(self, window_size=11, size_average=True, channel=3):
        super().__init__()
        self.window_size, self.size_average, self.channel = window_size, size_average, channel

    def forward(self, img1, img2):
        return msssim(img1, img2, window_size=self.window_size, size_average=self.size_average)
