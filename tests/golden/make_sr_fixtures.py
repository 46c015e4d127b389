"""Generates tests/golden/sr_reference.npz by IMPORTING the reference's Python modules
(`models`, `utils`) from /root/reference/SuperresolutionNetwork, unmodified, in the build
container.  Only the resulting vectors are committed; the reference never travels.

Semantics pinned while generating: torch 2.10 CPU, fp32, one thread,
`F.grid_sample` forced to `align_corners=True` (the torch-1.0.1 default the reference was written
against, SuperresolutionNetwork/Requirements.txt); `F.interpolate`/`nn.Upsample` at their defaults.

Not covered here: `losses` (its package import needs torchvision, absent from this image, and no
stand-in is written for it) and `inference` (needs cv2).

Run:  python tests/golden/make_sr_fixtures.py
"""
import argparse
import functools
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference/SuperresolutionNetwork"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sr_reference.npz")


def main():
    torch.set_num_threads(1)
    torch.use_deterministic_algorithms(True)
    F.grid_sample = functools.partial(F.grid_sample, align_corners=True)
    sys.path.insert(0, REF)
    import models
    import utils
    from models import VideoTools

    out = {}
    opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10)

    # --- S1/S2/S10: EnhanceNet init + forward
    torch.manual_seed(0)
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt).eval()
    out["net_param_abs_sum"] = np.float64(sum(p.abs().double().sum().item() for p in net.parameters()))
    out["net_param_count"] = np.int64(sum(p.numel() for p in net.parameters()))
    names = [n for n, _ in net.state_dict().items()]
    out["net_state_keys"] = np.array(names)
    out["net_param_sums"] = np.array([p.double().sum().item() for p in net.state_dict().values()])
    torch.manual_seed(1)
    x = torch.rand(1, 101, 8, 8)
    with torch.no_grad():
        y, raw = net(x)
        feat = net.preblock(x)
        b0 = feat + net.blocks[0](feat)
    out["net_x_seed"] = np.int64(1)
    out["net_y"] = y.numpy()
    out["net_raw"] = raw.numpy()
    out["net_pre_mean_abs"] = np.float64(feat.abs().double().mean().item())
    out["net_block0_mean_abs"] = np.float64(b0.abs().double().mean().item())
    torch.manual_seed(1)
    x16 = torch.rand(1, 101, 16, 16)
    with torch.no_grad():
        y16, r16 = net(x16)
    out["net_ka2"] = np.array([y16.mean().item(), y16.abs().mean().item(), r16.abs().mean().item()])

    # --- S3/S4: VideoTools
    torch.manual_seed(2)
    img = torch.rand(1, 6, 32, 32)
    img[:, 0] = img[:, 0] * 2 - 1
    flow = (torch.rand(1, 2, 8, 8) - 0.5) * 0.1
    out["vt_img"] = img.numpy()
    out["vt_flow"] = flow.numpy()
    out["vt_warp_special"] = VideoTools.warp_upscale(img, flow, 4, special_mask=True).numpy()
    out["vt_warp_plain"] = VideoTools.warp_upscale(img, flow, 4, special_mask=False).numpy()
    out["vt_warp_zero_flow"] = VideoTools.warp_upscale(img, torch.zeros_like(flow), 4, special_mask=True).numpy()
    out["vt_flatten"] = VideoTools.flatten_high(img, 4).numpy()
    flow_big = (torch.rand(1, 2, 8, 8) - 0.5) * 1.5          # pushes samples outside the image
    out["vt_flow_big"] = flow_big.numpy()
    out["vt_warp_big"] = VideoTools.warp_upscale(img, flow_big, 4, special_mask=True).numpy()

    # --- S6: ScreenSpaceShading
    sh = utils.ScreenSpaceShading('cpu')
    sh.fov(30)
    sh.ambient_light_color(np.array([0.1, 0.1, 0.1]))
    sh.diffuse_light_color(np.array([1.0, 1.0, 1.0]))
    sh.specular_light_color(np.array([0.2, 0.2, 0.2]))
    sh.specular_exponent(16)
    sh.light_direction(np.array([0.1, 0.1, 1.0]))
    sh.material_color(np.array([1.0, 0.3, 0.0]))
    sh.ambient_occlusion(1.0)
    sh.background(np.array([0.2, 0.4, 0.6]))
    g = img.clone()
    g[:, 1:4] = utils.ScreenSpaceShading.normalize(g[:, 1:4] * 2 - 1, dim=1)
    out["sh_in"] = g.numpy()
    out["sh_out"] = sh(g).numpy()
    sh.inverse_ao = True
    sh.ambient_occlusion(0.6)
    out["sh_out_invao"] = sh(g).numpy()
    sh.inverse_ao = False
    sh.enable_specular = False
    out["sh_out_nospec"] = sh(g[:, 0:5]).numpy()
    z = torch.zeros(1, 3, 2, 2)
    z[0, :, 0, 0] = torch.tensor([3.0, 0.0, 4.0])
    out["sh_normalize"] = utils.ScreenSpaceShading.normalize(z, dim=1).numpy()

    # --- S5: initialImage
    torch.manual_seed(3)
    low = torch.rand(2, 5, 4, 6)
    out["ii_low"] = low.numpy()
    out["ii_input"] = utils.initialImage(low, 6, 'input', False, 4).numpy()
    out["ii_unshaded"] = utils.initialImage(low, 6, 'unshaded', False, 4).contiguous().numpy()
    out["ii_unshaded_inv"] = utils.initialImage(low, 6, 'unshaded', True, 4).contiguous().numpy()
    out["ii_zero_shape"] = np.array(utils.initialImage(low, 6, 'zero', False, 4).shape)

    # --- PSNR (utils/psnr.py)
    torch.manual_seed(4)
    a, b = torch.rand(2, 3, 16, 16), torch.rand(2, 3, 16, 16)
    m = (torch.rand(2, 1, 16, 16) > 0.4).float()
    out["psnr_a"], out["psnr_b"], out["psnr_m"] = a.numpy(), b.numpy(), m.numpy()
    out["psnr_plain"] = utils.PSNR()(a, b).numpy()
    out["psnr_masked"] = utils.PSNR()(a, b, m).numpy()

    # --- SSIM (utils/ssim.py), "next" row 4
    try:
        out["ssim"] = np.float64(utils.SSIM()(a, b).item())
    except Exception as e:  # ordinary python error -> recorded, not worked around
        print("SSIM not generated:", repr(e))

    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
