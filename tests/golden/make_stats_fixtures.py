"""Generates tests/golden/stats_reference.npz by IMPORTING the reference's `utils` package (SSIM / MS-SSIM / PSNR,
SuperresolutionNetwork/utils/ssim.py, psnr.py) unmodified, in the build container.  Only scalars are committed: the inputs are
regenerated from seeds by the test (`stats_inputs` below is imported by tests/test_stats_cpu.py) and a checksum of them is stored so
that a different random stream cannot pass unnoticed.  The statistics script itself (mainPSNR3_AllStats.py) runs at import time and
needs cv2 / imageio / checkpoints on disk: it cannot be imported; its ingredients can.

Run:  python tests/golden/make_stats_fixtures.py
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference/SuperresolutionNetwork"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stats_reference.npz")


def stats_inputs():
    """Correlated (prediction, ground truth) pairs in the value ranges the statistics see: colour / depth / AO in [0, 1], normals in
    [-1, 1], at a size that keeps the 11-tap window through all five MS-SSIM levels (192 x 256) and at one that does not (72 x 100)."""
    g = torch.Generator().manual_seed(1234)
    out = {}
    for tag, (c, h, w) in (("color", (3, 192, 256)), ("depth", (1, 192, 256)), ("small", (3, 72, 100))):
        yy, xx = torch.meshgrid(torch.linspace(0, 1, h), torch.linspace(0, 1, w), indexing="ij")
        base = torch.stack([0.5 + 0.4 * torch.sin(6.0 * xx + k) * torch.cos(5.0 * yy - k) for k in range(c)]).unsqueeze(0)
        gt = (base + 0.05 * torch.rand(1, c, h, w, generator=g)).clamp(0, 1)
        pred = (gt + 0.08 * (torch.rand(1, c, h, w, generator=g) - 0.5)).clamp(0, 1)
        out[tag] = (pred, gt)
    pred, gt = out["color"]
    out["normal"] = (pred * 2 - 1, gt * 2 - 1)
    return out


def main():
    torch.set_num_threads(1)
    sys.path.insert(0, REF)
    import utils
    res = {}
    for tag, (pred, gt) in stats_inputs().items():
        res["sum_" + tag] = np.float64(pred.double().sum().item() + 2.0 * gt.double().sum().item())
        res["msssim_" + tag] = np.float64(utils.MSSSIM()(pred, gt).item())
        res["ssim_" + tag] = np.float64(utils.SSIM()(pred, gt).item())
        res["psnr_" + tag] = np.float64(utils.PSNR()(pred, gt).item())
        m = (gt[:, 0:1] > 0.5).float()
        res["psnr_masked_" + tag] = np.float64(utils.PSNR()(pred, gt, mask=m).item())
    np.savez_compressed(OUT, **res)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
    for k, v in res.items():
        print(k, v)


if __name__ == "__main__":
    main()
