"""PSNR as the reference measures it (``SuperresolutionNetwork/utils/psnr.py:10-22``)."""
import torch


class PSNR(torch.nn.Module):
    def forward(self, img1, img2, mask=None, epsilon=1e-7):
        """img: BxCxHxW in [0,1]; optional mask Bx1xHxW in [0,1] (masked pixels are ignored and the
        result is rescaled by H*W / sum(mask))."""
        if mask is None:
            return 10 * torch.log10(1 / (epsilon + torch.mean((img1 - img2) ** 2, dim=[1, 2, 3])))
        B, C, H, W = mask.shape
        factor = (H * W) / torch.sum(mask, dim=[1, 2, 3])
        mse = torch.mean((mask * img1 - mask * img2) ** 2, dim=[1, 2, 3])
        return 10 * factor * torch.log10(1 / (epsilon + mse))
