"""Generates tests/golden/ref_checkpoint.pth + ref_checkpoint_io.npz by IMPORTING the reference's `models` package
from /root/reference/SuperresolutionNetwork, unmodified, in the build container: a checkpoint exactly as the
reference's training script writes it (`mainVideoUnshaded.py:799-811`: the whole pickled generator object under
'model', the option dictionary under 'parameters', plus optimizer and scheduler) and the reference network's own
output on a fixed input.  Only these data files are committed; the reference never travels.

The weights are rounded to 12 mantissa bits so that the pickle compresses to a small fixture (zip, ~1.5 MB).

Run:  python tests/golden/make_checkpoint_fixture.py
"""
import argparse
import os
import sys
import zipfile

import numpy as np
import torch

REF = "/root/reference/SuperresolutionNetwork"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    torch.set_num_threads(1)
    sys.path.insert(0, REF)
    import models
    opt = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10)
    torch.manual_seed(7)
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt)
    with torch.no_grad():
        for p in net.parameters():      # coarser mantissas -> compressible; the values are what they are
            p.copy_((p.view(torch.int32) & ~0x7FF).view(torch.float32))
    optimizer = torch.optim.Adam(net.parameters(), lr=1e-4)
    scheduler = torch.optim.lr_scheduler.StepLR(optimizer, 500, 0.5)
    opt_dict = {"model": "EnhanceNet", "upsample": "bilinear", "reconType": "residual", "useBN": False,
                "numResidualLayers": 10, "initialImage": "zero", "upscale_factor": 4, "aoInverted": False}
    state = {'epoch': 13, 'model': net, 'parameters': opt_dict, 'optimizer': optimizer, 'scheduler': scheduler}
    raw_path = os.path.join(HERE, "model_epoch_12.pth")
    torch.save(state, raw_path)
    with zipfile.ZipFile(os.path.join(HERE, "ref_checkpoint.zip"), "w", zipfile.ZIP_DEFLATED, compresslevel=9) as z:
        z.write(raw_path, "model_epoch_12.pth")
    os.remove(raw_path)
    torch.manual_seed(8)
    x = torch.rand(1, 101, 12, 10)
    net.eval()
    with torch.no_grad():
        y, raw = net(x)
    np.savez_compressed(os.path.join(HERE, "ref_checkpoint_io.npz"), x=x.numpy(), y=y.numpy(), raw=raw.numpy(),
                        param_count=np.int64(sum(p.numel() for p in net.parameters())))


if __name__ == "__main__":
    main()
