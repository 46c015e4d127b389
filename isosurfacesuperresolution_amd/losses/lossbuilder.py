"""``LossBuilder``: factory of the elementary loss modules used by ``LossNetUnshaded``.

Only the terms of the hot-path training recipe are built (``l1``, ``mse``; README.md:45-64).  The
reference's VGG perceptual/texture losses need a network download of VGG19 weights
(``losses/lossbuilder.py:10,173``) and the GAN losses are not part of the recipe -- both are out
of scope and raise.
"""
import torch.nn as nn


class LossBuilder:
    def __init__(self, device):
        self.device = device

    def mse(self):
        return nn.MSELoss()

    def l1_loss(self):
        return nn.L1Loss()

    def _unsupported(self, what):
        raise NotImplementedError("%s is outside the accelerated hot path (SURVEY.md section 2, row 12)" % what)

    def downsample_loss(self, *a, **k):
        self._unsupported("downsample loss")

    def gan_loss(self, *a, **k):
        self._unsupported("adversarial loss")

    def get_style_and_content_loss(self, *a, **k):
        self._unsupported("VGG perceptual/texture loss")
