"""``LossNetUnshaded``: training loss on the unshaded G-buffer (mask, normal, depth, AO).

Public surface of ``SuperresolutionNetwork/losses/lossnet_unshaded.py`` (constructor arguments,
``forward(gt, pred, input, prev_input_warped, prev_pred_warped) -> (loss, {key: float})``,
static ``pad``) for the l1 / mse(l2) / temp-l2 terms of the README recipe
``l1:mask:1,l1:ao:1,l1:normal:10,l1:depth:10,temp-l2:color:0.1``:

* spec mini-language ``name:target[:weight]`` (``:30,45-107``); ``('mse','color')`` with weight 0 is
  always evaluated because PSNR is derived from it (``:37-38``);
* a border of ``padding`` pixels of gt / pred / prev is zeroed first (``pad``, ``:171-185`` -- note the
  reference crops both axes with the *height*, so only square inputs are meaningful there; here
  each axis uses its own size, identical for squares);
* normals are re-normalised, the gt mask ``clamp(m/2+1/2, 0, 1)`` gates normal / ao / depth terms
  (``:236-256``); colours come from an internal shader with fov 30, light (0,0,1), white material,
  specular off, strengths from ``opt.lossAmbient / lossDiffuse / lossSpecular / lossAO`` (``:116-126``);
* ``temp-l2`` compares against the warped previous prediction (``:357-388``).

Parity note: this module is checked against hand-derived values only -- importing the reference's
``losses`` package needs torchvision, which this image lacks (parity unpinned for S7).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..utils import ScreenSpaceShading


class LossBuilder:
    """Factory of the elementary loss modules ``LossNetUnshaded`` is assembled from (``SuperresolutionNetwork/losses/lossbuilder.py``): the
    terms of the hot-path training recipe (``l1``, ``mse``; README.md:45-64).  The reference's VGG perceptual / texture losses need a download
    of VGG19 weights (``lossbuilder.py:10,173``) and the GAN losses are not part of the recipe: out of scope,
    they raise."""

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


_TARGETS = ('mask', 'normal', 'color', 'ao', 'depth', 'all')


class LossNetUnshaded(nn.Module):
    # forward() accepts the upsampled low-resolution input and its warped predecessor like the reference's, but none of
    # the supported terms reads them (only the adversarial ones would): train.clip_loss skips producing them
    uses_input = False

    def __init__(self, device, input_channels, output_channels, high_res, padding, opt):
        super().__init__()
        # the reference returns the per-term values as Python floats (one device synchronisation per term and frame);
        # lazy_values = True returns detached tensors instead, so that a training step can be enqueued without stalls
        self.lazy_values = False
        # GPU tensors: the whole forward is ONE launch of isrLossUnshadedForward (+ a one-workgroup reduction) and the
        # whole backward ONE launch of isrLossUnshadedBackward (csrc/sr_train.hip) instead of ~350 PyTorch launches per
        # frame; fused = False keeps the module path below (what the CPU runs, and what the GPU tests compare against)
        self.fused = True
        self._fused_cfg = (None, None)
        self.padding = padding
        self.upsample = opt.upsample
        assert input_channels == 5
        assert output_channels == 6
        self.loss_list = [s.split(':') for s in opt.losses.split(',')]
        builder = LossBuilder(device)
        loss_dict = {'mse': builder.mse()}
        self.weight_dict = {('mse', 'color'): 0.0}
        self.has_discriminator = False
        self.has_style_or_content_loss = False
        self.has_temporal_l2_loss = False
        for entry in self.loss_list:
            if len(entry) < 2:
                raise ValueError("illegal format for loss list: " + ':'.join(entry))
            name, target = entry[0], entry[1]
            weight = float(entry[2]) if len(entry) > 2 else 1.0
            if target not in _TARGETS:
                raise ValueError("Unknown target: " + target)
            if name in ('mse', 'l2', 'l2_loss'):
                self.weight_dict[('mse', target)] = weight
            elif name in ('l1', 'l1_loss'):
                loss_dict['l1'] = builder.l1_loss()
                self.weight_dict[('l1', target)] = weight
            elif name in ('tl2', 'temp-l2'):
                loss_dict['temp-l2'] = builder.mse()
                self.weight_dict[('temp-l2', target)] = weight
                self.has_temporal_l2_loss = True
            elif name in ('l2-ds', 'l1-ds', 'perceptual', 'texture', 'adv', 'gan', 'tgan', 'sgan'):
                raise NotImplementedError("loss '%s' is outside the accelerated hot path" % name)
            else:
                raise ValueError('unknown loss %s' % name)
        self.loss_dict = nn.ModuleDict(loss_dict)
        print('Loss weights:', self.weight_dict)

        sh = ScreenSpaceShading(device)
        sh.fov(30)
        sh.ambient_light_color(np.array([opt.lossAmbient] * 3))
        sh.diffuse_light_color(np.array([opt.lossDiffuse] * 3))
        sh.specular_light_color(np.array([getattr(opt, 'lossSpecular', 0.0)] * 3))
        sh.specular_exponent(16)
        sh.enable_specular = False
        sh.light_direction(np.array([0.0, 0.0, 1.0]))
        sh.material_color(np.array([1.0, 1.0, 1.0]))
        sh.ambient_occlusion(opt.lossAO)
        self.shading = sh
        print('LossNet: ambient occlusion strength:', opt.lossAO)

    def get_discr_parameters(self):
        return []

    @staticmethod
    def pad(img, border):
        """Overwrites a ``border``-pixel frame of ``img`` with zeros (size unchanged)."""
        if border == 0:
            return img
        b, c, h, w = img.shape
        inner = img[:, :, border:h - border, border:w - border]
        out = F.pad(inner, (border, border, border, border), 'constant', 0)
        assert out.shape[2] == h and out.shape[3] == w
        return out

    def _fusable(self, gt, pred, prev):
        return (self.fused and pred.is_cuda and gt.is_cuda and pred.dtype == torch.float32 and gt.dtype == torch.float32
                and pred.shape[3] % 4 == 0 and 2 * self.padding < min(pred.shape[2], pred.shape[3])
                and not self.shading.enable_specular
                and (prev is None or (prev.is_cuda and prev.dtype == torch.float32 and prev.shape == pred.shape)))

    def _forward_fused(self, gt, pred, prev):
        key = (tuple(self.shading.packed_parameters()), self.shading._ao, bool(self.shading.inverse_ao), self.padding,
               tuple(sorted(self.weight_dict.items())))
        if self._fused_cfg[0] != key:
            self._fused_cfg = (key, ops.loss_unshaded_config(self.weight_dict, self.padding, self.shading))
        vals = ops.loss_unshaded(gt, pred, prev if self.has_temporal_l2_loss else None, self._fused_cfg[1])
        host = None if self.lazy_values else vals.detach().tolist()
        values = {}
        for kind in ('mse', 'l1', 'temp-l2'):
            for target in ('mask', 'normal', 'ao', 'depth', 'color'):
                if (kind, target) in self.weight_dict:
                    t = ops.LOSS_KINDS.index(kind) * 5 + ops.LOSS_TARGETS.index(target)
                    values[(kind, target)] = vals[t].detach() if self.lazy_values else host[t]
        return vals[15], values

    def forward(self, gt, pred, input, prev_input_warped, prev_pred_warped):
        B, Cout, Hh, Wh = gt.shape
        assert Cout == 6
        assert gt.shape == pred.shape
        if self._fusable(gt, pred, prev_pred_warped):
            return self._forward_fused(gt, pred, prev_pred_warped)
        gt = LossNetUnshaded.pad(gt, self.padding)
        pred = LossNetUnshaded.pad(pred, self.padding)
        if prev_pred_warped is not None:
            prev_pred_warped = LossNetUnshaded.pad(prev_pred_warped, self.padding)

        norm = ScreenSpaceShading.normalize
        gt_mask, pred_mask = gt[:, 0:1], pred[:, 0:1]
        gate = torch.clamp(gt_mask * 0.5 + 0.5, 0, 1)
        fields = {
            'mask': (gt_mask, pred_mask),
            'normal': (norm(gt[:, 1:4], dim=1) * gate, norm(pred[:, 1:4], dim=1) * gate),
            'ao': (gt[:, 5:6] * gate, pred[:, 5:6] * gate),
            'depth': (gt[:, 4:5] * gate, pred[:, 4:5] * gate),
        }
        gt_color = self.shading(gt)
        pred_color = self.shading(pred)
        fields['color'] = (gt_color, pred_color)

        total = 0.0
        values = {}
        for name in ('mse', 'l1'):
            for target in ('mask', 'normal', 'ao', 'depth', 'color'):
                key = (name, target)
                if key in self.weight_dict:
                    a, b = fields[target]
                    loss = self.loss_dict[name](a, b)
                    values[key] = loss.detach() if self.lazy_values else loss.item()
                    total = total + self.weight_dict[key] * loss

        if self.has_temporal_l2_loss:
            crit = self.loss_dict['temp-l2']
            prev = prev_pred_warped
            tfields = {
                'mask': lambda: (pred_mask, prev[:, 0:1]),
                'normal': lambda: (fields['normal'][1], norm(prev[:, 1:4], dim=1) * gate),
                'ao': lambda: (fields['ao'][1], prev[:, 5:6] * gate),
                'depth': lambda: (fields['depth'][1], prev[:, 4:5] * gate),
                'color': lambda: (pred_color, self.shading(prev)),
            }
            for target in ('mask', 'normal', 'ao', 'depth', 'color'):
                key = ('temp-l2', target)
                if key in self.weight_dict:
                    a, b = tfields[target]()
                    loss = crit(a, b)
                    values[key] = loss.detach() if self.lazy_values else loss.item()
                    total = total + self.weight_dict[key] * loss
        return total, values
