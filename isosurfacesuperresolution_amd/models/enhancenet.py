"""EnhanceNet generator (Sajjadi et al.) as used by the reference's unshaded video pipeline.

Same constructor, attributes, sub-module names and ``state_dict`` keys as
``SuperresolutionNetwork/models/enhancenet.py`` (``preblock.0``, ``blocks.{0..9}.{0,2}``,
``postblock.{1,4,6,8}``), so reference checkpoints load; the forward pass is written against
``ops.conv3x3`` so that on an MI355X every 3x3 convolution runs the fused HIP/MFMA kernel
(bias + ReLU + residual add + optional x2 bilinear upsample of the input in one launch) instead
of conv -> activation -> add as separate library calls.

Architecture (enhancenet.py:92-125): pre: conv(Cin->64)+ReLU; 10 x [conv+ReLU, conv] with
skip; post: up x2, conv+ReLU, up x2, conv+ReLU, conv+ReLU, conv(64->Cout).  Reconstruction
(``:51-90``): with ``reconType='residual'`` the first ``len(channel_mask)`` output channels get
the bilinearly resized first ``len(channel_mask)`` input channels added (sliced, not gathered).
Initialisation (``:127-133``): orthogonal with gain sqrt(2) for the convs inside ``blocks`` only.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


class EnhanceNet(nn.Module):
    def __init__(self, upscale_factor, input_channels, channel_mask, output_channels, opt):
        super().__init__()
        assert upscale_factor == 4
        self.upscale_factor = upscale_factor
        self.upsample = opt.upsample
        self.recon_type = opt.reconType
        self.use_bn = opt.useBN
        self.channel_mask = channel_mask
        self.input_channels = input_channels
        self.output_channels = output_channels
        self._build(input_channels, output_channels)
        self._initialize_weights()

    def _upsample(self, factor):
        if self.upsample in ('nearest', 'bilinear', 'bicubic'):
            return nn.Upsample(scale_factor=factor, mode=self.upsample)
        # the reference's pixel-shuffle branch references a missing attribute (enhancenet.py:48)
        raise ValueError("unsupported upsample mode '%s'" % self.upsample)

    def _build(self, cin, cout):
        def conv(i, o):
            return nn.Conv2d(i, o, 3, padding=1)

        self.preblock = nn.Sequential(conv(cin, 64), nn.ReLU())
        blocks = []
        for _ in range(10):   # hard-coded as in the reference (enhancenet.py:98)
            if self.use_bn:
                blocks.append(nn.Sequential(conv(64, 64), nn.BatchNorm2d(64), nn.ReLU(),
                                            conv(64, 64), nn.BatchNorm2d(64)))
            else:
                blocks.append(nn.Sequential(conv(64, 64), nn.ReLU(), conv(64, 64)))
        self.blocks = nn.ModuleList(blocks)
        self.postblock = nn.Sequential(
            self._upsample(2), conv(64, 64), nn.ReLU(),
            self._upsample(2), conv(64, 64), nn.ReLU(),
            conv(64, 64), nn.ReLU(),
            conv(64, cout))

    def _initialize_weights(self):
        gain = nn.init.calculate_gain('relu')
        for block in self.blocks:
            for m in block.modules():
                if isinstance(m, nn.Conv2d):
                    nn.init.orthogonal_(m.weight, gain)

    # ------------------------------------------------------------------ forward
    def _recon_image(self, inputs, outputs):
        k = len(self.channel_mask)
        if self.recon_type != 'residual':
            return outputs, outputs
        if k > self.output_channels:
            raise ValueError("number of output channels must be at least the number of masked input channels")
        if self.upsample == 'bilinear' and ops.recon_residual_supported(outputs, inputs, k):
            return ops.recon_residual(outputs, inputs, k), outputs     # slice + resize + add + cat as one HIP launch
        resized = F.interpolate(inputs[:, 0:k], size=[outputs.shape[2], outputs.shape[3]], mode=self.upsample,
                                **({'align_corners': False} if self.upsample in ('bilinear', 'bicubic') else {}))
        if k == self.output_channels:
            return resized + outputs, outputs
        return torch.cat([resized + outputs[:, 0:k], outputs[:, k:]], dim=1), outputs

    def _fused_ok(self):
        return (not self.use_bn) and self.upsample == 'bilinear'

    def trunk_convs(self):
        """[(weight, bias)] of the low-resolution trunk: preblock, then conv1 / conv2 of every block (what ops.trunk_supported takes)."""
        pre = self.preblock[0]
        return [(pre.weight, pre.bias)] + [(m.weight, m.bias) for block in self.blocks for m in (block[0], block[2])]

    def forward_features(self, inputs, last_layer=True, last_two=True, last_three=True, after_trunk=None, packed_tail=False):
        """The convolutional trunk only: the tensor ``_recon_image`` receives (used by the fused
        frame pipeline, which folds the reconstruction into its finishing kernel).  ``last_layer=False`` stops
        before the final 64 -> 6 convolution (``self.postblock[8]``), which the pipeline fuses with the finishing;
        ``last_two=False`` stops before ``self.postblock[6]`` as well (the fused 1080p tail, ``ops.tail_conv_finish``),
        ``last_three=False`` before ``self.postblock[4]`` (whose output the pipeline hands to the tail packed-split).
        ``after_trunk``: called once the low-resolution trunk has been enqueued (the frame pipeline starts the next frame's
        ray-march there: beside the multi-round 1080p kernels instead of beside the one-round dataflow trunk).
        ``packed_tail`` (with ``last_two=False`` or ``last_three=False``): the caller takes an ``ops.PackedSplit`` -- the upsampling
        layers may then run phase-decomposed on the trunk's packed-split result (``ops.conv3x3_ups_phase``)."""
        assert self._fused_ok()
        c = ops.conv3x3
        pre = self.preblock[0]
        convs = self.trunk_convs()
        if ops.trunk_supported(inputs, convs):
            f = ops.trunk_dataflow(inputs, convs)       # preblock + all blocks in ONE dataflow launch (csrc/sr_conv_trunk.hip)
        elif getattr(inputs, '_isr_prepacked', None) is not None:
            raise RuntimeError("forward_features: this input exists only packed-split inside the dataflow trunk's workspace "
                               "(ops.assemble_input_packed) and the dataflow trunk does not take it; assemble it with ops.assemble_input")
        else:
            f = c(inputs, pre.weight, pre.bias, act='relu')
            for block in self.blocks:
                f = ops.residual_block(f, block[0].weight, block[0].bias, block[2].weight, block[2].bias)
        if after_trunk is not None:
            after_trunk()
        p = self.postblock
        fp = getattr(f, '_isr_packed', None)
        if packed_tail and fp is not None and ops.ups_phase_supported(fp, p[1].weight) and tuple(p[4].weight.shape) == (64, 64, 3, 3):
            # the dataflow trunk left its result packed-split: both upsampling layers run phase-decomposed (no interpolation at run
            # time, operands by LDS-DMA; csrc/sr_conv_upsp.h), packed-split from the trunk to the fused tail
            g = ops.conv3x3_ups_phase(fp, p[1].weight, p[1].bias, act='relu')
            if not last_three:
                return g
            if ops.ups_phase_supported(g, p[4].weight):
                return ops.conv3x3_ups_phase(g, p[4].weight, p[4].bias, act='relu')
            raise RuntimeError("forward_features(packed_tail=True): the second upsampling layer does not take the packed-split tensor")
        f = c(f, p[1].weight, p[1].bias, act='relu', upsample2x=True)
        if not last_three:
            return f
        f = c(f, p[4].weight, p[4].bias, act='relu', upsample2x=True)
        if not last_two:
            return f
        f = c(f, p[6].weight, p[6].bias, act='relu')
        return c(f, p[8].weight, p[8].bias) if last_layer else f

    def forward(self, inputs):
        if not self._fused_ok():
            features = self.preblock(inputs)
            for block in self.blocks:
                features = features + block(features)
            outputs = self.postblock(features)
            return self._recon_image(inputs, outputs)
        return self._recon_image(inputs, self.forward_features(inputs))
