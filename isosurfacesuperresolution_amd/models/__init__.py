"""Generator networks.  ``createNetwork`` mirrors ``SuperresolutionNetwork/models/__init__.py:21-49``;
only EnhanceNet is on the hot path, the alternative generators
(SubpixelNet, TecoGAN, RCAN) are out of scope and raise."""
from .enhancenet import EnhanceNet
from .videotools import VideoTools


def createNetwork(name, upscale_factor, input_channels, channel_mask, output_channels, additional_opt):
    print('upscale_factor:', upscale_factor)
    print('input_channels:', input_channels)
    print('channel_mask:', channel_mask)
    print('output_channels:', output_channels)
    key = name.lower()
    if key == 'enhancenet':
        return EnhanceNet(upscale_factor, input_channels, channel_mask, output_channels, additional_opt)
    if key in ('subpixelnet', 'tecogan', 'rcan'):
        raise NotImplementedError("generator '%s' is outside the accelerated hot path" % name)
    raise ValueError('Unknown model %s' % name)
