"""Shared skeleton of the two network plugins.

Both reference plugins (network/vnet.py:20-51, network/vbnet.py:21-53) are the same encoder/decoder: a 16-channel
stem, four stride-2 stages that double the width, four transposed stages that take a skip connection, and a class
head.  They differ only in which stages use bottleneck residual blocks.  The stage table below is the single source
of truth; attribute names match the reference so checkpoints keep their keys.
"""
import torch
import torch.nn as nn

from segmentation3d import _ops

from segmentation3d.network.module.vnet_inblock import InputBlock
from segmentation3d.network.module.vnet_outblock import OutputBlock
from segmentation3d.network.module.vnet_upblock import UpBlock
from segmentation3d.network.module.vnet_downblock import DownBlock

STEM_WIDTH = 16
# (attribute, input width, number of residual convs)
ENCODER_STAGES = (('down_32', 16, 1), ('down_64', 32, 2), ('down_128', 64, 3), ('down_256', 128, 3))
# (attribute, input width, output width incl. skip, number of residual convs, encoder feature used as skip)
DECODER_STAGES = (('up_256', 256, 256, 3, 'down_128'), ('up_128', 256, 128, 3, 'down_64'),
                  ('up_64', 128, 64, 2, 'down_32'), ('up_32', 64, 32, 1, 'in_block'))
MAX_STRIDE = 2 ** len(ENCODER_STAGES)


class VNetBase(nn.Module):
    """volumetric segmentation network; `bottleneck` lists the stage names that use BottResidualBlock3"""

    def __init__(self, in_channels, out_channels, bottleneck=()):
        super(VNetBase, self).__init__()
        self.in_block = InputBlock(in_channels, STEM_WIDTH)
        for name, width, convs in ENCODER_STAGES:
            setattr(self, name, DownBlock(width, convs, compression=name in bottleneck))
        for name, cin, cout, convs, _ in DECODER_STAGES:
            setattr(self, name, UpBlock(cin, cout, convs, compression=name in bottleneck))
        self.out_block = OutputBlock(DECODER_STAGES[-1][2], out_channels)

    def forward(self, input):
        if not isinstance(input, torch.Tensor):
            raise TypeError('input must be a torch.Tensor')
        if input.dim() != 5 or any(int(s) % MAX_STRIDE for s in input.shape[2:]):
            raise ValueError('input must be [N, C, D, H, W] with D, H, W divisible by {} (got {})'.format(
                MAX_STRIDE, tuple(input.shape)))
        feats = {'in_block': self.in_block(input)}
        x, source = feats['in_block'], 'in_block'
        # every encoder feature but the deepest has two consumers: the next DownBlock and a decoder skip.  The link lets the
        # DownBlock's data-gradient kernel add the skip gradient instead of autograd summing two full tensors.
        links = {}
        for name, _, _ in ENCODER_STAGES:
            links[source] = _ops.ResidualLink() if (x.requires_grad and torch.is_grad_enabled()) else None
            x = getattr(self, name)(x, skip_link=links[source])
            feats[name], source = x, name
        for name, _, _, _, skip in DECODER_STAGES:
            x = getattr(self, name)(x, feats[skip], skip_link=links.get(skip))
        return self.out_block(x)

    def max_stride(self):
        return MAX_STRIDE


def init_parameters(net, initializer):
    """apply a per-module initialiser (kaiming / gaussian) to every sub-module, as `net.apply` does"""
    net.apply(initializer)
    return net
