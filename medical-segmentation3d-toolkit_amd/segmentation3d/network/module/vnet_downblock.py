"""DownBlock of the V-Net encoder (reference: network/module/vnet_downblock.py:5-22).

Halves the resolution and doubles the width with a 2x2x2 stride-2 convolution + GroupNorm + ReLU (one fused HIP op,
csrc/conv_k2_mfma.hip + gn.hip), then refines with a residual block of `num_convs` 3x3x3 units (bottleneck variant
when `compression`).  Attribute names (`down_conv`, `down_gn`, `down_act`, `rblock`) are the reference's, so its
checkpoints load unchanged.
"""
import torch.nn as nn

from segmentation3d.network.module.layers import attach_unit, run_unit
from segmentation3d.network.module.residual_block3 import make_residual_block

_DOWN = ('down_conv', 'down_gn', 'down_act')


class DownBlock(nn.Module):

    def __init__(self, in_channels, num_convs, compression=False, ratio=4):
        super(DownBlock, self).__init__()
        width = 2 * in_channels
        attach_unit(self, _DOWN, 'k2s2', in_channels, width)
        self.rblock = make_residual_block(width, num_convs, compression, ratio)

    def forward(self, input, skip_link=None, out_slot=None):
        """skip_link: _ops.ResidualLink shared with the UpBlock that concatenates `input` as its skip tensor; in backward
        that block parks the skip gradient there and the stride-2 conv's data-gradient kernel adds it.
        out_slot (inference): the block's output goes straight into its half of the decoder's concatenated skip buffer"""
        return self.rblock(run_unit(self, _DOWN, input, relu=True, link_in=skip_link), out_slot=out_slot)
