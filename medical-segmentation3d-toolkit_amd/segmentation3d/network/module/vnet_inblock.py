"""InputBlock: the V-Net stem (reference: network/module/vnet_inblock.py:4-15).

One 3x3x3 convolution from the image modalities to `out_channels` features, GroupNorm(1, C) and ReLU, executed as a
single fused op (thin-input MFMA kernel, csrc/conv_thin.hip).  Sub-modules keep the reference names `conv`, `gn`, `act`.
"""
import torch.nn as nn

from segmentation3d.network.module.layers import attach_unit, run_unit

_STEM = ('conv', 'gn', 'act')


class InputBlock(nn.Module):

    def __init__(self, in_channels, out_channels):
        super(InputBlock, self).__init__()
        attach_unit(self, _STEM, 'k3', in_channels, out_channels)

    def forward(self, input, out_slot=None):
        return run_unit(self, _STEM, input, relu=True, out_slot=out_slot)
