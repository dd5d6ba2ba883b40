"""InputBlock -- mirrors network/module/vnet_inblock.py:4-15: conv k3 p1 -> GroupNorm(1, C) -> ReLU"""
import torch.nn as nn

from segmentation3d import _ops
from segmentation3d.network.module.layers import Conv3d, GroupNorm, ReLU


class InputBlock(nn.Module):
    """ input block of vb-net """

    def __init__(self, in_channels, out_channels):
        super(InputBlock, self).__init__()
        self.conv = Conv3d(in_channels, out_channels, kernel_size=3, padding=1)
        self.gn = GroupNorm(1, num_channels=out_channels)
        self.act = ReLU(inplace=True)

    def forward(self, input):
        return _ops.conv_gn_act(input, self.conv.weight, self.conv.bias, self.gn.weight, self.gn.bias, kind='k3',
                                relu=True, eps=self.gn.eps)
