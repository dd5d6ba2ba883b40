"""DownBlock -- mirrors network/module/vnet_downblock.py:5-22: conv k2 s2 (C -> 2C) -> GN -> ReLU -> residual block"""
import torch.nn as nn

from segmentation3d import _ops
from segmentation3d.network.module.layers import Conv3d, GroupNorm, ReLU
from segmentation3d.network.module.residual_block3 import ResidualBlock3, BottResidualBlock3


class DownBlock(nn.Module):
    """ downsample block of v-net """

    def __init__(self, in_channels, num_convs, compression=False, ratio=4):
        super(DownBlock, self).__init__()
        out_channels = in_channels * 2
        self.down_conv = Conv3d(in_channels, out_channels, kernel_size=2, stride=2, groups=1)
        self.down_gn = GroupNorm(1, num_channels=out_channels)
        self.down_act = ReLU(inplace=True)
        if compression:
            self.rblock = BottResidualBlock3(out_channels, 3, 1, 1, ratio, num_convs)
        else:
            self.rblock = ResidualBlock3(out_channels, 3, 1, 1, num_convs)

    def forward(self, input):
        out = _ops.conv_gn_act(input, self.down_conv.weight, self.down_conv.bias, self.down_gn.weight, self.down_gn.bias,
                               kind='k2s2', relu=True, eps=self.down_gn.eps)
        return self.rblock(out)
