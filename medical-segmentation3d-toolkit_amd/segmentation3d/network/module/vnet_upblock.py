"""UpBlock -- mirrors network/module/vnet_upblock.py:6-23: convT k2 s2 -> GN -> ReLU -> cat((up, skip), 1) -> residual block"""
import torch.nn as nn

from segmentation3d import _ops
from segmentation3d.network.module.layers import ConvTranspose3d, GroupNorm, ReLU
from segmentation3d.network.module.residual_block3 import ResidualBlock3, BottResidualBlock3


class UpBlock(nn.Module):
    """ Upsample block of v-net """

    def __init__(self, in_channels, out_channels, num_convs, compression=False, ratio=4):
        super(UpBlock, self).__init__()
        self.up_conv = ConvTranspose3d(in_channels, out_channels // 2, kernel_size=2, stride=2, groups=1)
        self.up_gn = GroupNorm(1, out_channels // 2)
        self.up_act = ReLU(inplace=True)
        if compression:
            self.rblock = BottResidualBlock3(out_channels, 3, 1, 1, ratio, num_convs)
        else:
            self.rblock = ResidualBlock3(out_channels, 3, 1, 1, num_convs)

    def forward(self, input, skip):
        half = self.up_conv.weight.shape[1]
        if half % 4 == 0 and skip.shape[1] % 4 == 0:
            # up first, then skip (vnet_upblock.py:21), the up-branch normalised straight into the concatenated buffer
            out = _ops.up_cat(input, self.up_conv.weight, self.up_conv.bias, self.up_gn.weight, self.up_gn.bias, skip,
                              relu=True, eps=self.up_gn.eps)
        else:
            out = _ops.conv_gn_act(input, self.up_conv.weight, self.up_conv.bias, self.up_gn.weight, self.up_gn.bias,
                                   kind='convT', relu=True, eps=self.up_gn.eps)
            out = _ops.cat_channels(out, skip)
        return self.rblock(out)
