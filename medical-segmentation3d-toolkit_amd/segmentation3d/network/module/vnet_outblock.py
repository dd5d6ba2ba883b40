"""OutputBlock -- mirrors network/module/vnet_outblock.py:4-24:
conv k3 (C -> classes) -> GN -> ReLU -> conv k1 (classes -> classes) -> GN -> Softmax(dim=1)"""
import torch.nn as nn

from segmentation3d import _ops
from segmentation3d.network.module.layers import Conv3d, GroupNorm, ReLU, Softmax


class OutputBlock(nn.Module):
    """ output block of v-net: per-voxel class probabilities, contiguous [N, classes, D, H, W] """

    def __init__(self, in_channels, out_channels):
        super(OutputBlock, self).__init__()
        self.conv1 = Conv3d(in_channels, out_channels, kernel_size=3, padding=1)
        self.gn1 = GroupNorm(1, out_channels)
        self.act1 = ReLU(inplace=True)
        self.conv2 = Conv3d(out_channels, out_channels, kernel_size=1)
        self.gn2 = GroupNorm(1, out_channels)
        self.softmax = Softmax(dim=1)

    def forward(self, input):
        out = _ops.conv_gn_act(input, self.conv1.weight, self.conv1.bias, self.gn1.weight, self.gn1.bias, kind='k3',
                               relu=True, eps=self.gn1.eps)
        out = _ops.conv_gn_act(out, self.conv2.weight, self.conv2.bias, self.gn2.weight, self.gn2.bias, kind='k1',
                               relu=False, eps=self.gn2.eps)
        return self.softmax(out)
