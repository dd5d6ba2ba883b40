"""ResidualBlock3 / BottResidualBlock3 -- mirrors network/module/residual_block3.py:5-46 of the reference.

`ReLU(input + ops(input))` where the last op has no activation; the add + ReLU is fused into the last op's
GroupNorm-apply kernel, so the block costs no extra pass over the tensor.
"""
import torch.nn as nn

from segmentation3d.network.module.conv_gn_relu3 import ConvGnRelu3, BottConvGnRelu3
from segmentation3d.network.module.layers import ReLU


class ResidualBlock3(nn.Module):
    """ residual block with variable number of convolutions """

    def __init__(self, channels, ksize, stride, padding, num_convs):
        super(ResidualBlock3, self).__init__()
        layers = []
        for i in range(num_convs):
            layers.append(ConvGnRelu3(channels, channels, ksize, stride, padding, do_act=(i != num_convs - 1)))
        self.ops = nn.Sequential(*layers)
        self.act = ReLU(inplace=True)

    def forward(self, input):
        output = input
        n = len(self.ops)
        for i, op in enumerate(self.ops):
            if i != n - 1:
                output = op(output)
            else:
                output = op(output, residual=input, force_act=True)  # act(input + ops(input))
        return output


class BottResidualBlock3(nn.Module):
    """ block with bottle neck conv"""

    def __init__(self, channels, ksize, stride, padding, ratio, num_convs):
        super(BottResidualBlock3, self).__init__()
        layers = []
        for i in range(num_convs):
            layers.append(BottConvGnRelu3(channels, channels, ksize, stride, padding, ratio, do_act=(i != num_convs - 1)))
        self.ops = nn.Sequential(*layers)
        self.act = ReLU(inplace=True)

    def forward(self, input):
        output = input
        n = len(self.ops)
        for i, op in enumerate(self.ops):
            if i != n - 1:
                output = op(output)
            else:
                output = op(output, residual=input, force_act=True)
        return output
