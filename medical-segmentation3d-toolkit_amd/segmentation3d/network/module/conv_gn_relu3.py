"""ConvGnRelu3 / BottConvGnRelu3 -- mirrors network/module/conv_gn_relu3.py:4-34 of the reference.

Same constructor arguments, sub-module names (`conv`, `gn`, `act`; `conv1..3`) and forward semantics
`act?(GN(conv(x)))`; the three ops run as one fused HIP unit (conv -> GroupNorm statistics in the conv epilogue ->
normalise + ReLU [+ residual add]) instead of three ATen kernels.
"""
import torch.nn as nn

from segmentation3d import _ops
from segmentation3d.network.module.layers import Conv3d, GroupNorm, ReLU


class ConvGnRelu3(nn.Module):
    """ classic combination: conv + group normalization [+ relu], post-activation mode """

    def __init__(self, in_channels, out_channels, ksize, stride, padding, do_act=True, bias=True):
        super(ConvGnRelu3, self).__init__()
        self.conv = Conv3d(in_channels, out_channels, ksize, stride=stride, padding=padding, groups=1, bias=bias)
        self.gn = GroupNorm(1, out_channels)
        self.do_act = do_act
        if do_act:
            self.act = ReLU(inplace=True)

    def forward(self, input, residual=None, force_act=False, link_in=None, link_out=None):
        """`residual`/`force_act` let a residual block fuse `ReLU(input + GN(conv(.)))` into this unit
        (residual_block3.py:24); `link_in`/`link_out` (see _ops.ResidualLink) let the block's first and last unit fuse
        the backward sum of the identity-path and conv-path gradients.  Without them this is the reference forward."""
        return _ops.conv_gn_act(input, self.conv.weight, self.conv.bias, self.gn.weight, self.gn.bias,
                                residual=residual, kind=self.conv.kind, relu=self.do_act or force_act, eps=self.gn.eps,
                                link_in=link_in, link_out=link_out)


class BottConvGnRelu3(nn.Module):
    """Bottle neck structure: C -> C/ratio -> C/ratio -> C, all three with the same ksize (conv_gn_relu3.py:28-30)"""

    def __init__(self, in_channels, out_channels, ksize, stride, padding, ratio, do_act=True, bias=True):
        super(BottConvGnRelu3, self).__init__()
        self.conv1 = ConvGnRelu3(in_channels, in_channels // ratio, ksize, stride, padding, do_act=True, bias=bias)
        self.conv2 = ConvGnRelu3(in_channels // ratio, in_channels // ratio, ksize, stride, padding, do_act=True, bias=bias)
        self.conv3 = ConvGnRelu3(in_channels // ratio, out_channels, ksize, stride, padding, do_act=do_act, bias=bias)

    def forward(self, input, residual=None, force_act=False, link_in=None, link_out=None):
        out = self.conv2(self.conv1(input, link_in=link_in))
        return self.conv3(out, residual=residual, force_act=force_act, link_out=link_out)
