"""UpBlock of the V-Net decoder (reference: network/module/vnet_upblock.py:6-23).

A 2x2x2 stride-2 transposed convolution brings the coarse features to the skip connection's resolution with half of
`out_channels`, GroupNorm + ReLU follow, the result is concatenated with the skip tensor (up-branch first) and refined
by a residual block.  Here the transposed conv, its GroupNorm / ReLU AND the concatenation are one autograd function
(_ops.up_cat): the normalised up-branch is written straight into its channel slice of the concatenated buffer.
Attribute names (`up_conv`, `up_gn`, `up_act`, `rblock`) are the reference's.
"""
import torch.nn as nn

from segmentation3d import _ops
from segmentation3d.network.module.layers import attach_unit, run_unit
from segmentation3d.network.module.residual_block3 import make_residual_block

_UP = ('up_conv', 'up_gn', 'up_act')


class UpBlock(nn.Module):

    def __init__(self, in_channels, out_channels, num_convs, compression=False, ratio=4):
        super(UpBlock, self).__init__()
        attach_unit(self, _UP, 'convT', in_channels, out_channels // 2)
        self.rblock = make_residual_block(out_channels, num_convs, compression, ratio)

    def forward(self, input, skip, skip_link=None, cat_buf=None):
        """cat_buf (inference): the concatenated buffer whose second half already holds `skip` (VNetBase.forward)"""
        conv, gn = self.up_conv, self.up_gn
        if conv.out_channels % 4 == 0 and skip.shape[1] % 4 == 0:
            merged = _ops.up_cat(input, conv.weight, conv.bias, gn.weight, gn.bias, skip, relu=True, eps=gn.eps,
                                 link_out=skip_link, cat_buf=cat_buf)
        else:   # odd channel counts: separate concatenation kernel
            merged = _ops.cat_channels(run_unit(self, _UP, input, relu=True), skip)
        return self.rblock(merged)
