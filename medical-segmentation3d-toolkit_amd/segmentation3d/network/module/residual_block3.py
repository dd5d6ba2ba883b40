"""ResidualBlock3 / BottResidualBlock3 -- mirrors network/module/residual_block3.py:5-46 of the reference.

`ReLU(input + ops(input))` where the last op has no activation; the add + ReLU is fused into the last op's
GroupNorm-apply kernel, so the block costs no extra pass over the tensor.
"""
import torch.nn as nn

from segmentation3d import _ops

from segmentation3d.network.module.conv_gn_relu3 import ConvGnRelu3, BottConvGnRelu3
from segmentation3d.network.module.layers import ReLU


def _residual_forward(ops, input, out_slot=None):
    """act(input + ops(input)) with the add + ReLU fused into the last unit (residual_block3.py:21-26, 44-46) and, for
    backward, the identity-path gradient routed from the last unit to the first unit's data-gradient kernel."""
    n = len(ops)
    if n == 1:
        return ops[0](input, residual=input, force_act=True, out_slot=out_slot)
    link = _ops.ResidualLink() if input.requires_grad else None
    output = ops[0](input, link_in=link)
    for i in range(1, n - 1):
        output = ops[i](output)
    return ops[n - 1](output, residual=input, force_act=True, link_out=link, out_slot=out_slot)


class _ResidualBase(nn.Module):
    """`ops` (nn.Sequential of units, the last one without activation) + `act`, the reference's attribute names"""

    def _build(self, make_unit, num_convs):
        self.ops = nn.Sequential(*[make_unit(i != num_convs - 1) for i in range(num_convs)])
        self.act = ReLU(inplace=True)

    def forward(self, input, out_slot=None):
        return _residual_forward(self.ops, input, out_slot)


class ResidualBlock3(_ResidualBase):
    """num_convs plain 3x3x3 units (residual_block3.py:5-26)"""

    def __init__(self, channels, ksize, stride, padding, num_convs):
        super(ResidualBlock3, self).__init__()
        self._build(lambda act: ConvGnRelu3(channels, channels, ksize, stride, padding, do_act=act), num_convs)


class BottResidualBlock3(_ResidualBase):
    """num_convs bottleneck units (residual_block3.py:29-46)"""

    def __init__(self, channels, ksize, stride, padding, ratio, num_convs):
        super(BottResidualBlock3, self).__init__()
        self._build(lambda act: BottConvGnRelu3(channels, channels, ksize, stride, padding, ratio, do_act=act), num_convs)


def make_residual_block(channels, num_convs, compression=False, ratio=4):
    """the residual stage of a Down / Up block: 3x3x3, stride 1, padding 1 (vnet_downblock.py:14-17, vnet_upblock.py:14-17)"""
    if compression:
        return BottResidualBlock3(channels, 3, 1, 1, ratio, num_convs)
    return ResidualBlock3(channels, 3, 1, 1, num_convs)
