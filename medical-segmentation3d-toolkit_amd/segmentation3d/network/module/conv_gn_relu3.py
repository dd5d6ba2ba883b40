"""ConvGnRelu3 / BottConvGnRelu3: the 3x3x3 building unit of the V-Net and its VB-Net bottleneck form
(reference: network/module/conv_gn_relu3.py:4-34).

`ConvGnRelu3`      act?(GroupNorm_1(conv(x)))                                   sub-modules `conv`, `gn` (, `act`)
`BottConvGnRelu3`  C -> C/ratio -> C/ratio -> Cout, every stage with the same kernel size   `conv1`, `conv2`, `conv3`
Constructor arguments and state_dict keys are the reference's.  A unit is ONE fused HIP op (conv with the GroupNorm
statistics in its epilogue, then normalise + affine [+ residual] [+ ReLU]) instead of three ATen kernels; the extra
forward arguments let a residual block fuse its add + ReLU (and, in backward, the identity-path gradient) into the
block's last / first unit.
"""
import torch.nn as nn

from segmentation3d.network.module.layers import attach_unit, run_unit

_UNIT = ('conv', 'gn', 'act')


def _kind(ksize, stride, padding):
    for kind, geo in (('k3', (3, 1, 1)), ('k2s2', (2, 2, 0)), ('k1', (1, 1, 0))):
        if (int(ksize), int(stride), int(padding)) == geo:
            return kind
    raise ValueError('unsupported Conv3d geometry ksize={} stride={} padding={} (the reference networks use k3/s1/p1, '
                     'k2/s2/p0 and k1)'.format(ksize, stride, padding))


class ConvGnRelu3(nn.Module):

    def __init__(self, in_channels, out_channels, ksize, stride, padding, do_act=True, bias=True):
        super(ConvGnRelu3, self).__init__()
        self.do_act = do_act
        attach_unit(self, _UNIT, _kind(ksize, stride, padding), in_channels, out_channels, act=do_act, bias=bias)

    def forward(self, input, residual=None, force_act=False, link_in=None, link_out=None, out_slot=None):
        """plain call = the reference forward.  residual / force_act: compute ReLU(residual + GN(conv(input))) in this
        unit (residual_block3.py:24); link_in / link_out: _ops.ResidualLink side channel for the backward pass;
        out_slot (inference): slice of another buffer the output is written into (_ops.conv_gn_act)."""
        return run_unit(self, _UNIT, input, relu=self.do_act or force_act, residual=residual, link_in=link_in,
                        link_out=link_out, out_slot=out_slot)


class BottConvGnRelu3(nn.Module):

    def __init__(self, in_channels, out_channels, ksize, stride, padding, ratio, do_act=True, bias=True):
        super(BottConvGnRelu3, self).__init__()
        narrow = in_channels // ratio
        stages = ((in_channels, narrow, True), (narrow, narrow, True), (narrow, out_channels, do_act))
        for k, (cin, cout, act) in enumerate(stages, start=1):
            setattr(self, 'conv{}'.format(k), ConvGnRelu3(cin, cout, ksize, stride, padding, do_act=act, bias=bias))

    def forward(self, input, residual=None, force_act=False, link_in=None, link_out=None, out_slot=None):
        squeezed = self.conv2(self.conv1(input, link_in=link_in))
        return self.conv3(squeezed, residual=residual, force_act=force_act, link_out=link_out, out_slot=out_slot)
