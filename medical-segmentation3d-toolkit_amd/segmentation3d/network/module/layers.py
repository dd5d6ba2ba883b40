"""Parameter-holding layer modules with the reference's class names and state_dict keys.

The reference composes stock `nn.Conv3d`, `nn.ConvTranspose3d` and `nn.GroupNorm(1, C)`
(network/module/conv_gn_relu3.py:10-11, vnet_upblock.py:11).  These classes keep the same attribute names
(`weight`, `bias`), shapes and default initialisation, so `state_dict()` / `load_state_dict()` round-trip reference
checkpoints unchanged and `kaiming_weight_init`'s class-name matching (weight_init.py:6-10) keeps working, but their
forward runs the HIP kernels.  The fused blocks read the parameters directly and skip these forwards.
"""
import math

import torch
import torch.nn as nn

from segmentation3d import _ops


class Conv3d(nn.Module):
    """nn.Conv3d restricted to what the reference uses: k3 s1 p1, k2 s2 p0, k1 s1 p0; groups=1"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, groups=1, bias=True):
        super(Conv3d, self).__init__()
        kernel_size, stride, padding = int(kernel_size), int(stride), int(padding)
        if groups != 1:
            raise ValueError('only groups=1 is supported')
        if (kernel_size, stride, padding) == (3, 1, 1):
            self.kind = 'k3'
        elif (kernel_size, stride, padding) == (2, 2, 0):
            self.kind = 'k2s2'
        elif (kernel_size, stride, padding) == (1, 1, 0):
            self.kind = 'k1'
        else:
            raise ValueError('unsupported Conv3d geometry ksize={} stride={} padding={} (the reference networks use '
                             'k3/s1/p1, k2/s2/p0 and k1)'.format(kernel_size, stride, padding))
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = (kernel_size,) * 3, (stride,) * 3, (padding,) * 3
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        # same defaults as torch.nn.modules.conv._ConvNd.reset_parameters
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.weight.shape[1] * self.weight[0, 0].numel()
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, input):
        return _ops.conv(input, self.weight, self.bias, self.kind)

    def extra_repr(self):
        return '{}, {}, kernel_size={}, stride={}, padding={}'.format(self.in_channels, self.out_channels,
                                                                      self.kernel_size, self.stride, self.padding)


class ConvTranspose3d(nn.Module):
    """nn.ConvTranspose3d with kernel_size=2, stride=2 (weight [Cin, Cout, 2, 2, 2])"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, groups=1, bias=True):
        super(ConvTranspose3d, self).__init__()
        if int(kernel_size) != 2 or int(stride) != 2 or groups != 1:
            raise ValueError('only ConvTranspose3d(kernel_size=2, stride=2, groups=1) is supported')
        self.kind = 'convT'
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = (2, 2, 2), (2, 2, 2)
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels, 2, 2, 2))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.weight.shape[1] * self.weight[0, 0].numel()  # torch uses size(1) * receptive field
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, input):
        return _ops.conv(input, self.weight, self.bias, self.kind)

    def extra_repr(self):
        return '{}, {}, kernel_size=2, stride=2'.format(self.in_channels, self.out_channels)


class GroupNorm(nn.Module):
    """nn.GroupNorm(1, C): per-sample normalisation over C*D*H*W, affine, eps 1e-5"""

    def __init__(self, num_groups, num_channels, eps=1e-5, affine=True):
        super(GroupNorm, self).__init__()
        if num_groups != 1:
            raise ValueError('only GroupNorm(1, C) is supported (all reference call sites use one group)')
        if not affine:
            raise ValueError('affine=False is not supported')
        self.num_groups, self.num_channels, self.eps, self.affine = 1, num_channels, eps, True
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))

    def forward(self, input):
        return _ops.group_norm(input, self.weight, self.bias, relu=False, eps=self.eps)

    def extra_repr(self):
        return '1, {}, eps={}'.format(self.num_channels, self.eps)


class ReLU(nn.Module):
    """placeholder for nn.ReLU(inplace=True): the activation is fused into the GroupNorm kernel of the owning block;
    calling it on its own applies torch's relu (device plumbing, not on the hot path)."""

    def __init__(self, inplace=False):
        super(ReLU, self).__init__()
        self.inplace = inplace

    def forward(self, input):
        return torch.relu(input)


class Softmax(nn.Module):
    """nn.Softmax(dim=1) over channels; output is contiguous NCDHW"""

    def __init__(self, dim=1):
        super(Softmax, self).__init__()
        if dim != 1:
            raise ValueError('only dim=1 is supported')
        self.dim = 1

    def forward(self, input):
        return _ops.softmax_channels(input)


# ---------------------------------------------------------------------------------------------------------------------
# fused conv + GroupNorm (+ ReLU) units
# ---------------------------------------------------------------------------------------------------------------------
_GEOMETRY = {'k3': dict(kernel_size=3, stride=1, padding=1), 'k2s2': dict(kernel_size=2, stride=2, padding=0),
             'k1': dict(kernel_size=1, stride=1, padding=0)}


def attach_unit(owner, names, kind, cin, cout, act=True, bias=True):
    """register the parameter holders of one conv -> GroupNorm(1, C) [-> ReLU] unit on `owner` under the attribute
    names the reference uses (`names` = (conv, gn, act); act name None or act=False: no activation module), in that
    order, so `state_dict()` keys and their order equal the reference's.  kind: 'k3' | 'k2s2' | 'k1' | 'convT'."""
    conv_name, gn_name, act_name = names
    if kind == 'convT':
        conv = ConvTranspose3d(cin, cout, kernel_size=2, stride=2, groups=1, bias=bias)
    else:
        conv = Conv3d(cin, cout, groups=1, bias=bias, **_GEOMETRY[kind])
    setattr(owner, conv_name, conv)
    setattr(owner, gn_name, GroupNorm(1, cout))
    if act and act_name:
        setattr(owner, act_name, ReLU(inplace=True))


def run_unit(owner, names, input, relu, **fusion):
    """forward of a unit registered with attach_unit as ONE fused HIP op: conv (GroupNorm statistics in its epilogue)
    -> normalise + affine [+ residual] [+ ReLU].  `fusion`: residual / link_in / link_out of _ops.conv_gn_act."""
    conv, gn = getattr(owner, names[0]), getattr(owner, names[1])
    return _ops.conv_gn_act(input, conv.weight, conv.bias, gn.weight, gn.bias, kind=conv.kind, relu=relu, eps=gn.eps,
                            **fusion)
