"""Per-module weight initialisers with the reference's names (network/module/weight_init.py:4-29).

Convolution weights (every class whose name contains 'Conv3d' / 'ConvTranspose3d') get Kaiming-normal
(fan_in, gain sqrt 2) or N(0, conv_std) values and a zero bias; GroupNorm is deliberately left at gamma = 1,
beta = 0 because the reference only matches 'BatchNorm' / 'Linear' besides the convolutions.  Host-side torch RNG.
"""
import torch.nn as nn

from segmentation3d import _ops

_CONV_TAGS = ('Conv3d', 'ConvTranspose3d')


def _is_conv(module):
    name = type(module).__name__
    return any(tag in name for tag in _CONV_TAGS)


def _zero_bias(module):
    if getattr(module, 'bias', None) is not None:
        module.bias.data.zero_()


def kaiming_weight_init(m, bn_std=0.02):
    _ops.PACK_CACHE.invalidate()   # `.data` edits below do not bump version counters
    if _is_conv(m) or 'Linear' in type(m).__name__:
        nn.init.kaiming_normal_(m.weight)
        _zero_bias(m)
    elif 'BatchNorm' in type(m).__name__:
        m.weight.data.normal_(1.0, bn_std)
        _zero_bias(m)


def gaussian_weight_init(m, conv_std=0.01, bn_std=0.01):
    _ops.PACK_CACHE.invalidate()
    if _is_conv(m):
        m.weight.data.normal_(0, conv_std)
        _zero_bias(m)
    elif 'BatchNorm' in type(m).__name__:
        m.weight.data.normal_(1.0, bn_std)
        _zero_bias(m)
