"""OutputBlock: the V-Net head (reference: network/module/vnet_outblock.py:4-24).

features -> 3x3x3 conv to `out_channels` classes -> GroupNorm -> ReLU -> 1x1x1 conv -> GroupNorm -> Softmax over the
class axis; returns contiguous [N, classes, D, H, W] probabilities.  Reference attribute names: `conv1`, `gn1`, `act1`,
`conv2`, `gn2`, `softmax`.
"""
import torch.nn as nn

from segmentation3d.network.module.layers import Softmax, attach_unit, run_unit

_FIRST, _SECOND = ('conv1', 'gn1', 'act1'), ('conv2', 'gn2', None)


class OutputBlock(nn.Module):

    def __init__(self, in_channels, out_channels):
        super(OutputBlock, self).__init__()
        attach_unit(self, _FIRST, 'k3', in_channels, out_channels)
        attach_unit(self, _SECOND, 'k1', out_channels, out_channels, act=False)
        self.softmax = Softmax(dim=1)

    def forward(self, input):
        hidden = run_unit(self, _FIRST, input, relu=True)
        return self.softmax(run_unit(self, _SECOND, hidden, relu=False))
