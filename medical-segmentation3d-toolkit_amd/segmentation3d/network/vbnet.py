"""VB-Net plugin (`cfg.net.name = 'vbnet'`, the default in config/train_config.py:106): drop-in for the reference's
network/vbnet.py:11-53.  Same skeleton as V-Net; the 64/128/256-wide encoder stages and the 256/128-wide decoder
stages use bottleneck residual blocks (C -> C/4 -> C/4 -> C, all 3x3x3).  228 state_dict tensors, 8,664,096
parameters for 1 -> 2.
"""
from segmentation3d.network._vnet_base import VNetBase, init_parameters
from segmentation3d.network.module.weight_init import kaiming_weight_init, gaussian_weight_init

BOTTLENECK_STAGES = ('down_64', 'down_128', 'down_256', 'up_256', 'up_128')  # vbnet.py:27-32 compression=True


class SegmentationNet(VNetBase):
    def __init__(self, in_channels, out_channels):
        super(SegmentationNet, self).__init__(in_channels, out_channels, bottleneck=BOTTLENECK_STAGES)


def parameters_kaiming_init(net):
    init_parameters(net, kaiming_weight_init)


def parameters_gaussian_init(net):
    init_parameters(net, gaussian_weight_init)
