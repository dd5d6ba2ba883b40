"""V-Net plugin (`cfg.net.name = 'vnet'`): drop-in for the reference's network/vnet.py:10-51.

Discovered by name through `importlib.import_module('segmentation3d.network.' + name)` (core/seg_train.py:72,
core/seg_infer.py:117,127) and must export `SegmentationNet(in_channels, out_channels)` with `forward` and
`max_stride()`, plus `parameters_kaiming_init(net)` / `parameters_gaussian_init(net)`.  Plain residual blocks in every
stage; 116 state_dict tensors, 14,563,296 parameters for 1 -> 2, identical keys and shapes to the reference.
"""
from segmentation3d.network._vnet_base import VNetBase, init_parameters
from segmentation3d.network.module.weight_init import kaiming_weight_init, gaussian_weight_init


class SegmentationNet(VNetBase):
    def __init__(self, in_channels, out_channels):
        super(SegmentationNet, self).__init__(in_channels, out_channels, bottleneck=())


def parameters_kaiming_init(net):
    init_parameters(net, kaiming_weight_init)


def parameters_gaussian_init(net):
    init_parameters(net, gaussian_weight_init)
