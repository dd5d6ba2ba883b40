"""CPU-side checks of the plugin API mirror: import paths, constructor signatures, state_dict contract, error
behaviour (no CPU fallback), config/partition host logic."""
import importlib

import numpy as np
import pytest
import torch

from conftest import golden_json


@pytest.mark.parametrize('plugin', ['vnet', 'vbnet'])
@pytest.mark.parametrize('cin,ncls', [(1, 2), (1, 5), (4, 4)])
def test_state_dict_contract(plugin, cin, ncls):
    ref = golden_json('state_dict_shapes')['{}_{}_{}'.format(plugin, cin, ncls)]
    mod = importlib.import_module('segmentation3d.network.' + plugin)       # core/seg_train.py:72
    net = mod.SegmentationNet(cin, ncls)
    sd = net.state_dict()
    assert [[k, list(v.shape)] for k, v in sd.items()] == ref['keys']       # same keys, order and shapes
    assert sum(p.numel() for p in net.parameters()) == ref['num_params']
    assert net.max_stride() == ref['max_stride'] == 16
    assert all(v.dtype == torch.float32 for v in sd.values())


def test_kaiming_init_matches_reference_semantics():
    from segmentation3d.network import vnet
    net = vnet.SegmentationNet(1, 2)
    torch.manual_seed(0)
    vnet.parameters_kaiming_init(net)
    sd = net.state_dict()
    assert float(sd['in_block.conv.bias'].abs().max()) == 0.0              # conv biases zeroed (weight_init.py:9-10)
    assert torch.all(sd['in_block.gn.weight'] == 1) and torch.all(sd['in_block.gn.bias'] == 0)  # GN untouched
    w = sd['up_32.rblock.ops.0.conv.weight']
    assert abs(float(w.std()) - (2.0 / (32 * 27)) ** 0.5) < 5e-4            # kaiming_normal_, fan_in, gain sqrt(2)
    vnet.parameters_gaussian_init(net)
    assert abs(float(net.state_dict()['up_32.rblock.ops.0.conv.weight'].std()) - 0.01) < 1e-3


def test_module_prefix_checkpoints_load():
    """DataParallel checkpoints carry a 'module.' prefix (core/seg_infer.py:130-142)"""
    from segmentation3d.network import vbnet
    from segmentation3d.utils.model_io import strip_module_prefix
    net = vbnet.SegmentationNet(1, 2)
    sd = {'module.' + k: v.clone() for k, v in net.state_dict().items()}
    net2 = vbnet.SegmentationNet(1, 2)
    net2.load_state_dict(strip_module_prefix(sd))
    for k, v in net.state_dict().items():
        assert torch.equal(v, net2.state_dict()[k])


def test_no_cpu_fallback():
    """the product path must fail loudly on CPU tensors instead of silently using torch CPU ops"""
    from segmentation3d.network import vnet
    from segmentation3d.loss.multi_dice_loss import MultiDiceLoss
    from segmentation3d.loss.focal_loss import FocalLoss
    from segmentation3d._engine import Seg3dEngineError
    net = vnet.SegmentationNet(1, 2)
    with pytest.raises(Seg3dEngineError):
        net(torch.zeros(1, 1, 16, 16, 16))
    p = torch.full((1, 2, 4, 4, 4), 0.5)
    t = torch.zeros(1, 1, 4, 4, 4)
    with pytest.raises(Seg3dEngineError):
        MultiDiceLoss([1, 1], 2, use_gpu=False)(p, t)
    with pytest.raises(Seg3dEngineError):
        FocalLoss(2, use_gpu=False)(p, t)


def test_input_validation():
    from segmentation3d.network import vnet
    net = vnet.SegmentationNet(1, 2)
    with pytest.raises(ValueError):
        net(torch.zeros(1, 1, 20, 16, 16))          # not divisible by max_stride
    with pytest.raises(AssertionError):
        from segmentation3d.loss.multi_dice_loss import MultiDiceLoss
        MultiDiceLoss([1, 1, 1], 2, use_gpu=False)  # len(weights) != num_class (multi_dice_loss.py:16)


def test_reference_unit_test_shapes_are_constructible():
    """network/module/conv_gn_relu3_test.py:19-38 builds ConvGnRelu3 with (k3,s1,p1) and (k2,s2,p0)"""
    from segmentation3d.network.module.conv_gn_relu3 import ConvGnRelu3
    a = ConvGnRelu3(1, 16, 3, 1, 1, do_act=True)
    b = ConvGnRelu3(1, 16, 2, 2, 0, do_act=False)
    assert tuple(a.conv.weight.shape) == (16, 1, 3, 3, 3) and tuple(b.conv.weight.shape) == (16, 1, 2, 2, 2)
    with pytest.raises(ValueError):
        ConvGnRelu3(1, 16, 5, 1, 2)                 # 5x5x5 does not exist in the reference networks


def test_partition_matches_reference_tables():
    from segmentation3d.utils.image_tools import image_partition_by_fixed_size
    for name, case in golden_json('partitions').items():
        starts, ends = image_partition_by_fixed_size(
            (case['size'], case['spacing']), list(case['bbox_start']), list(case['bbox_end']), case['partition_size'],
            case['partition_stride'], case['max_stride'])
        assert starts == case['starts'], name
        assert ends == case['ends'], name
    c = golden_json('partitions')['vol512x512x400_96_48']
    assert len(c['starts']) == 800
