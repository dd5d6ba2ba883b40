"""Checkpoint layout of the reference (utils/model_io.py:7-92): `<model>/checkpoints/chk_<epoch>/{params.pth,
optimizer.pth}`; `params.pth` holds epoch, batch, net name, max_stride, state_dict, spacing, interpolation,
in/out channels and the crop normaliser dictionaries.  The format is the compatibility contract, so parameters stay in
the reference layouts inside `state_dict`; packed / NDHWC weights are derived on the device at run time.
"""
import glob
import os
import shutil
from collections import OrderedDict

import torch


def get_checkpoint_folder(chk_root, epoch):
    """folder of the checkpoint with the given epoch; epoch < 0 selects the latest `chk_<n>`"""
    assert os.path.isdir(chk_root), 'The folder does not exist: {}'.format(chk_root)
    if epoch < 0:
        epochs = [int(os.path.basename(p).split('_')[-1]) for p in glob.glob(os.path.join(chk_root, 'chk_*'))]
        epoch = max(epochs) if epochs else -1
    return os.path.join(chk_root, 'chk_{}'.format(epoch))


def strip_module_prefix(state_dict):
    """DataParallel-trained checkpoints prefix every key with 'module.' (core/seg_infer.py:130-142)"""
    if not any(k.startswith('module.') for k in state_dict):
        return state_dict
    return OrderedDict((k[len('module.'):] if k.startswith('module.') else k, v) for k, v in state_dict.items())


def load_checkpoint(epoch_idx, net, opt, save_dir):
    """restore network + optimizer from `<save_dir>/checkpoints/chk_<epoch_idx>`; returns (epoch, batch)"""
    chk_dir = os.path.join(save_dir, 'checkpoints', 'chk_{}'.format(epoch_idx))
    chk_file = os.path.join(chk_dir, 'params.pth')
    assert os.path.isfile(chk_file), 'checkpoint file not found: {}'.format(chk_file)
    state = torch.load(chk_file, map_location='cpu', weights_only=True)
    target = net.module if hasattr(net, 'module') else net
    target.load_state_dict(strip_module_prefix(state['state_dict']))
    opt_file = os.path.join(chk_dir, 'optimizer.pth')
    assert os.path.isfile(opt_file), 'optimizer file not found: {}'.format(opt_file)
    opt.load_state_dict(torch.load(opt_file, map_location='cpu', weights_only=True))
    return state['epoch'], state['batch']


def save_checkpoint(net, opt, epoch_idx, batch_idx, cfg, max_stride, num_modality):
    """write params.pth / optimizer.pth (+ a copy of train_config.py) for `epoch_idx`"""
    model_folder = os.path.join(cfg.general.save_dir, cfg.general.model_scale)
    chk_folder = os.path.join(model_folder, 'checkpoints', 'chk_{}'.format(epoch_idx))
    os.makedirs(chk_folder, exist_ok=True)
    state = {'epoch': epoch_idx,
             'batch': batch_idx,
             'net': cfg.net.name,
             'max_stride': max_stride,
             'state_dict': OrderedDict((k, v.detach().cpu()) for k, v in net.state_dict().items()),
             'spacing': cfg.dataset.spacing,
             'interpolation': cfg.dataset.interpolation,
             'in_channels': num_modality,
             'out_channels': cfg.dataset.num_classes,
             'crop_normalizers': [normalizer.to_dict() for normalizer in cfg.dataset.crop_normalizers]}
    torch.save(state, os.path.join(chk_folder, 'params.pth'))
    torch.save(opt.state_dict(), os.path.join(chk_folder, 'optimizer.pth'))
    cfg_copy = os.path.join(model_folder, 'train_config.py')
    if os.path.isfile(cfg_copy):
        shutil.copy(cfg_copy, os.path.join(chk_folder, 'train_config.py'))
