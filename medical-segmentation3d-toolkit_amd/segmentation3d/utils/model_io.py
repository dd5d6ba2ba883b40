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


def _chk_dir(model_folder, epoch_idx):
    return os.path.join(model_folder, 'checkpoints', 'chk_{}'.format(epoch_idx))


def _read(path, what):
    assert os.path.isfile(path), '{} file not found: {}'.format(what, path)
    return torch.load(path, map_location='cpu', weights_only=True)   # plain tensors / containers only


def load_checkpoint(epoch_idx, net, opt, save_dir):
    """restore network + optimizer from `<save_dir>/checkpoints/chk_<epoch_idx>`; returns (epoch, batch)"""
    folder = _chk_dir(save_dir, epoch_idx)
    state = _read(os.path.join(folder, 'params.pth'), 'checkpoint')
    getattr(net, 'module', net).load_state_dict(strip_module_prefix(state['state_dict']))
    opt.load_state_dict(_read(os.path.join(folder, 'optimizer.pth'), 'optimizer'))
    return state['epoch'], state['batch']


def checkpoint_state(net, epoch_idx, batch_idx, cfg, max_stride, num_modality):
    """the `params.pth` dictionary (fields of utils/model_io.py:75-84; tensors moved to the host)"""
    weights = OrderedDict((key, value.detach().cpu()) for key, value in net.state_dict().items())
    geometry = {'spacing': cfg.dataset.spacing, 'interpolation': cfg.dataset.interpolation, 'max_stride': max_stride}
    channels = {'in_channels': num_modality, 'out_channels': cfg.dataset.num_classes}
    state = {'epoch': epoch_idx, 'batch': batch_idx, 'net': cfg.net.name, 'state_dict': weights,
             'crop_normalizers': [n.to_dict() for n in cfg.dataset.crop_normalizers]}
    state.update(geometry)
    state.update(channels)
    return state


def save_checkpoint(net, opt, epoch_idx, batch_idx, cfg, max_stride, num_modality):
    """write params.pth / optimizer.pth (+ a copy of train_config.py) for `epoch_idx`"""
    model_folder = os.path.join(cfg.general.save_dir, cfg.general.model_scale)
    folder = _chk_dir(model_folder, epoch_idx)
    os.makedirs(folder, exist_ok=True)
    torch.save(checkpoint_state(net, epoch_idx, batch_idx, cfg, max_stride, num_modality), os.path.join(folder, 'params.pth'))
    torch.save(opt.state_dict(), os.path.join(folder, 'optimizer.pth'))
    config_copy = os.path.join(model_folder, 'train_config.py')
    if os.path.isfile(config_copy):
        shutil.copy(config_copy, os.path.join(folder, 'train_config.py'))
