"""Training engine -- MI355X-native replacement of the hot loop of the reference's core/seg_train.py:110-127
(zero_grad -> net(crops) -> loss(outputs, masks) -> backward -> Adam step) and of its `nn.DataParallel` wrap (:76-78).

`TrainStep` is the unit both `train()` and bench.py drive: one process per GPU, the network's 3-D conv / GroupNorm /
loss / Adam arithmetic in HIP kernels, gradients exchanged with a bucketed RCCL all-reduce overlapped with backward
(core/ddp.py) when torch.distributed is initialised.
"""
import importlib
import gc
import os
import shutil
import time

import numpy as np
import torch
import torch.distributed as dist

from segmentation3d import _ops
from segmentation3d.core.ddp import GradientReducer
from segmentation3d.loss.cross_entropy_loss import CrossEntropyLoss
from segmentation3d.loss.focal_loss import FocalLoss
from segmentation3d.loss.multi_dice_loss import MultiDiceLoss
from segmentation3d.optim.fused_adam import FusedAdam


def build_loss(name, num_classes, obj_weight=None, focal_gamma=2, use_gpu=True):
    """loss selection of core/seg_train.py:91-102"""
    if name == 'Focal':
        return FocalLoss(class_num=num_classes, alpha=obj_weight, gamma=focal_gamma, use_gpu=use_gpu)
    if name == 'Dice':
        weights = obj_weight if obj_weight is not None else [1.0 / num_classes] * num_classes
        return MultiDiceLoss(weights=weights, num_class=num_classes, use_gpu=use_gpu)
    if name == 'CE':
        return CrossEntropyLoss()
    raise ValueError('Unknown loss function')


class TrainStep(object):
    """network + loss + FusedAdam (+ gradient reducer when distributed) on one device"""

    def __init__(self, net_name, in_channels, num_classes, loss_name='Dice', obj_weight=None, focal_gamma=2, lr=1e-4,
                 betas=(0.9, 0.999), device=None, seed=0, distributed=None, num_buckets=4, use_graph=False):
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        if self.device.type == 'cuda' and self.device.index is not None:
            torch.cuda.set_device(self.device)   # the engine launches on the current device's stream (_engine.stream_ptr)
        net_module = importlib.import_module('segmentation3d.network.' + net_name)      # core/seg_train.py:72
        torch.manual_seed(seed)
        self.net = net_module.SegmentationNet(in_channels, num_classes)
        self.max_stride = self.net.max_stride()
        net_module.parameters_kaiming_init(self.net)                                    # core/seg_train.py:75
        self.net = self.net.to(self.device)
        self.opt = FusedAdam(self.net.parameters(), lr=lr, betas=betas)                 # core/seg_train.py:83
        _ops.weight_cache(True)   # packed conv weights are refreshed by FusedAdam.step() with one launch per step
        self.loss_func = build_loss(loss_name, num_classes, obj_weight, focal_gamma, use_gpu=True)
        if distributed is None:
            distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.reducer = None
        if distributed:
            self.reducer = GradientReducer(self.opt.flat_grads(), self.opt.flat_layout(), num_buckets=num_buckets)
            self.reducer.broadcast_parameters([f['params'] for f in self.opt._flat if f is not None], src=0)
            self.opt.grad_scale = 1.0 / self.reducer.world_size
            _ops.PACK_CACHE.invalidate()   # the broadcast rewrote the parameters
        # use_graph (single process only): after two eager steps the whole step -- zero_grad, forward, loss, backward,
        # Adam, weight re-pack: ~420 launches -- is captured ONCE in a hipGraph and replayed; the host then costs one
        # replay per step instead of ~10 ms of launch work (what bounds the bf16 mode, whose kernels take about as long).
        # Inputs are copied into static buffers; a new input shape re-captures.  Not combined with the gradient reducer:
        # its collectives stay outside hipGraphs.
        self.use_graph = bool(use_graph) and self.reducer is None
        self._graph, self._gx, self._gt, self._gloss, self._eager_calls = None, None, None, None, 0
        if self.device.type == 'cuda':
            # the weight-gradient side stream is chosen by a probe that synchronises the device: here, not inside the first backward
            _ops.prepare_side_stream(self.device)

    def __call__(self, crops, masks):
        """one optimisation step; returns the (device) loss tensor of this rank's batch"""
        if self.use_graph:
            return self._graphed(crops, masks)
        return self._eager(crops, masks)

    def _graphed(self, crops, masks):
        if self._graph is not None and (self._gx.shape != crops.shape or self._gt.shape != masks.shape or
                                        self._gx.dtype != crops.dtype or self._gt.dtype != masks.dtype):
            self._graph, self._eager_calls = None, 0          # new geometry: warm up and capture again
        if self._graph is None:
            if self._eager_calls < 2:                         # allocator, packed-weight images, plans warm up eagerly
                self._eager_calls += 1
                return self._eager(crops, masks)
            self.opt.use_device_step()
            self._gx, self._gt = crops.clone(), masks.clone()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            host_steps = [None if f is None else f['step'] for f in self.opt._flat]
            try:
                with torch.cuda.graph(graph):
                    self._gloss = self._eager(self._gx, self._gt)
            except Exception as exc:                           # capture refused (driver / allocator state): stay eager
                import warnings
                warnings.warn('TrainStep: hipGraph capture of the train step failed ({}); continuing eagerly'.format(exc))
                torch.cuda.synchronize()
                for f, st in zip(self.opt._flat, host_steps):
                    if f is not None:
                        f['step'] = st
                        f['step_dev'].fill_(int(st))
                self.use_graph = False
                return self._eager(crops, masks)
            for f, st in zip(self.opt._flat, host_steps):     # capture ran the host side of opt.step() once, no kernels
                if f is not None:
                    f['step'] = st
            self._graph = graph
        self._gx.copy_(crops)
        self._gt.copy_(masks)
        self._graph.replay()
        self.opt.note_replayed_step()
        return self._gloss

    def _eager(self, crops, masks):
        self.opt.zero_grad()
        if self.reducer is not None:
            self.reducer.begin_step()
        outputs = self.net(crops)
        loss = self.loss_func(outputs, masks)
        loss.backward()
        if self.reducer is not None:
            self.reducer.finish_step()
        self.opt.step()
        return loss


def epoch_of_batch(batch_idx, batchsize, num_samples, world_size=1):
    """epoch index after `batch_idx` steps (core/seg_train.py:135: batch_idx * batchsize // len(dataset)).  Under data
    parallelism every step consumes world_size * batchsize samples -- each rank draws its 1/world shard of a pass
    (EpochConcateDistributedSampler) -- so the GLOBAL batch is what advances the epoch; with world_size = 1 this is the
    reference's formula."""
    return int(batch_idx) * int(batchsize) * int(world_size) // int(num_samples)


def train(train_config_file, data_iter_factory=None):
    """training engine with the reference's config schema (config/train_config.py) and checkpoint layout.

    Data: by default the GPU-resident `SegmentationDataset` (dataloader/dataset.py: cases read once, crops resampled
    and normalised on the device) driven by `EpochConcateSampler` -- or its distributed variant, one shard per rank --
    exactly as core/seg_train.py:56-70 wires the reference's dataset; alternatively pass
    `data_iter_factory(cfg) -> iterator of (crops, masks[, frames, names])`.  Model folder, config copies, seeding,
    loss selection, logging format and checkpoint cadence follow core/seg_train.py:22-152.
    """
    from segmentation3d.utils.file_io import load_config, setup_logger
    from segmentation3d.utils.model_io import load_checkpoint, save_checkpoint
    assert os.path.isfile(train_config_file), 'Config not found: {}'.format(train_config_file)
    cfg = load_config(train_config_file)
    model_folder = os.path.join(cfg.general.save_dir, cfg.general.model_scale)
    distributed = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank() if distributed else 0
    world_size = dist.get_world_size() if distributed else 1
    if rank == 0:
        if os.path.isdir(model_folder) and cfg.general.resume_epoch < 0:
            shutil.rmtree(model_folder)
        os.makedirs(model_folder, exist_ok=True)
        shutil.copy(train_config_file, os.path.join(model_folder, 'train_config.py'))
        infer_cfg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'config', 'infer_config.py')
        if os.path.isfile(infer_cfg):
            shutil.copy(infer_cfg, os.path.join(cfg.general.save_dir, 'infer_config.py'))
    logger = setup_logger(os.path.join(model_folder, 'train_log.txt'), 'seg3d') if rank == 0 else None
    np.random.seed(cfg.general.seed)
    if cfg.general.num_gpus <= 0:
        raise RuntimeError('segmentation3d HIP engine needs general.num_gpus > 0 (no CPU training path)')
    num_modality = int(getattr(cfg.dataset, 'num_modality', 1))
    _ops.set_activation_dtype(str(getattr(cfg.train, 'compute_dtype', 'fp32')))
    step = TrainStep(cfg.net.name, num_modality, cfg.dataset.num_classes, cfg.loss.name, cfg.loss.obj_weight,
                     cfg.loss.focal_gamma, cfg.train.lr, tuple(cfg.train.betas), seed=cfg.general.seed,
                     use_graph=bool(getattr(cfg.train, 'use_graph', False)))
    assert np.all(np.array(cfg.dataset.crop_size) % step.max_stride == 0), 'crop size not divisible by max stride'
    last_save_epoch, batch_idx = 0, 0
    if cfg.general.resume_epoch >= 0:
        last_save_epoch, batch_idx = load_checkpoint(cfg.general.resume_epoch, step.net, step.opt, model_folder)
    if data_iter_factory is None:
        from segmentation3d.dataloader.dataset import SegmentationDataset, DeviceCropLoader
        from segmentation3d.dataloader.sampler import EpochConcateSampler, EpochConcateDistributedSampler
        dataset = SegmentationDataset(
            imlist_file=cfg.general.imseg_list, num_classes=cfg.dataset.num_classes, spacing=cfg.dataset.spacing,
            crop_size=cfg.dataset.crop_size, sampling_method=cfg.dataset.sampling_method,
            random_translation=cfg.dataset.random_translation, random_scale=cfg.dataset.random_scale,
            interpolation=cfg.dataset.interpolation, crop_normalizers=cfg.dataset.crop_normalizers, device=step.device)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            sampler = EpochConcateDistributedSampler(dataset, cfg.train.epochs, max(0, cfg.general.resume_epoch))
        else:
            sampler = EpochConcateSampler(dataset, cfg.train.epochs)
        batches = DeviceCropLoader(dataset, sampler, cfg.train.batchsize)
        num_samples = len(dataset)
    else:
        batches = data_iter_factory(cfg)
        num_samples = int(getattr(cfg.dataset, 'num_samples', cfg.train.batchsize))
    steps_this_run = 0
    for batch in batches:
        crops, masks = batch[0], batch[1]
        begin_t = time.time()
        crops, masks = crops.to(step.device, non_blocking=True), masks.to(step.device, non_blocking=True)
        loss = step(crops, masks)
        epoch_idx = epoch_of_batch(batch_idx, cfg.train.batchsize, num_samples, world_size)
        batch_idx += 1
        steps_this_run += 1
        if steps_this_run == 3 and bool(getattr(cfg.train, 'gc_freeze', True)):   # (also after a resume: counted per run)
            # everything that lives for the whole run (modules, packed-weight cache, dataset) leaves the collector's young
            # generations: the collections that a step's short-lived autograd objects trigger were the largest single
            # host cost of an eager step (bf16 mode: 10.6 -> 6.5 ms per step)
            gc.collect()
            gc.freeze()
        value = loss.item()
        sample_duration = (time.time() - begin_t) / cfg.train.batchsize
        if logger is not None:
            logger.info('epoch: {}, batch: {}, train_loss: {:.4f}, time: {:.4f} s/vol'.format(epoch_idx, batch_idx, value,
                                                                                             sample_duration))
        if rank == 0 and epoch_idx != 0 and epoch_idx % cfg.train.save_epochs == 0 and last_save_epoch != epoch_idx:
            save_checkpoint(step.net, step.opt, epoch_idx, batch_idx, cfg, step.max_stride, num_modality)
            last_save_epoch = epoch_idx
    return step
