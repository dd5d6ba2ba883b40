"""Dice evaluation metric with the reference's name and return convention (utils/metrics.py:5-37 `cal_dsc`):
(dsc, seg_type) with seg_type in {'TN', 'FP', 'FN', 'TP'} decided by the voxel-count threshold.

The counting runs on the GPU: ONE pass over the two label volumes (libseg3d_hip.so: seg3d_label_overlap_counts)
yields area_gt / area_seg / intersection for every requested label; the reference makes three numpy passes per label.
Inputs may be Image3d, numpy arrays or torch tensors (host data is uploaded; device tensors are used in place).
"""
import ctypes

import numpy as np
import torch

from segmentation3d import _engine as E
from segmentation3d.utils.image3d import Image3d

_DTYPE_CODES = {torch.int8: 0, torch.uint8: 1, torch.int16: 2, torch.int32: 3, torch.float32: 4}
MAX_LABELS_PER_PASS = 16


def _as_device_labels(vol, device):
    if isinstance(vol, Image3d):
        vol = vol.array
    if isinstance(vol, np.ndarray):
        if vol.dtype == np.float64:
            vol = vol.astype(np.float32)
        elif vol.dtype in (np.int64, np.uint16, np.uint32, np.uint64):
            vol = vol.astype(np.int32)
        elif vol.dtype == np.bool_:
            vol = vol.astype(np.uint8)
        vol = torch.from_numpy(np.array(vol, order='C'))   # private writable copy (file readers return read-only views)
    if not isinstance(vol, torch.Tensor):
        raise TypeError('label volume must be an Image3d, a numpy array or a torch tensor')
    if vol.dtype == torch.float64:
        vol = vol.float()
    elif vol.dtype == torch.int64:
        vol = vol.int()
    elif vol.dtype == torch.bool:
        vol = vol.to(torch.uint8)
    if vol.dtype not in _DTYPE_CODES:
        raise ValueError('unsupported label dtype {}'.format(vol.dtype))
    return vol.to(device).contiguous()


def label_overlap_counts(gt, seg, labels, device=None):
    """[(area_gt, area_seg, intersection)] per label, as Python ints, from one device pass per 16 labels"""
    if device is None:
        device = gt.device if isinstance(gt, torch.Tensor) and gt.is_cuda else torch.device('cuda', torch.cuda.current_device())
    g, s = _as_device_labels(gt, device), _as_device_labels(seg, device)
    if g.dtype != s.dtype:                      # compare in a common type, like numpy's == does
        common = torch.float32 if torch.float32 in (g.dtype, s.dtype) else torch.int32
        g, s = g.to(common), s.to(common)
    if g.numel() != s.numel():
        raise ValueError('ground truth and segmentation differ in size: {} vs {}'.format(tuple(g.shape), tuple(s.shape)))
    E.require_device(g, s)
    labels = [int(l) for l in labels]
    out = []
    for k0 in range(0, len(labels), MAX_LABELS_PER_PASS):
        chunk = labels[k0:k0 + MAX_LABELS_PER_PASS]
        counts = torch.zeros(3 * len(chunk), dtype=torch.int64, device=device)
        arr = (ctypes.c_int * len(chunk))(*chunk)
        E.call('seg3d_label_overlap_counts', E.ptr(g), E.ptr(s), _DTYPE_CODES[g.dtype], g.numel(), arr, len(chunk),
               E.ptr(counts), E.stream_ptr())
        c = counts.cpu().tolist()
        out.extend((c[3 * k], c[3 * k + 1], c[3 * k + 2]) for k in range(len(chunk)))
    return out


def _classify(area_gt, area_seg, intersection, threshold):
    """utils/metrics.py:24-35"""
    if area_gt < threshold and area_seg < threshold:
        return 1.0, 'TN'
    if area_gt < threshold and area_seg >= threshold:
        return 0.0, 'FP'
    if area_gt >= threshold and area_seg < threshold:
        return 0.0, 'FN'
    return 2 * intersection / (area_gt + area_seg), 'TP'


def cal_dsc(gt_npy, seg_npy, label, threshold):
    """Dice ratio of one label.
    :return: (dsc, seg_type) exactly as the reference's cal_dsc"""
    (a_gt, a_seg, inter), = label_overlap_counts(gt_npy, seg_npy, [label])
    return _classify(a_gt, a_seg, inter, threshold)


def cal_dsc_labels(gt, seg, labels, threshold):
    """[(dsc, seg_type)] for several labels from a single pass over the volumes"""
    return [_classify(a, b, c, threshold) for a, b, c in label_overlap_counts(gt, seg, labels)]
