"""helpers shared by the GPU parity tests"""
import os

import numpy as np
import torch

from conftest import REPO

_REPORT = os.path.join(REPO, 'gpurun_out', 'parity_report.txt')


def report(name, **errs):
    """append max-error figures to gpurun_out/parity_report.txt (merged back from the GPU box)"""
    os.makedirs(os.path.dirname(_REPORT), exist_ok=True)
    with open(_REPORT, 'a') as f:
        f.write('{:60s} {}\n'.format(name, '  '.join('{}={:.3e}'.format(k, v) for k, v in errs.items())))


def max_err(a, b):
    a = a.detach().cpu().double() if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a)).double()
    b = b.detach().cpu().double() if isinstance(b, torch.Tensor) else torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max())


def rel_err(a, b):
    """max |a-b| / max|b|"""
    a = a.detach().cpu().double() if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a)).double()
    b = b.detach().cpu().double() if isinstance(b, torch.Tensor) else torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def mask_flips(mask, ref_mask, ref_probs):
    """arg-max masks are integer work: a voxel may only differ from the oracle's where the oracle's two largest class
    probabilities tie within the fp32 parity bar.  Returns (number of differing voxels, the largest top1 - top2 gap of the
    oracle among them -- 0.0 when the masks are equal).  `ref_probs` is [C, Z, Y, X]."""
    mask = np.asarray(mask)
    ref_mask = np.asarray(ref_mask)
    assert mask.shape == ref_mask.shape == tuple(ref_probs.shape[1:]), (mask.shape, ref_mask.shape, ref_probs.shape)
    diff = mask != ref_mask
    n = int(diff.sum())
    if n == 0:
        return 0, 0.0
    p = np.sort(np.asarray(ref_probs, dtype=np.float64)[:, diff], axis=0)
    return n, float((p[-1] - p[-2]).max())


# every mismatching voxel must have an oracle top1 - top2 gap below this (2 x the 1e-4 bar on each probability)
TIE_GAP = 2e-4


def check_mask(name, mask, probs, ref_probs, pick_largest_cc=False, remove_small_cc=0):
    """the integer half of the whole-volume parity: (1) the arg-max of the device probabilities differs from the arg-max of
    the oracle's only at oracle ties (gap < TIE_GAP), (2) the returned mask is EXACTLY the reference's post-processing
    (oracle/numpy_ref.connected_component_filter) of the device's own arg-max.  Reports the observed flip count."""
    from oracle import numpy_ref
    probs = np.asarray(probs)
    ref_probs = np.asarray(ref_probs)
    raw = np.argmax(probs, axis=0).astype(np.int8)
    raw_ref = np.argmax(ref_probs, axis=0).astype(np.int8)
    flips, gap = mask_flips(raw, raw_ref, ref_probs)
    expect = raw
    labels = list(range(1, probs.shape[0]))
    if pick_largest_cc and labels:
        expect = numpy_ref.connected_component_filter(expect, labels, 'largest')
    if remove_small_cc > 0 and labels:
        expect = numpy_ref.connected_component_filter(expect, labels, 'min_size', remove_small_cc)
    exact = bool(np.array_equal(np.asarray(mask), expect))
    report(name + '_mask', argmax_flips=float(flips), voxels=float(raw.size), worst_oracle_gap=gap, postproc_exact=float(exact))
    assert gap < TIE_GAP, '{}: {} arg-max flips, one at an oracle top1-top2 gap of {}'.format(name, flips, gap)
    assert exact, '{}: mask differs from the post-processing of its own arg-max'.format(name)
    return flips
