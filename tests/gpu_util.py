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
