import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'medical-segmentation3d-toolkit_amd')
GOLDEN = os.path.join(REPO, 'tests', 'golden')
for p in (PKG, REPO):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def golden_npz(name):
    return dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))


def golden_json(name):
    with open(os.path.join(GOLDEN, name + '.json')) as f:
        return json.load(f)


@pytest.fixture(scope='session')
def hip_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail('GPU test selected but no ROCm device is visible')
    return torch.device('cuda:0')
