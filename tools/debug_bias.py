"""debug: error statistics of MFMA vs direct k3 conv against an fp64 CPU reference on gradient-like (small, sparse) data"""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'medical-segmentation3d-toolkit_amd')); sys.path.insert(0, REPO)
from oracle import detgen
from segmentation3d import _ops
dev = torch.device('cuda:0')
for (N, C, D) in ((1, 32, 32), (1, 64, 16)):
    x = torch.from_numpy(detgen.normal(1, 'bx%d' % C, (N, C, D, D, D)))
    m = torch.from_numpy(detgen.uniform(2, 'bm%d' % C, (N, C, D, D, D)) > 0.5)
    x = (x * m * 1e-5).float()                       # sparse, tiny values like a ReLU-masked gradient
    w = torch.from_numpy(detgen.normal(3, 'bw%d' % C, (C, C, 3, 3, 3), std=(2.0 / (27 * C)) ** 0.5))
    ref = F.conv3d(x.double(), w.double(), None, padding=1)
    cpu32 = F.conv3d(x, w, None, padding=1)
    xd, wd = x.to(dev), w.to(dev)
    outs = {'cpu32': cpu32.double()}
    for name, force in (('mfma', False), ('direct', True)):
        _ops.FORCE_DIRECT = force
        y, _ = _ops.conv_forward(_ops.to_ndhwc(xd), wd, None, 'k3')
        _ops.FORCE_DIRECT = False
        outs[name] = _ops.from_ndhwc(y).cpu().double()
    print('C=%d D=%d' % (C, D))
    for name, o in outs.items():
        e = o - ref
        print('  {:7s} max_err/max={:.2e}  rms_err/rms={:.2e}  mean_err/rms_err={:+.3f}  chan_sum_relerr={:.2e}  frac_exact={:.3f}'.format(
            name, float(e.abs().max() / ref.abs().max()), float(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()),
            float(e.mean() / (e.pow(2).mean().sqrt() + 1e-300)),
            float((e.sum((0, 2, 3, 4)).abs() / (ref.sum((0, 2, 3, 4)).abs() + 1e-300)).max()),
            float((e == 0).double().mean())))
