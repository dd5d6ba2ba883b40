"""debug: record the tensors flowing through backward (gn_backward inputs/outputs, dgrad outputs) in MFMA and direct
mode and print where the two modes start to diverge"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'medical-segmentation3d-toolkit_amd')); sys.path.insert(0, REPO)
from oracle import detgen
from segmentation3d import _ops
from segmentation3d.network import vnet
from segmentation3d.loss.focal_loss import FocalLoss

tag, cin, ncls = 'vnet_1_2', 1, 2
dev = torch.device('cuda:0')
net = vnet.SegmentationNet(cin, ncls)
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
sd_np = detgen.state_dict_like(shapes, 21)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
net = net.to(dev)
x = torch.from_numpy(detgen.normal(22, tag + '/x', (1, cin, 32, 32, 32)))
t = torch.from_numpy(detgen.labels(23, tag + '/t', (1, 1, 32, 32, 32), ncls))
rec = None
o_gnb, o_dgrad = _ops.gn_backward, _ops.conv_dgrad
def gnb(doutn, outn, yn, mean_rstd, gamma, relu, want_dres, want_dbias=True):
    r = o_gnb(doutn, outn, yn, mean_rstd, gamma, relu, want_dres, want_dbias)
    rec.append(('gnb C=%d S=%d relu=%d res=%d' % (yn.shape[-1], yn.shape[1] * yn.shape[2] * yn.shape[3], relu, want_dres),
                dict(dout=doutn.clone(), y=yn.clone(), out=None if outn is None else outn.clone(), mr=mean_rstd.clone(), dy=r[0].clone(),
                     dres=None if r[1] is None else r[1].clone(), dgamma=r[2].clone(), dbeta=r[3].clone(), dbias=None if r[4] is None else r[4].clone())))
    return r
def dgrad(dyn, w, kind):
    r = o_dgrad(dyn, w, kind)
    rec.append(('dgrad %s %s' % (kind, tuple(dyn.shape)), dict(dx=r.clone())))
    return r
_ops.gn_backward, _ops.conv_dgrad = gnb, dgrad
def run(force):
    global rec
    rec = []
    _ops.FORCE_DIRECT = force
    net.zero_grad()
    p = net(x.to(dev))
    FocalLoss(ncls, use_gpu=True)(p, t.to(dev)).backward()
    torch.cuda.synchronize()
    _ops.FORCE_DIRECT = False
    return rec
A, B = run(False), run(True)
def rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))
for (na, da), (nb, db) in zip(A, B):
    assert na == nb
    print('{:40s} '.format(na) + '  '.join('{}={:.1e}'.format(k, rel(da[k], db[k])) for k in da if da[k] is not None))
    if na.startswith('gnb'):
        m = (da['out'] > 0) != (db['out'] > 0) if da['out'] is not None else None
        if m is not None:
            print('{:40s}   relu-mask mismatches: {} of {}; |dout| at mismatches max {:.2e} vs global max {:.2e}'.format(
                '', int(m.sum()), m.numel(), float(db['dout'][m].abs().max()) if int(m.sum()) else 0.0, float(db['dout'].abs().max())))
