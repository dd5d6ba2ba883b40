"""debug: inside a real vnet backward, compare every MFMA conv call against the direct kernel on the same tensors"""
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

def rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))

o_dgrad, o_wgrad, o_fwd = _ops.conv_dgrad, _ops.conv_wgrad, _ops.conv_forward
def dgrad(dyn, w, kind):
    r = o_dgrad(dyn, w, kind)
    if kind == 'k3':
        _ops.FORCE_DIRECT = True; d = o_dgrad(dyn, w, kind); _ops.FORCE_DIRECT = False
        print('dgrad {:28s} w{:22s} rel={:.2e} nan={} dy_absmax={:.2e}'.format(str(tuple(dyn.shape)), str(tuple(w.shape)), rel(r, d), bool(torch.isnan(r).any()), float(dyn.abs().max())))
    return r
def wgrad(xn, dyn, ws, kind):
    r = o_wgrad(xn, dyn, ws, kind)
    if kind == 'k3':
        _ops.FORCE_DIRECT = True; d = o_wgrad(xn, dyn, ws, kind); _ops.FORCE_DIRECT = False
        print('wgrad {:28s} w{:22s} rel={:.2e}'.format(str(tuple(xn.shape)), str(ws), rel(r, d)))
    return r
def fwd(xn, w, b, kind, want_stats=False):
    r, p = o_fwd(xn, w, b, kind, want_stats)
    if kind == 'k3':
        _ops.FORCE_DIRECT = True; d, _ = o_fwd(xn, w, b, kind, False); _ops.FORCE_DIRECT = False
        extra = ''
        if p is not None:
            s = p.double().sum(1)[0]; yy = d.double().reshape(-1)
            extra = ' stats_rel=({:.1e},{:.1e})'.format(abs(float(s[0] - yy.sum())) / (abs(float(yy.sum())) + 1e-30), abs(float(s[1] - (yy * yy).sum())) / float((yy * yy).sum()))
        print('fwd   {:28s} w{:22s} rel={:.2e}{}'.format(str(tuple(xn.shape)), str(tuple(w.shape)), rel(r, d), extra))
    return r, p
_ops.conv_dgrad, _ops.conv_wgrad, _ops.conv_forward = dgrad, wgrad, fwd
p = net(x.to(dev))
FocalLoss(ncls, use_gpu=True)(p, t.to(dev)).backward()
torch.cuda.synchronize()

# final gradients of this instrumented (MFMA-path) run vs the CPU oracle
from oracle import torch_ref
sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in sd_np.items()}
probs = torch_ref.segmentation_net(x, sd, 'vnet')
torch_ref.focal_loss(probs, t, ncls, None, 2).backward()
for k in ('out_block.conv1.weight', 'up_32.rblock.ops.0.conv.weight', 'up_32.up_conv.weight', 'up_32.up_conv.bias', 'up_32.up_gn.weight', 'up_64.rblock.ops.1.conv.weight', 'in_block.conv.weight'):
    print('final', k, rel(dict(net.named_parameters())[k].grad.cpu(), sd[k].grad))
