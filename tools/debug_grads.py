"""debug: per-parameter gradient error of the HIP path vs the CPU oracle, twice (determinism) and with the direct kernels"""
import json, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'medical-segmentation3d-toolkit_amd')); sys.path.insert(0, REPO)
from oracle import detgen, torch_ref
from segmentation3d import _ops
from segmentation3d.network import vnet
from segmentation3d.loss.focal_loss import FocalLoss

tag, cin, ncls = sys.argv[1] if len(sys.argv) > 1 else 'vnet_1_2', 1, 2
dev = torch.device('cuda:0')
net = vnet.SegmentationNet(cin, ncls)
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
sd_np = detgen.state_dict_like(shapes, 21)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
net = net.to(dev)
x = torch.from_numpy(detgen.normal(22, tag + '/x', (1, cin, 32, 32, 32)))
t = torch.from_numpy(detgen.labels(23, tag + '/t', (1, 1, 32, 32, 32), ncls))
sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in sd_np.items()}
probs = torch_ref.segmentation_net(x, sd, 'vnet')
torch_ref.focal_loss(probs, t, ncls, None, 2).backward()
ref = {k: v.grad.double() for k, v in sd.items()}

def run(force):
    _ops.FORCE_DIRECT = force
    net.zero_grad()
    p = net(x.to(dev))
    FocalLoss(ncls, use_gpu=True)(p, t.to(dev)).backward()
    torch.cuda.synchronize()
    _ops.FORCE_DIRECT = False
    return {k: v.grad.detach().cpu().double().clone() for k, v in net.named_parameters()}

a, c = run(False), run(True)
_orig = _ops.gn_stats
_ops.gn_stats = lambda yn, partial=None, eps=1e-5: _orig(yn, None, eps)   # ignore the conv-epilogue partials
b = run(False)
_ops.gn_stats = _orig
names = list(shapes)
print('{:45s} {:>10s} {:>10s} {:>10s}'.format('param', 'mfma_err', 'nopartial', 'direct_err'))
for k in reversed(names):
    e = lambda u, v: float((u - v).abs().max() / (v.abs().max() + 1e-30))
    print('{:45s} {:10.2e} {:10.2e} {:10.2e}'.format(k, e(a[k], ref[k]), e(b[k], ref[k]), e(c[k], ref[k])))
