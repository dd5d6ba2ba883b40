"""debug: per-unit forward activation error and ReLU-mask mismatches, HIP path vs CPU oracle"""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'medical-segmentation3d-toolkit_amd')); sys.path.insert(0, REPO)
from oracle import detgen, torch_ref
from segmentation3d.network import vnet
from segmentation3d.network.module.conv_gn_relu3 import ConvGnRelu3

tag, cin, ncls = 'vnet_1_2', 1, 2
dev = torch.device('cuda:0')
net = vnet.SegmentationNet(cin, ncls)
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
sd_np = detgen.state_dict_like(shapes, 21)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
net = net.to(dev)
x = torch.from_numpy(detgen.normal(22, tag + '/x', (1, cin, 32, 32, 32)))
sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}

# oracle trace: wrap conv_gn_relu3 to record its output AND the pre-activation (GN output [+ residual])
trace = {}
orig = torch_ref.conv_gn_relu3
def traced(xx, s, prefix, do_act=True):
    y = F.conv3d(xx, s[prefix + '.conv.weight'], s.get(prefix + '.conv.bias'), stride=1, padding=1)
    trace[prefix + '/conv'] = y
    g = F.group_norm(y, 1, s[prefix + '.gn.weight'], s[prefix + '.gn.bias'], 1e-5)
    trace[prefix + '/gn'] = g
    return F.relu(g) if do_act else g
torch_ref.conv_gn_relu3 = traced
with torch.no_grad():
    torch_ref.segmentation_net(x, sd, 'vnet')

got = {}
def mk(name):
    def hook(mod, inp, out):
        got[name] = out.detach().cpu()
    return hook
for name, m in net.named_modules():
    if isinstance(m, ConvGnRelu3):
        m.register_forward_hook(mk(name))
with torch.no_grad():
    net(x.to(dev))
torch.cuda.synchronize()
print('{:35s} {:>10s} {:>12s} {:>10s} {:>10s}'.format('unit', 'n_elem', 'min|preact|', 'out_err', 'gn_mean'))
for name in got:
    g = trace[name + '/gn']
    o = got[name]
    last = not name.endswith(tuple('.ops.{}'.format(i) for i in range(5))) or False
    # units without activation get the residual fused on the GPU side: compare only units with plain relu
    ref_out = F.relu(g)
    err = float((o - ref_out).abs().max()) if o.shape == ref_out.shape else float('nan')
    print('{:35s} {:10d} {:12.3e} {:10.2e} {:10.3f}'.format(name, g.numel(), float(g.abs().min()), err, float(trace[name + '/conv'].mean())))
