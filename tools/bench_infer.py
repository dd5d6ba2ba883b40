"""whole-volume sliding-window inference timing (used under rocprofv3 for the inference kernel breakdown)"""
import os, sys, time, json
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'medical-segmentation3d-toolkit_amd')); sys.path.insert(0, REPO)
import bench
from segmentation3d.network import vnet
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = vnet.SegmentationNet(1, 2); vnet.parameters_kaiming_init(net); net = net.to(dev).eval()
vol = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '512,512,400').split(','))
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
r = bench.time_inference(net, vol, 96, 48, 2, batch, dev)
print(json.dumps(r))
r = bench.time_inference(net, vol, 96, 48, 2, batch, dev)
print(json.dumps(r))
