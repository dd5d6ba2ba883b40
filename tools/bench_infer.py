"""whole-volume sliding-window inference timing (used under rocprofv3 for the inference kernel breakdown)
usage: python tools/bench_infer.py [X,Y,Z] [batch] [--single-stream]   (--single-stream: the two half batches of a replay
run one after the other on one stream, so that a kernel trace shows stand-alone kernel durations)"""
import os, sys, time, json
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'medical-segmentation3d-toolkit_amd')); sys.path.insert(0, REPO)
import bench
from segmentation3d.network import vnet
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = vnet.SegmentationNet(1, 2); vnet.parameters_kaiming_init(net); net = net.to(dev).eval()
argv = [a for a in sys.argv[1:] if not a.startswith('--')]
vol = tuple(int(v) for v in (argv[0] if len(argv) > 0 else '512,512,400').split(','))
batch = int(argv[1]) if len(argv) > 1 else 16
r = bench.time_inference(net, vol, 96, 48, 2, batch, dev, two_streams='--single-stream' not in sys.argv)
print(json.dumps(r))
