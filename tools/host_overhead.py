"""how long does the host need to ISSUE one train step (no sync) vs. the GPU time per step"""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'medical-segmentation3d-toolkit_amd')); sys.path.insert(0, REPO)
import bench
from segmentation3d.core.seg_train import TrainStep
dev = torch.device('cuda:0')
patch = int(sys.argv[1]) if len(sys.argv) > 1 else 96
step = TrainStep('vnet', 1, 2, 'Dice', [0.5, 0.5], device=dev, seed=0)
x, t = bench.synthetic_batch(4, 1, 2, patch, dev, 1)
for _ in range(3):
    step(x, t)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step(x, t)
t_issue = (time.perf_counter() - t0) / 10
torch.cuda.synchronize()
t_total = (time.perf_counter() - t0) / 10
print('patch {}: host issue time per step {:.2f} ms, wall per step {:.2f} ms'.format(patch, t_issue * 1e3, t_total * 1e3))
