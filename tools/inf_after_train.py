import os, sys, time, json, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'medical-segmentation3d-toolkit_amd')); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from segmentation3d.core.seg_train import TrainStep
dev = torch.device('cuda:0')
step = TrainStep('vnet', 1, 2, 'Dice', [0.5, 0.5], device=dev, seed=0)
x, t = bench.synthetic_batch(4, 1, 2, 96, dev, 1)
for _ in range(8):
    step(x, t)
torch.cuda.synchronize()
print('mem after train: allocated %.1f GB reserved %.1f GB' % (torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9))
for k in range(3):
    r = bench.time_inference(step.net, (512, 512, 400), 96, 48, 2, 16, dev)
    print(k, r['seconds'], 'reserved %.1f GB' % (torch.cuda.memory_reserved() / 1e9))
torch.cuda.empty_cache()
r = bench.time_inference(step.net, (512, 512, 400), 96, 48, 2, 16, dev)
print('after empty_cache', r['seconds'])
