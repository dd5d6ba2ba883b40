"""diagnostic: whole-volume inference timed (a) in a fresh process, (b) after a training phase in the same process, with two
streams and with one, three jobs each; used to chase an intermittent slow inference seen in full bench.py runs
usage: python tools/infer_diag.py"""
import os, sys, time, json
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'medical-segmentation3d-toolkit_amd')); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tools'))
import bench
from clock_watch import ClockWatch
from segmentation3d.core.seg_train import TrainStep
dev = torch.device('cuda:0')
out = {}
step = TrainStep('vnet', 1, 2, 'Dice', [0.5, 0.5], device=dev, seed=0, use_graph=True)
net = step.net


def infer(tag, two):
    with ClockWatch(0.02) as cw:
        r = bench.time_inference(net, (512, 512, 400), 96, 48, 2, 16, dev, two_streams=two)
    s = cw.summary()
    out[tag] = {'runs': r['seconds_all_runs'], 'sclk_mean': s.get('sclk_hz', {}).get('mean'), 'power_max_W': (s.get('power_in_uw', {}).get('max') or 0) / 1e6,
                'mem_alloc_GB': torch.cuda.memory_allocated() / 1e9, 'mem_reserved_GB': torch.cuda.memory_reserved() / 1e9}
    print(tag, json.dumps(out[tag]), flush=True)


infer('fresh_two_streams', True)
infer('fresh_one_stream', False)
x, t = bench.synthetic_batch(4, 1, 2, 96, dev, 1000)
for _ in range(30):
    loss = step(x, t)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(20):
    loss = step(x, t)
torch.cuda.synchronize()
print('train ms/step', 1e3 * (time.time() - t0) / 20, flush=True)
infer('after_train_two_streams', True)
infer('after_train_one_stream', False)
infer('after_train_two_streams_again', True)
