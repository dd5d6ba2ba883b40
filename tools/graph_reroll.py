"""does the replay time of the captured train step depend on which hardware queues hipGraph's internal branch streams land on?
Captures the step several times, creating k extra dummy streams before each capture (which shifts HIP's round-robin stream ->
hardware-queue assignment), and times the replays.  usage: python tools/graph_reroll.py [captures]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'medical-segmentation3d-toolkit_amd')); sys.path.insert(0, REPO)
import torch
import bench
from segmentation3d.core.seg_train import TrainStep
dev = torch.device('cuda:0')
torch.cuda.set_device(0)
x, t = bench.synthetic_batch(4, 1, 2, 96, dev, 1000)
step = TrainStep('vnet', 1, 2, 'Dice', [0.5, 0.5], device=dev, seed=0, use_graph=True)
keep = []
def timed(fn, n=20):
    for _ in range(3):
        fn(x, t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn(x, t)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
for _ in range(3):
    step(x, t)
print('eager {:.3f} ms'.format(timed(step._eager)), flush=True)
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    step._graph, step._eager_calls = None, 2          # force a new capture
    keep.append(torch.cuda.Stream())                   # shifts the stream -> queue assignment of what is created next
    ms = timed(step)
    print('capture {} (after {} extra streams): replay {:.3f} ms'.format(k, len(keep), ms), flush=True)
