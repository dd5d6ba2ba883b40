"""host-side cost of one train step: cProfile over steps on a 32^3 patch (GPU time negligible there), top functions by
cumulative and own time.  usage: python tools/host_profile.py [fp32|bf16] [steps]"""
import cProfile, os, pstats, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'medical-segmentation3d-toolkit_amd')); sys.path.insert(0, REPO)
from segmentation3d import _ops
from segmentation3d.core.seg_train import TrainStep
import bench

mode = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
_ops.set_activation_dtype(mode)
dev = torch.device('cuda:0')
step = TrainStep('vnet', 1, 2, 'Dice', [0.5, 0.5], device=dev, seed=0)
x, t = bench.synthetic_batch(4, 1, 2, 32, dev, 1)
for _ in range(5):
    step(x, t)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step(x, t)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print('mode {}: host issue time per step {:.3f} ms, with final sync {:.3f} ms'.format(mode, 1e3 * t_issue / steps, 1e3 * t_all / steps))
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step(x, t)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
