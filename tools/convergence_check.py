"""trains vnet(1,2) for K steps on a fixed synthetic batch with (a) the direct MFMA kernels, (b) the F(2,3)/F(3,2) Winograd
kernels, (c) the F(2x2,3x3)/F(3x3,2x2) kernels (default) from the same seed and prints the loss trajectories side by side:
the Winograd forms differ from the direct kernels by rounding only, so the curves must agree to ~1e-5 early on and stay
together.  usage: python tools/convergence_check.py [steps]"""
import os, sys, json
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'medical-segmentation3d-toolkit_amd')); sys.path.insert(0, REPO)
import bench
from segmentation3d import _ops
from segmentation3d.core.seg_train import TrainStep
K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device('cuda:0')
x, t = bench.synthetic_batch(4, 1, 2, 96, dev, 1000)
curves = {}
for name, w1, w2 in (('direct', False, False), ('winograd_1d', True, False), ('winograd_2d', True, True)):
    _ops.WINOGRAD, _ops.WINOGRAD2D = w1, w2
    _ops.PACK_CACHE.clear()
    step = TrainStep('vnet', 1, 2, 'Dice', [0.5, 0.5], lr=1e-3, device=dev, seed=0, use_graph=False)
    losses = []
    for i in range(K):
        losses.append(float(step(x, t).detach()))
    curves[name] = losses
    del step
    torch.cuda.synchronize()
pick = [0, 1, 2, 5, 10, 20, 50, 100, 150, K - 1]
pick = [i for i in pick if i < K]
print('step   ' + '  '.join('{:>12s}'.format(n) for n in curves))
for i in pick:
    print('{:5d}  '.format(i) + '  '.join('{:12.6f}'.format(curves[n][i]) for n in curves))
d1 = max(abs(a - b) for a, b in zip(curves['direct'][:10], curves['winograd_2d'][:10]))
dl = abs(curves['direct'][-1] - curves['winograd_2d'][-1])
print(json.dumps({'steps': K, 'max_abs_loss_diff_first_10_steps_direct_vs_2d': d1, 'abs_loss_diff_last_step': dl,
                  'final': {n: c[-1] for n, c in curves.items()}}))
