"""time the F(3x3,2x2) weight-gradient entry point (kernel + slab reduce) for one C -> C layer shape; SEG3D_HIP_LIB selects the
library build (A/B of kernel variants on one box).  usage: python tools/bench_wgrad.py N D H W C [iters]"""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'medical-segmentation3d-toolkit_amd')); sys.path.insert(0, REPO)
from segmentation3d import _engine as E

N, D, H, W, C = (int(v) for v in sys.argv[1:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 20
dev = torch.device('cuda:0')
x = torch.randn(N, D, H, W, C, device=dev)
dy = torch.randn(N, D, H, W, C, device=dev)
dw = torch.empty(C, C, 3, 3, 3, device=dev)
ws = torch.empty(E.query('seg3d_conv3d_k3_wino2d_wgrad_workspace_floats', N, D, H, W, C, C), device=dev)
fn = lambda: E.call('seg3d_conv3d_k3_wino2d_wgrad', E.ptr(x), E.ptr(dy), E.ptr(dw), E.ptr(ws), N, D, H, W, C, C, 0, E.stream_ptr())
for _ in range(3): fn()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(iters): fn()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / iters
print('N={} {}^3 C={}: wgrad {:.4f} ms  {:.1f} TFLOP/s algorithmic  ({})'.format(N, D, C, ms, 2.0 * N * D * H * W * 27 * C * C / ms / 1e9,
                                                                        os.path.basename(os.environ.get('SEG3D_HIP_LIB', 'default'))))
