import os, sys, torch
sys.path.insert(0, '/root/repo/medical-segmentation3d-toolkit_amd'); sys.path.insert(0, '/root/repo')
from segmentation3d import _engine as E
def timed(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    c.record(); torch.cuda.synchronize()
    return a.elapsed_time(c) / iters
dev = torch.device('cuda:0')
for N, D, C in [(8, 12, 256), (8, 12, 128), (8, 6, 256), (4, 12, 256), (4, 12, 128), (4, 6, 256), (16, 12, 256), (2, 12, 256), (8, 8, 256), (4, 16, 128), (4, 8, 256)]:
    H = W = D
    x = torch.randn(N, D, H, W, C, device=dev); w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.05
    b = torch.zeros(C, device=dev); y = torch.empty(N, D, H, W, C, device=dev)
    wp = torch.empty(E.query('seg3d_packed_mfma_floats', C, C, 27), device=dev)
    E.call('seg3d_pack_weights_mfma', E.ptr(w), E.ptr(wp), C, C, 27, 27, C * 27, 0, E.stream_ptr())
    st = torch.empty(N, max(1, E.query('seg3d_conv3d_k3_mfma_stats_count', N, D, H, W, C, C)), 2, device=dev)
    ws = torch.empty(max(1, E.query('seg3d_conv3d_k3_mfma_fwd_workspace_floats', N, D, H, W, C, C)), device=dev)
    ms = timed(lambda: E.call('seg3d_conv3d_k3_mfma_fwd', E.ptr(x), E.ptr(wp), E.ptr(b), None, E.ptr(y), E.ptr(st), E.ptr(ws), N, D, H, W, C, C, E.stream_ptr()))
    print('N={} {}^3 C={}: variant {} {:.1f} us {:.1f} TF'.format(N, D, C, E.query('seg3d_conv3d_k3_mfma_variant', N, D, H, W, C, C), ms * 1e3, 2.0 * N * D * H * W * 27 * C * C / ms / 1e9), flush=True)
