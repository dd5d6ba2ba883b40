"""micro-benchmark: direct implicit-GEMM kernel vs the Winograd F(2,3) and F(2x2,3x3) kernels (forward and weight gradient)
for one C -> C 3x3x3 layer shape, or a list of the network's shapes
usage: python tools/bench_k3.py N D H W C [iters]"""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'medical-segmentation3d-toolkit_amd')); sys.path.insert(0, REPO)
from segmentation3d import _ops, _engine as E

def timed(fn, iters):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    c.record(); torch.cuda.synchronize()
    return a.elapsed_time(c) / iters

def main():
    shapes = [tuple(int(v) for v in sys.argv[1:6])] if len(sys.argv) >= 6 else [(4, 96, 96, 96, 32), (8, 96, 96, 96, 32), (4, 48, 48, 48, 64), (8, 48, 48, 48, 64), (4, 24, 24, 24, 128), (8, 24, 24, 24, 128), (4, 48, 48, 48, 32), (4, 24, 24, 24, 64), (4, 12, 12, 12, 128), (4, 12, 12, 12, 256)]
    iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
    dev = torch.device('cuda:0')
    for N, D, H, W, C in shapes:
        x = torch.randn(N, D, H, W, C, device=dev)
        w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.05
        b = torch.zeros(C, device=dev)
        y = torch.empty(N, D, H, W, C, device=dev)
        fl = 2.0 * N * D * H * W * 27 * C * C
        wp = torch.empty(E.query('seg3d_packed_mfma_floats', C, C, 27), device=dev)
        E.call('seg3d_pack_weights_mfma', E.ptr(w), E.ptr(wp), C, C, 27, 27, C * 27, 0, E.stream_ptr())
        st = torch.empty(N, E.query('seg3d_conv3d_k3_mfma_stats_count', N, D, H, W, C, C), 2, device=dev)
        ws = torch.empty(max(1, E.query('seg3d_conv3d_k3_mfma_fwd_workspace_floats', N, D, H, W, C, C)), device=dev)
        ms = timed(lambda: E.call('seg3d_conv3d_k3_mfma_fwd', E.ptr(x), E.ptr(wp), E.ptr(b), None, E.ptr(y), E.ptr(st), E.ptr(ws), N, D, H, W, C, C, E.stream_ptr()), iters)
        line = 'N={} {}^3 C={}: direct {:8.3f} ms {:6.1f} TF'.format(N, D, C, ms, fl / ms / 1e9)
        if E.query('seg3d_conv3d_k3_wino_supported', N, D, H, W, C, C):
            wq = torch.empty(E.query('seg3d_packed_mfma_floats', C, C, 36), device=dev)
            E.call('seg3d_pack_weights_mfma', E.ptr(w), E.ptr(wq), C, C, 36, 27, C * 27, 0, E.stream_ptr())
            st2 = torch.empty(N, E.query('seg3d_conv3d_k3_wino_stats_count', N, D, H, W, C, C), 2, device=dev)
            ms2 = timed(lambda: E.call('seg3d_conv3d_k3_wino_fwd', E.ptr(x), E.ptr(wq), E.ptr(b), None, E.ptr(y), E.ptr(st2), N, D, H, W, C, C, E.stream_ptr()), iters)
            line += '   winograd {:8.3f} ms {:6.1f} TF (algorithmic)  x{:.2f}'.format(ms2, fl / ms2 / 1e9, ms / ms2)
        if E.query('seg3d_conv3d_k3_wino2d_supported', N, D, H, W, C, C):
            wq2 = torch.empty(E.query('seg3d_packed_mfma_floats', C, C, 48), device=dev)
            E.call('seg3d_pack_weights_mfma', E.ptr(w), E.ptr(wq2), C, C, 48, 27, C * 27, 0, E.stream_ptr())
            st3 = torch.empty(N, E.query('seg3d_conv3d_k3_wino2d_stats_count', N, D, H, W, C, C), 2, device=dev)
            ms3 = timed(lambda: E.call('seg3d_conv3d_k3_wino2d_fwd', E.ptr(x), E.ptr(wq2), E.ptr(b), None, E.ptr(y), E.ptr(st3), N, D, H, W, C, C, E.stream_ptr()), iters)
            line += '   winograd2d {:8.3f} ms {:6.1f} TF (algorithmic)  x{:.2f}'.format(ms3, fl / ms3 / 1e9, ms / ms3)
        print(line, flush=True)
        # weight gradient: 27-tap kernel vs Winograd F(3,2)
        dyv = torch.randn(N, D, H, W, C, device=dev)
        dw = torch.empty(C, C, 3, 3, 3, device=dev)
        wsg = torch.empty(E.query('seg3d_conv3d_k3_mfma_wgrad_workspace_floats', N, D, H, W, C, C), device=dev)
        ms = timed(lambda: E.call('seg3d_conv3d_k3_mfma_wgrad', E.ptr(x), E.ptr(dyv), E.ptr(dw), E.ptr(wsg), N, D, H, W, C, C, 0, E.stream_ptr()), iters)
        line = '    wgrad direct {:8.3f} ms {:6.1f} TF'.format(ms, fl / ms / 1e9)
        if E.query('seg3d_conv3d_k3_wino_wgrad_supported', N, D, H, W, C, C):
            wsq = torch.empty(E.query('seg3d_conv3d_k3_wino_wgrad_workspace_floats', N, D, H, W, C, C), device=dev)
            ms2 = timed(lambda: E.call('seg3d_conv3d_k3_wino_wgrad', E.ptr(x), E.ptr(dyv), E.ptr(dw), E.ptr(wsq), N, D, H, W, C, C, 0, E.stream_ptr()), iters)
            line += '   winograd {:8.3f} ms {:6.1f} TF (algorithmic)  x{:.2f}'.format(ms2, fl / ms2 / 1e9, ms / ms2)
        if E.query('seg3d_conv3d_k3_wino2d_wgrad_supported', N, D, H, W, C, C):
            wsq2 = torch.empty(E.query('seg3d_conv3d_k3_wino2d_wgrad_workspace_floats', N, D, H, W, C, C), device=dev)
            ms3 = timed(lambda: E.call('seg3d_conv3d_k3_wino2d_wgrad', E.ptr(x), E.ptr(dyv), E.ptr(dw), E.ptr(wsq2), N, D, H, W, C, C, 0, E.stream_ptr()), iters)
            line += '   winograd2d {:8.3f} ms {:6.1f} TF (algorithmic)  x{:.2f}'.format(ms3, fl / ms3 / 1e9, ms / ms3)
        print(line, flush=True)

main()
