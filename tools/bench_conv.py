"""micro-benchmark of single conv launches (events on the launch stream); used for kernel A/B work and PMC runs.
usage: python tools/bench_conv.py [fwd|fwd16|head16|headwgrad16|wgrad|wgrad16|all] [N D H W Cin Cout] [--iters K]   (fwd16 / wgrad16 = bf16-input kernels)"""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'medical-segmentation3d-toolkit_amd')); sys.path.insert(0, REPO)
from segmentation3d import _ops, _engine as E

def run(kind, N, D, H, W, Cin, Cout, iters=10):
    dev = torch.device('cuda:0')
    x = torch.randn(N, D, H, W, Cin, device=dev)
    dy = torch.randn(N, D, H, W, Cout, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) * 0.05
    b = torch.zeros(Cout, device=dev)
    if kind == 'fwd':
        wp = _ops._pack_mfma(w, Cin, Cout, 27, 27, Cin * 27)
        y = torch.empty(N, D, H, W, Cout, device=dev)
        cnt = E.query('seg3d_conv3d_k3_mfma_stats_count', N, D, H, W, Cin, Cout)
        nws = E.query('seg3d_conv3d_k3_mfma_fwd_workspace_floats', N, D, H, W, Cin, Cout)
        wsp = torch.empty(max(nws, 1), device=dev)
        st = torch.empty(N, cnt, 2, device=dev)
        ad = torch.randn(N, D, H, W, Cout, device=dev) if '--addend' in sys.argv else None
        nostats = '--nostats' in sys.argv
        fn = lambda: E.call('seg3d_conv3d_k3_mfma_fwd', E.ptr(x), E.ptr(wp), None if nostats else E.ptr(b), E.ptr(ad), E.ptr(y), None if nostats else E.ptr(st), E.ptr(wsp), N, D, H, W, Cin, Cout, 0, E.stream_ptr())
    elif kind == 'fwd16':
        xb = x.bfloat16()
        wp = torch.empty(E.query('seg3d_packed_mfma_bf16_elems', Cin, Cout, 27), dtype=torch.bfloat16, device=dev)
        E.call('seg3d_pack_weights_mfma_bf16', E.ptr(w), E.ptr(wp), Cin, Cout, 27, 27, Cin * 27, 0, E.stream_ptr())
        y = torch.empty(N, D, H, W, Cout, device=dev)
        cnt = E.query('seg3d_conv3d_k3_bf16_stats_count', N, D, H, W, Cin, Cout)
        nws = E.query('seg3d_conv3d_k3_bf16_fwd_workspace_floats', N, D, H, W, Cin, Cout)
        wsp = torch.empty(max(nws, 1), device=dev)
        st = torch.empty(N, cnt, 2, device=dev)
        print('variant', E.query('seg3d_conv3d_k3_bf16_variant', N, D, H, W, Cin, Cout), 'ks', nws // (N * D * H * W * Cout))
        fn = lambda: E.call('seg3d_conv3d_k3_bf16_fwd', E.ptr(xb), E.ptr(wp), E.ptr(b), None, E.ptr(y), E.ptr(st), E.ptr(wsp), N, D, H, W, Cin, Cout, 0, E.stream_ptr())
    elif kind == 'head16':   # thin-output head on bf16 activations: MFMA kernel, or the VALU kernel with --valu
        xb = x.bfloat16()
        y = torch.empty(N, D, H, W, Cout, device=dev)
        st = torch.empty(N, E.query('seg3d_conv3d_k3_thin_out_stats_count', D, H, W), 2, device=dev)
        if '--valu' in sys.argv:
            CO = 2 if Cout <= 2 else (4 if Cout <= 4 else 8)
            wq = torch.empty((Cin + 7) // 8 * 27 * 8 * CO, device=dev)
            E.call('seg3d_pack_weights_thin_out', E.ptr(w), E.ptr(wq), Cin, Cout, CO, 27, Cin * 27, 0, E.stream_ptr())
            fn = lambda: E.call('seg3d_conv3d_k3_thin_out_bf16_fwd', E.ptr(xb), E.ptr(wq), E.ptr(b), E.ptr(y), E.ptr(st), N, D, H, W, Cin, Cout, CO, E.stream_ptr())
        else:
            wq = torch.empty(E.query('seg3d_thin_out_mfma_packed_elems', Cin), dtype=torch.bfloat16, device=dev)
            E.call('seg3d_pack_weights_thin_out_mfma', E.ptr(w), E.ptr(wq), Cin, Cout, 27, Cin * 27, 0, E.stream_ptr())
            fn = lambda: E.call('seg3d_conv3d_k3_thin_out_mfma_fwd', E.ptr(xb), E.ptr(wq), E.ptr(b), E.ptr(y), E.ptr(st), N, D, H, W, Cin, Cout, E.stream_ptr())
    elif kind == 'headwgrad16':   # head weight gradient in bf16 mode: thin = fp32 dy (Cout), fat = bf16 x (Cin)
        xb = x.bfloat16()
        ws = torch.empty(E.query('seg3d_k3_thin_wgrad_workspace_floats', N, D, H, W, Cout, Cin), device=dev)
        dw = torch.empty(Cout, Cin, 3, 3, 3, device=dev)
        fn = lambda: E.call('seg3d_k3_thin_wgrad_fatbf16', E.ptr(dy), E.ptr(xb), E.ptr(dw), E.ptr(ws), N, D, H, W, Cout, Cin, Cin * 27, 27, 1, 0, E.stream_ptr())
    elif kind == 'wgrad16':
        xb, dyb = x.bfloat16(), dy.bfloat16()
        ws = torch.empty(E.query('seg3d_conv3d_k3_bf16_wgrad_workspace_floats', N, D, H, W, Cin, Cout), device=dev)
        dw = torch.empty(Cout, Cin, 3, 3, 3, device=dev)
        fn = lambda: E.call('seg3d_conv3d_k3_bf16_wgrad', E.ptr(xb), E.ptr(dyb), E.ptr(dw), E.ptr(ws), N, D, H, W, Cin, Cout, 0, E.stream_ptr())
    else:
        ws = torch.empty(E.query('seg3d_conv3d_k3_mfma_wgrad_workspace_floats', N, D, H, W, Cin, Cout), device=dev)
        dw = torch.empty(Cout, Cin, 3, 3, 3, device=dev)
        fn = lambda: E.call('seg3d_conv3d_k3_mfma_wgrad', E.ptr(x), E.ptr(dy), E.ptr(dw), E.ptr(ws), N, D, H, W, Cin, Cout, 0, E.stream_ptr())
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    c.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(c) / iters
    fl = 2.0 * N * D * H * W * 27 * Cin * Cout
    print('{:5s} N={} {}x{}x{} {}->{}  {:8.3f} ms  {:7.2f} TFLOP/s'.format(kind, N, D, H, W, Cin, Cout, ms, fl / ms / 1e9), flush=True)

if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    iters = 10
    if '--iters' in sys.argv:
        iters = int(sys.argv[sys.argv.index('--iters') + 1]); args = [a for a in args if a != str(iters)]
    kind = args[0] if args else 'all'
    shapes = [tuple(int(v) for v in args[1:7])] if len(args) >= 7 else [
        (4, 96, 96, 96, 32, 32), (4, 48, 48, 48, 64, 64), (4, 48, 48, 48, 32, 32), (4, 24, 24, 24, 128, 128),
        (4, 24, 24, 24, 64, 64), (4, 12, 12, 12, 256, 256), (4, 12, 12, 12, 128, 128), (4, 6, 6, 6, 256, 256)]
    for s in shapes:
        for k in (['fwd', 'wgrad'] if kind == 'all' else [kind]):
            run(k, *s, iters=iters)
