"""micro-benchmark of the 2x2x2 stride-2 kernels (gather = Conv3d k2s2 fwd / ConvT dgrad, scatter = ConvT fwd / Conv dgrad,
pair-reduce wgrad) and the thin stem/head kernels at the V-Net's top levels, reported against their HBM bytes.
usage: python tools/bench_k2.py [--iters K]"""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'medical-segmentation3d-toolkit_amd')); sys.path.insert(0, REPO)
from segmentation3d import _ops

def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    c.record()
    torch.cuda.synchronize()
    return a.elapsed_time(c) / iters

def main():
    iters = int(sys.argv[sys.argv.index('--iters') + 1]) if '--iters' in sys.argv else 20
    dev = torch.device('cuda:0')
    N = 4
    for (S, Ci, Co) in [(48, 16, 32), (24, 32, 64), (48, 16, 64), (24, 32, 128)]:   # S = coarse size; fine = 2S
        fine_ci = torch.randn(N, 2 * S, 2 * S, 2 * S, Ci, device=dev)
        coarse_co = torch.randn(N, S, S, S, Co, device=dev)
        w_conv = torch.randn(Co, Ci, 2, 2, 2, device=dev) * 0.1       # Conv3d(Ci -> Co, k2 s2): fine -> coarse
        b = torch.zeros(Co, device=dev)
        mb = lambda *ts: sum(t.numel() for t in ts) * 4 / 1e6
        ms = timeit(lambda: _ops.conv_forward(fine_ci, w_conv, b, 'k2s2', want_stats=True), iters)
        byt = mb(fine_ci, coarse_co)
        print('gather  conv k2s2 fwd   {:3d}^3 {:3d}->{:3d}: {:7.1f} us  {:6.0f} MB  {:5.2f} TB/s'.format(2 * S, Ci, Co, ms * 1e3, byt, byt / ms / 1e3), flush=True)
        ms = timeit(lambda: _ops.conv_dgrad(coarse_co, w_conv, 'k2s2'), iters)
        print('scatter conv k2s2 dgrad {:3d}^3 {:3d}->{:3d}: {:7.1f} us  {:6.0f} MB  {:5.2f} TB/s'.format(S, Co, Ci, ms * 1e3, byt, byt / ms / 1e3), flush=True)
        ms = timeit(lambda: _ops.conv_wgrad(fine_ci, coarse_co, tuple(w_conv.shape), 'k2s2'), iters)
        print('wgrad   conv k2s2       {:3d}^3 {:3d}x{:3d}: {:7.1f} us  {:6.0f} MB  {:5.2f} TB/s'.format(2 * S, Ci, Co, ms * 1e3, byt, byt / ms / 1e3), flush=True)
        # ConvTranspose3d(Co -> Ci): coarse (Co ch) -> fine (Ci ch); weight [Co, Ci, 2,2,2]
        w_t = torch.randn(Co, Ci, 2, 2, 2, device=dev) * 0.1
        bt = torch.zeros(Ci, device=dev)
        ms = timeit(lambda: _ops.conv_forward(coarse_co, w_t, bt, 'convT', want_stats=True), iters)
        print('scatter convT fwd       {:3d}^3 {:3d}->{:3d}: {:7.1f} us  {:6.0f} MB  {:5.2f} TB/s'.format(S, Co, Ci, ms * 1e3, byt, byt / ms / 1e3), flush=True)
        ms = timeit(lambda: _ops.conv_dgrad(fine_ci, w_t, 'convT'), iters)
        print('gather  convT dgrad     {:3d}^3 {:3d}->{:3d}: {:7.1f} us  {:6.0f} MB  {:5.2f} TB/s'.format(2 * S, Ci, Co, ms * 1e3, byt, byt / ms / 1e3), flush=True)
        ms = timeit(lambda: _ops.conv_wgrad(coarse_co, fine_ci, tuple(w_t.shape), 'convT'), iters)
        print('wgrad   convT           {:3d}^3 {:3d}x{:3d}: {:7.1f} us  {:6.0f} MB  {:5.2f} TB/s'.format(2 * S, Ci, Co, ms * 1e3, byt, byt / ms / 1e3), flush=True)
    # thin kernels at 96^3
    S = 96
    x1 = torch.randn(N, S, S, S, 1, device=dev)
    y16 = torch.randn(N, S, S, S, 16, device=dev)
    w_in = torch.randn(16, 1, 3, 3, 3, device=dev) * 0.1
    ms = timeit(lambda: _ops.conv_forward(x1, w_in, torch.zeros(16, device=dev), 'k3', want_stats=True), iters)
    byt = (x1.numel() + y16.numel()) * 4 / 1e6
    print('thin_in  stem fwd  96^3  1->16: {:7.1f} us  {:6.0f} MB  {:5.2f} TB/s'.format(ms * 1e3, byt, byt / ms / 1e3), flush=True)
    ms = timeit(lambda: _ops.conv_wgrad(x1, y16, tuple(w_in.shape), 'k3'), iters)
    print('thin wgrad stem    96^3  1x16: {:7.1f} us  {:6.0f} MB  {:5.2f} TB/s'.format(ms * 1e3, byt, byt / ms / 1e3), flush=True)
    x32 = torch.randn(N, S, S, S, 32, device=dev)
    y2 = torch.randn(N, S, S, S, 2, device=dev)
    w_out = torch.randn(2, 32, 3, 3, 3, device=dev) * 0.1
    byt = (x32.numel() + y2.numel()) * 4 / 1e6
    ms = timeit(lambda: _ops.conv_forward(x32, w_out, torch.zeros(2, device=dev), 'k3', want_stats=True), iters)
    print('thin_out head fwd  96^3 32->2 : {:7.1f} us  {:6.0f} MB  {:5.2f} TB/s'.format(ms * 1e3, byt, byt / ms / 1e3), flush=True)
    ms = timeit(lambda: _ops.conv_dgrad(y2, w_out, 'k3'), iters)
    print('thin_in  head dgrad 96^3 2->32: {:7.1f} us  {:6.0f} MB  {:5.2f} TB/s'.format(ms * 1e3, byt, byt / ms / 1e3), flush=True)
    ms = timeit(lambda: _ops.conv_wgrad(x32, y2, tuple(w_out.shape), 'k3'), iters)
    print('thin wgrad head    96^3 32x2 : {:7.1f} us  {:6.0f} MB  {:5.2f} TB/s'.format(ms * 1e3, byt, byt / ms / 1e3), flush=True)

if __name__ == '__main__':
    main()
