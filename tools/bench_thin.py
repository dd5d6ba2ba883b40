"""micro-benchmark of the fp32 thin 3x3x3 kernels (stem / head forward, data-gradient, weight gradients) and the
stride-2 kernels at the top level: one launch each, events on the launch stream, GB/s on ALGORITHMIC bytes
(input + output once) against the 8 TB/s HBM peak.
usage: python tools/bench_thin.py [N D H W] [--iters K] [--classes C]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'medical-segmentation3d-toolkit_amd'))
sys.path.insert(0, REPO)
from segmentation3d import _ops, _engine as E   # noqa: E402


def timed(fn, iters):
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


def line(name, ms, nbytes):
    gbs = nbytes / (ms * 1e-3) / 1e9
    print('{:44s} {:8.1f} us  {:7.0f} GB/s on algorithmic bytes = {:.2f} of 8 TB/s'.format(name, ms * 1e3, gbs, gbs / 8000.0),
          flush=True)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    iters = 20
    if '--iters' in sys.argv:
        iters = int(sys.argv[sys.argv.index('--iters') + 1])
        args = [a for a in args if a != str(iters)]
    ncls = 2
    if '--classes' in sys.argv:
        ncls = int(sys.argv[sys.argv.index('--classes') + 1])
        args = [a for a in args if a != str(ncls)]
    N, D, H, W = [int(v) for v in args[:4]] if len(args) >= 4 else (4, 96, 96, 96)
    dev = torch.device('cuda:0')
    vox = N * D * H * W
    x1 = torch.randn(N, D, H, W, 1, device=dev)
    x16 = torch.randn(N, D, H, W, 16, device=dev)
    x32 = torch.randn(N, D, H, W, 32, device=dev)
    dyc = torch.randn(N, D, H, W, ncls, device=dev)
    w_stem = torch.randn(16, 1, 3, 3, 3, device=dev) * 0.2
    w_head = torch.randn(ncls, 32, 3, 3, 3, device=dev) * 0.05
    b16, bc = torch.zeros(16, device=dev), torch.zeros(ncls, device=dev)

    # head forward: x32 -> ncls
    _ops.FORCE_DIRECT = False
    line('head fwd 32->{} (auto route)'.format(ncls), timed(lambda: _ops.conv_forward(x32, w_head, bc, 'k3', want_stats=True), iters),
         4.0 * vox * (32 + ncls))
    CO = 2 if ncls <= 2 else (4 if ncls <= 4 else 8)
    wq = torch.empty((32 + 7) // 8 * 27 * 8 * CO, device=dev)
    E.call('seg3d_pack_weights_thin_out', E.ptr(w_head), E.ptr(wq), 32, ncls, CO, 27, 32 * 27, 0, E.stream_ptr())
    y = torch.empty(N, D, H, W, ncls, device=dev)
    st = torch.empty(N, E.query('seg3d_conv3d_k3_thin_out_stats_count', D, H, W), 2, device=dev)
    line('head fwd 32->{} (VALU kernel)'.format(ncls),
         timed(lambda: E.call('seg3d_conv3d_k3_thin_out_fwd', E.ptr(x32), E.ptr(wq), E.ptr(bc), E.ptr(y), E.ptr(st), N, D, H, W,
                              32, ncls, CO, E.stream_ptr()), iters), 4.0 * vox * (32 + ncls))
    # head data-gradient: dy (ncls) -> dx (32)
    line('head dgrad {}->32'.format(ncls), timed(lambda: _ops.conv_dgrad(dyc, w_head, 'k3'), iters), 4.0 * vox * (32 + ncls))
    # head weight gradient
    line('head wgrad (x32, dy{})'.format(ncls), timed(lambda: _ops.conv_wgrad(x32, dyc, (ncls, 32, 3, 3, 3), 'k3'), iters),
         4.0 * vox * (32 + ncls))
    # stem
    line('stem fwd 1->16', timed(lambda: _ops.conv_forward(x1, w_stem, b16, 'k3', want_stats=True), iters), 4.0 * vox * 17)
    line('stem wgrad (x1, dy16)', timed(lambda: _ops.conv_wgrad(x1, x16, (16, 1, 3, 3, 3), 'k3'), iters), 4.0 * vox * 17)
    # stride-2 layers of the top level (16 -> 32 down, 64 -> 16 up at half resolution)
    if D % 2 == 0 and H % 2 == 0 and W % 2 == 0:
        w_dn = torch.randn(32, 16, 2, 2, 2, device=dev) * 0.1
        b32 = torch.zeros(32, device=dev)
        xh64 = torch.randn(N, D // 2, H // 2, W // 2, 64, device=dev)
        xh32 = torch.randn(N, D // 2, H // 2, W // 2, 32, device=dev)
        w_up = torch.randn(64, 16, 2, 2, 2, device=dev) * 0.1
        line('down conv k2s2 16->32 fwd', timed(lambda: _ops.conv_forward(x16, w_dn, b32, 'k2s2', want_stats=True), iters),
             4.0 * vox * (16 + 32 / 8.0))
        line('down conv k2s2 dgrad 32->16', timed(lambda: _ops.conv_dgrad(xh32, w_dn, 'k2s2'), iters), 4.0 * vox * (16 + 32 / 8.0))
        line('down conv k2s2 wgrad', timed(lambda: _ops.conv_wgrad(x16, xh32, (32, 16, 2, 2, 2), 'k2s2'), iters),
             4.0 * vox * (16 + 32 / 8.0))
        line('up convT k2s2 64->16 fwd', timed(lambda: _ops.conv_forward(xh64, w_up, b16, 'convT', want_stats=True), iters),
             4.0 * vox * (16 + 64 / 8.0))
        line('up convT k2s2 dgrad 16->64', timed(lambda: _ops.conv_dgrad(x16, w_up, 'convT'), iters), 4.0 * vox * (16 + 64 / 8.0))
        line('up convT k2s2 wgrad', timed(lambda: _ops.conv_wgrad(xh64, x16, (64, 16, 2, 2, 2), 'convT'), iters),
             4.0 * vox * (16 + 64 / 8.0))


if __name__ == '__main__':
    main()
