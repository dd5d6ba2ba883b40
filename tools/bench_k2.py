"""micro-benchmark of every stride-2 layer of vnet(1, 2) at N x D^3 (default 4 x 96^3): the four DownBlock convolutions
(2x2x2 stride 2: gather kernel forward, scatter kernel + skip addend data-gradient, pair-reduce weight gradient) and the four
UpBlock transposed convolutions (scatter forward, gather data-gradient), one launch each, events on the launch stream;
GB/s on ALGORITHMIC bytes (input + output once, + the addend where one is fused)
usage: python tools/bench_k2.py [N D] [--iters K]"""
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


def line(name, ms, nbytes, flop):
    gbs = nbytes / (ms * 1e-3) / 1e9
    print('{:46s} {:8.1f} us  {:6.0f} GB/s algorithmic = {:.2f} of 8 TB/s   {:6.1f} TFLOP/s'.format(name, ms * 1e3, gbs, gbs / 8000.0, flop / ms / 1e9),
          flush=True)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    iters = int(sys.argv[sys.argv.index('--iters') + 1]) if '--iters' in sys.argv else 20
    if '--iters' in sys.argv:
        args = [a for a in args if a != str(iters)]
    N, D = (int(args[0]), int(args[1])) if len(args) >= 2 else (4, 96)
    dev = torch.device('cuda:0')
    total = 0.0
    # DownBlocks: (Cin at the fine level, Cout at the coarse level, fine edge)
    for cin, cout, d in ((16, 32, D), (32, 64, D // 2), (64, 128, D // 4), (128, 256, D // 8)):
        fine, coarse = N * d ** 3, N * (d // 2) ** 3
        x = torch.randn(N, d, d, d, cin, device=dev)
        dy = torch.randn(N, d // 2, d // 2, d // 2, cout, device=dev)
        skipg = torch.randn(N, d, d, d, 2 * cin, device=dev)[..., cin:]      # the skip gradient: a channel slice of the concatenated one
        w = torch.randn(cout, cin, 2, 2, 2, device=dev) * 0.1
        b = torch.zeros(cout, device=dev)
        fl = 2.0 * coarse * 8 * cin * cout
        for name, fn, nb in (('down {:3d}->{:3d} {:2d}^3 fwd (gather)'.format(cin, cout, d), lambda: _ops.conv_forward(x, w, b, 'k2s2', want_stats=True), 4.0 * (fine * cin + coarse * cout)),
                             ('down {:3d}->{:3d} {:2d}^3 dgrad + skip addend (scatter)'.format(cin, cout, d), lambda: _ops.conv_dgrad(dy, w, 'k2s2', addend=skipg), 4.0 * (2 * fine * cin + coarse * cout)),
                             ('down {:3d}->{:3d} {:2d}^3 wgrad'.format(cin, cout, d), lambda: _ops.conv_wgrad(x, dy, (cout, cin, 2, 2, 2), 'k2s2'), 4.0 * (fine * cin + coarse * cout))):
            ms = timed(fn, iters)
            total += ms
            line(name, ms, nb, fl)
    # UpBlocks: convT Cin (coarse) -> Cout (fine)
    for cin, cout, d in ((256, 128, D // 16), (256, 64, D // 8), (128, 32, D // 4), (64, 16, D // 2)):
        coarse, fine = N * d ** 3, N * (2 * d) ** 3
        x = torch.randn(N, d, d, d, cin, device=dev)
        dy = torch.randn(N, 2 * d, 2 * d, 2 * d, cout, device=dev)
        w = torch.randn(cin, cout, 2, 2, 2, device=dev) * 0.1
        b = torch.zeros(cout, device=dev)
        fl = 2.0 * coarse * 8 * cin * cout
        for name, fn, nb in (('up   {:3d}->{:3d} {:2d}^3 fwd (scatter)'.format(cin, cout, d), lambda: _ops.conv_forward(x, w, b, 'convT', want_stats=True), 4.0 * (coarse * cin + fine * cout)),
                             ('up   {:3d}->{:3d} {:2d}^3 dgrad (gather)'.format(cin, cout, d), lambda: _ops.conv_dgrad(dy, w, 'convT'), 4.0 * (coarse * cin + fine * cout)),
                             ('up   {:3d}->{:3d} {:2d}^3 wgrad'.format(cin, cout, d), lambda: _ops.conv_wgrad(x, dy, (cin, cout, 2, 2, 2), 'convT'), 4.0 * (coarse * cin + fine * cout))):
            ms = timed(fn, iters)
            total += ms
            line(name, ms, nb, fl)
    print('sum of the 24 launches: {:.3f} ms'.format(total))


if __name__ == '__main__':
    main()
