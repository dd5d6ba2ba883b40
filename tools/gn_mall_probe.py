"""probe: does GroupNorm backward run faster sample by sample (second read of (dout, y) from the 256 MB Infinity Cache)
than over the whole batch?  usage: python tools/gn_mall_probe.py [N D C]"""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'medical-segmentation3d-toolkit_amd')); sys.path.insert(0, REPO)
from segmentation3d import _ops

def timed(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    c.record(); torch.cuda.synchronize()
    return a.elapsed_time(c) / iters * 1e3

def main():
    shapes = [tuple(int(v) for v in sys.argv[1:4])] if len(sys.argv) >= 4 else [(4, 96, 32), (4, 96, 16), (4, 48, 64), (4, 48, 32), (8, 96, 32)]
    dev = torch.device('cuda:0')
    for N, D, C in shapes:
        y = torch.randn(N, D, D, D, C, device=dev)
        dout = torch.randn(N, D, D, D, C, device=dev)
        gamma = torch.randn(C, device=dev); beta = torch.randn(C, device=dev)
        mr = _ops.gn_stats(y)
        out = _ops.gn_apply(y, mr, gamma, beta, None, True)
        for relu, res in ((True, False), (True, True)):
            o = out if res else None
            whole = timed(lambda: _ops.gn_backward(dout, o, y, mr, gamma, beta, relu, res))
            def per_sample():
                for n in range(N):
                    _ops.gn_backward(dout[n:n + 1], None if o is None else o[n:n + 1], y[n:n + 1], mr[n:n + 1], gamma, beta, relu, res)
            ps = timed(per_sample)
            def per_pair():
                for n in range(0, N, 2):
                    _ops.gn_backward(dout[n:n + 2], None if o is None else o[n:n + 2], y[n:n + 2], mr[n:n + 2], gamma, beta, relu, res)
            pp = timed(per_pair)
            gb = y.numel() * 4 * ((3 if not res else 5) + (2 if not res else 3)) / 1e9
            print('N={} {}^3 C={} residual={}: whole batch {:7.1f} us | sample by sample {:7.1f} us | two by two {:7.1f} us   ({:.2f} GB algorithmic)'.format(
                N, D, C, res, whole, ps, pp, gb))

main()
