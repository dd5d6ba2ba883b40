"""does an HBM-bound kernel share the chip with a persistent MFMA kernel of another stream?  times K launches of one 3x3x3
kernel form on stream A, K elementwise passes (GroupNorm-apply-like: 1 read + 1 write of a 453 MB tensor) on stream B, and
both at once.  usage: python tools/overlap_probe.py [wgrad_wino2d|wgrad_wino|wino2d|direct]"""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'medical-segmentation3d-toolkit_amd')); sys.path.insert(0, REPO)
from segmentation3d import _ops, _engine as E
form = sys.argv[1] if len(sys.argv) > 1 else 'wgrad_wino2d'
N, D, H, W, C = 4, 96, 96, 96, 32
dev = torch.device('cuda:0')
x = torch.randn(N, D, H, W, C, device=dev); dy = torch.randn(N, D, H, W, C, device=dev)
w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.05; b = torch.zeros(C, device=dev); y = torch.empty_like(x); dw = torch.empty_like(w)
a1 = torch.randn(N, D, H, W, C, device=dev); a2 = torch.empty_like(a1)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def mfma(stream):
    with torch.cuda.stream(stream):
        if form.startswith('wgrad'):
            name = {'wgrad_wino2d': 'wino2d', 'wgrad_wino': 'wino', 'wgrad': 'mfma'}[form]
            ws = mfma.ws.setdefault(name, torch.empty(E.query('seg3d_conv3d_k3_{}_wgrad_workspace_floats'.format(name), N, D, H, W, C, C), device=dev))
            E.call('seg3d_conv3d_k3_{}_wgrad'.format(name), E.ptr(x), E.ptr(dy), E.ptr(dw), E.ptr(ws), N, D, H, W, C, C, 0, E.stream_ptr())
        else:
            T = {'wino2d': 48, 'direct': 27}[form]
            if T not in mfma.ws:
                wp = torch.empty(E.query('seg3d_packed_mfma_floats', C, C, T), device=dev)
                E.call('seg3d_pack_weights_mfma', E.ptr(w), E.ptr(wp), C, C, T, 27, C * 27, 0, E.stream_ptr())
                mfma.ws[T] = wp
            if form == 'wino2d':
                E.call('seg3d_conv3d_k3_wino2d_fwd', E.ptr(x), E.ptr(mfma.ws[T]), E.ptr(b), None, E.ptr(y), None, N, D, H, W, C, C, E.stream_ptr())
            else:
                E.call('seg3d_conv3d_k3_mfma_fwd', E.ptr(x), E.ptr(mfma.ws[T]), E.ptr(b), None, E.ptr(y), None, None, N, D, H, W, C, C, E.stream_ptr())


mfma.ws = {}


def ew(stream):
    with torch.cuda.stream(stream):
        torch.mul(a1, 1.0001, out=a2)


def timed(fa, fb, k):
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(k):
        if fa: fa(sa)
        if fb:
            for _ in range(8): fb(sb)
    torch.cuda.synchronize()
    return 1e3 * (time.time() - t0) / k


for _ in range(2):
    mfma(sa); ew(sb)
K = 20
ta, tb, tab = timed(mfma, None, K), timed(None, ew, K), timed(mfma, ew, K)
print('{}: MFMA kernel alone {:.3f} ms, 8 elementwise passes alone {:.3f} ms, both streams {:.3f} ms (serial sum {:.3f})'.format(form, ta, tb, tab, ta + tb))


def chain(side_on, L=8):
    """the backward pattern: per layer a data-gradient (main), its weight gradient (side stream, after the data-gradient),
    then three elementwise passes (main) that the next data-gradient depends on"""
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(L):
        with torch.cuda.stream(sa):
            E.call('seg3d_conv3d_k3_wino2d_fwd', E.ptr(x), E.ptr(mfma.ws[48]), E.ptr(b), None, E.ptr(y), None, N, D, H, W, C, C, E.stream_ptr())
        st = sb if side_on else sa
        if side_on:
            sb.wait_stream(sa)
        with torch.cuda.stream(st):
            ws = mfma.ws['wino2d']
            E.call('seg3d_conv3d_k3_wino2d_wgrad', E.ptr(x), E.ptr(dy), E.ptr(dw), E.ptr(ws), N, D, H, W, C, C, 0, E.stream_ptr())
        with torch.cuda.stream(sa):
            for _ in range(3):
                torch.mul(a1, 1.0001, out=a2)
    sa.wait_stream(sb)
    torch.cuda.synchronize()
    return 1e3 * (time.time() - t0) / L


if form == 'chain':
    pass
form_saved = form
form = 'wino2d'; mfma(sa)
form = 'wgrad_wino2d'; mfma(sa)
form = form_saved
torch.cuda.synchronize()
for _ in range(2):
    chain(True); chain(False)
print('backward pattern per layer: one stream {:.3f} ms, weight gradient on a side stream {:.3f} ms'.format(chain(False), chain(True)))
