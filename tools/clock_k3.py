"""shader clock / power while one 3x3x3 kernel form runs back to back for a few seconds (tools/clock_watch.ClockWatch)
usage: python tools/clock_k3.py [direct|wino|wino2d|wgrad|wgrad_wino|wgrad_wino2d] [seconds]"""
import os, sys, time, json, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'medical-segmentation3d-toolkit_amd')); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tools'))
from segmentation3d import _engine as E
from clock_watch import ClockWatch
form = sys.argv[1] if len(sys.argv) > 1 else 'wino2d'
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
N, D, H, W, C = 8, 96, 96, 96, 32
dev = torch.device('cuda:0')
x = torch.randn(N, D, H, W, C, device=dev); w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.05; b = torch.zeros(C, device=dev)
y = torch.empty(N, D, H, W, C, device=dev); dw = torch.empty(C, C, 3, 3, 3, device=dev)
fl = 2.0 * N * D * H * W * 27 * C * C
if form in ('direct', 'wino', 'wino2d'):
    T = {'direct': 27, 'wino': 36, 'wino2d': 48}[form]
    wp = torch.empty(E.query('seg3d_packed_mfma_floats', C, C, T), device=dev)
    E.call('seg3d_pack_weights_mfma', E.ptr(w), E.ptr(wp), C, C, T, 27, C * 27, 0, E.stream_ptr())
    if form == 'direct':
        ws = torch.empty(max(1, E.query('seg3d_conv3d_k3_mfma_fwd_workspace_floats', N, D, H, W, C, C)), device=dev)
        fn = lambda: E.call('seg3d_conv3d_k3_mfma_fwd', E.ptr(x), E.ptr(wp), E.ptr(b), None, E.ptr(y), None, E.ptr(ws), N, D, H, W, C, C, E.stream_ptr())
    else:
        fn = lambda: E.call('seg3d_conv3d_k3_{}_fwd'.format(form), E.ptr(x), E.ptr(wp), E.ptr(b), None, E.ptr(y), None, N, D, H, W, C, C, E.stream_ptr())
else:
    name = {'wgrad': 'mfma', 'wgrad_wino': 'wino', 'wgrad_wino2d': 'wino2d'}[form]
    ws = torch.empty(E.query('seg3d_conv3d_k3_{}_wgrad_workspace_floats'.format(name), N, D, H, W, C, C), device=dev)
    fn = lambda: E.call('seg3d_conv3d_k3_{}_wgrad'.format(name), E.ptr(x), E.ptr(y.normal_() if False else y), E.ptr(dw), E.ptr(ws), N, D, H, W, C, C, 0, E.stream_ptr())
for _ in range(3): fn()
torch.cuda.synchronize()
n = 0
with ClockWatch(0.01) as cw:
    t0 = time.time()
    while time.time() - t0 < secs:
        for _ in range(20): fn()
        torch.cuda.synchronize(); n += 20
    dt = time.time() - t0
s = cw.summary()
print(json.dumps({'form': form, 'ms': 1e3 * dt / n, 'tflops_algorithmic': fl * n / dt / 1e12, 'clock': {k: v for k, v in s.items() if k not in ('files',)}}))
