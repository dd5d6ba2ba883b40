"""same-box A/B of the F(2x2, 3x3) tile kernel with and without its stream-K workspace (seg3d_conv3d_k3_wino2d_fwd_ws), for the
shapes the workspace query says it would be used for (and any shape given on the command line)
usage: python tools/bench_stream_k.py [N D H W Cin Cout] [iters]"""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'medical-segmentation3d-toolkit_amd')); sys.path.insert(0, REPO)
from segmentation3d import _ops, _engine as E

def timed(fn, iters):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    c.record(); torch.cuda.synchronize()
    return a.elapsed_time(c) / iters

def main():
    shapes = [tuple(int(v) for v in sys.argv[1:7])] if len(sys.argv) >= 7 else [
        (4, 24, 24, 24, 128, 128), (8, 24, 24, 24, 128, 128), (4, 48, 48, 48, 32, 32), (16, 24, 24, 24, 128, 128), (4, 24, 24, 24, 64, 128),
        (4, 24, 24, 24, 256, 128), (4, 48, 48, 48, 64, 64), (8, 48, 48, 48, 32, 32), (16, 48, 48, 48, 64, 64)]
    iters = int(sys.argv[7]) if len(sys.argv) > 7 else 20
    dev = torch.device('cuda:0')
    for N, D, H, W, A, B in shapes:
        x = torch.randn(N, D, H, W, A, device=dev)
        w = torch.randn(B, A, 3, 3, 3, device=dev) * 0.05
        b = torch.zeros(B, device=dev)
        y = torch.empty(N, D, H, W, B, device=dev)
        fl = 2.0 * N * D * H * W * 27 * A * B
        wq = torch.empty(E.query('seg3d_packed_mfma_floats', A, B, 48), device=dev)
        E.call('seg3d_pack_weights_mfma', E.ptr(w), E.ptr(wq), A, B, 48, 27, A * 27, 0, E.stream_ptr())
        st = torch.empty(N, E.query('seg3d_conv3d_k3_wino2d_stats_count', N, D, H, W, A, B), 2, device=dev)
        nws = E.query('seg3d_conv3d_k3_wino2d_fwd_workspace_floats', N, D, H, W, A, B)
        items = N * (D // 8) * (H // 8) * (W // 8) * (B // 32)
        line = 'N={} {}x{}x{} {}->{} ({} items x {} chunks): '.format(N, D, H, W, A, B, items, A // 4)
        res = []
        for _ in range(2):   # interleaved twice: clock drift shows as a difference between the repeats
            for ws in ([None, torch.empty(nws, device=dev)] if nws else [None]):
                ms = timed(lambda: E.call('seg3d_conv3d_k3_wino2d_fwd_ws', E.ptr(x), E.ptr(wq), E.ptr(b), None, E.ptr(y), E.ptr(st), E.ptr(ws), N, D, H, W, A, B, E.stream_ptr()), iters)
                res.append(ms)
        if nws:
            line += 'whole items {:.4f} / {:.4f} ms, stream-K {:.4f} / {:.4f} ms  ({:+.1f} %), {:.1f} TF algorithmic'.format(res[0], res[2], res[1], res[3], 100.0 * ((res[1] + res[3]) / (res[0] + res[2]) - 1.0), fl / min(res[1], res[3]) / 1e9)
        else:
            line += 'whole items {:.4f} / {:.4f} ms (no stream-K for this shape)'.format(res[0], res[1])
        print(line, flush=True)

main()
