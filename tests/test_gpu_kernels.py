"""GPU parity tests of the individual HIP operators against stock torch CPU ops (the operators the reference composes)
and the oracle's loss closed forms.  Every call goes through the C ABI of libseg3d_hip.so.
Tolerances: fp32 kernels, 1e-4 absolute on O(1) values (north_star), relative 1e-3..1e-4 on gradients."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import report, max_err, rel_err
from oracle import detgen, torch_ref, numpy_ref

pytestmark = pytest.mark.gpu


def _t(seed, name, shape, std=1.0):
    return torch.from_numpy(detgen.normal(seed, name, shape, std=std))


CONV_K3_CASES = [
    # N, Cin, Cout, D, H, W
    (2, 8, 32, 8, 8, 16),
    (1, 32, 32, 16, 16, 16),
    (1, 16, 16, 6, 10, 12),
    (1, 64, 96, 4, 6, 6),
    (1, 12, 20, 5, 7, 9),
    (2, 256, 64, 2, 2, 2),
    (2, 1, 16, 8, 8, 8),
    (1, 32, 2, 8, 8, 8),
    (1, 3, 5, 6, 6, 6),
    (1, 128, 128, 3, 5, 6),
    (1, 4, 16, 5, 9, 11),     # thin-in (stem, 4 modalities), masked tiles
    (2, 32, 5, 6, 10, 18),    # thin-out (head, 5 classes), masked tiles
    (1, 16, 4, 4, 8, 16),     # thin-out, 4 outputs
    (1, 8, 16, 4, 8, 8),      # thin-in with 8 channels / generic MFMA boundary
    (4, 1, 16, 16, 16, 16),
    (1, 16, 16, 12, 12, 12),  # weight-gradient tile 4x4x4 (extent divisible by 4, not by 8)
    (2, 32, 64, 6, 6, 6),     # weight-gradient tile 2x6x6; split-K forward
    (1, 24, 40, 4, 12, 20),   # partial 32-channel blocks on both sides
    (1, 16, 64, 8, 8, 16),    # two output-channel blocks
    (1, 24, 6, 5, 9, 11),     # thin-out shapes outside the fp32-MFMA head kernel: LDS-tiled VALU kernel
    (1, 32, 7, 4, 8, 16),
]


@pytest.mark.parametrize('force_direct', [False, True])
@pytest.mark.parametrize('case', CONV_K3_CASES)
def test_conv3d_k3_fwd_bwd(hip_device, case, force_direct):
    from segmentation3d import _ops
    N, Cin, Cout, D, H, W = case
    if force_direct and Cin * Cout > 64 * 96:
        pytest.skip('direct fallback not exercised on the largest shapes')
    name = 'conv_k3_{}_{}'.format('direct' if force_direct else 'auto', '_'.join(map(str, case)))
    x = _t(1, name + 'x', (N, Cin, D, H, W)).requires_grad_(True)
    w = _t(2, name + 'w', (Cout, Cin, 3, 3, 3), std=(2.0 / (Cin * 27)) ** 0.5).requires_grad_(True)
    b = _t(3, name + 'b', (Cout,), std=0.1).requires_grad_(True)
    g = _t(4, name + 'g', (N, Cout, D, H, W))
    ref = F.conv3d(x, w, b, padding=1)
    rdx, rdw, rdb = torch.autograd.grad(ref, (x, w, b), g)
    xd, wd, bd = (t.detach().to(hip_device).requires_grad_(True) for t in (x, w, b))
    _ops.FORCE_DIRECT = force_direct
    try:
        out = _ops.conv(xd, wd, bd, 'k3')
        dx, dw, db = torch.autograd.grad(out, (xd, wd, bd), g.to(hip_device))
    finally:
        _ops.FORCE_DIRECT = False
    torch.cuda.synchronize()
    e = dict(out=max_err(out, ref), dx=rel_err(dx, rdx), dw=rel_err(dw, rdw), db=rel_err(db, rdb))
    report(name, **e)
    assert e['out'] < 1e-4 and e['dx'] < 1e-4 and e['dw'] < 2e-4 and e['db'] < 1e-4, e


# the kernel instantiations the headline bench actually launches (BASELINE config 2: vnet(1,2), 4 x 96^3): variant codes of
# seg3d_conv3d_k3_mfma_variant -- 321 = conv3d_k3_mfma2w8_kernel<2,1> (512-voxel tiles, 8 waves; the kernel bench.py's
# `roofline` is quoted on), 131 = conv3d_k3_mfma2_kernel<3,1>, 311 = conv3d_k3_mfma2w8_kernel<1,1>
BENCH_VARIANT_CASES = [
    # (N, C, D, H, W), expected variant
    ((4, 64, 24, 24, 24), 321),      # up_128 / down_64 level at batch 4
    ((4, 128, 24, 24, 24), 321),     # up_128.rblock
    ((1, 64, 48, 48, 48), 321),
    ((1, 32, 96, 96, 96), 131),      # the 96^3 level of one patch
    ((4, 64, 48, 48, 48), 131),      # up_64.rblock
    ((4, 32, 48, 48, 48), 311),      # down_32.rblock
    ((1, 128, 24, 24, 24), 311),
    ((4, 32, 96, 96, 96), 321),      # up_32.rblock at batch 4: the largest launch of the step (most weight-gradient slabs)
]


@pytest.mark.parametrize('form', ['direct', 'wino', 'wino2d'])
@pytest.mark.parametrize('shape,variant', BENCH_VARIANT_CASES)
def test_conv3d_k3_bench_variants_full_size(hip_device, shape, variant, form, monkeypatch):
    """forward, data-gradient (fused addend included) and weight-gradient of the C -> C 3x3x3 convolution at the FULL sizes
    of the headline configuration against stock torch CPU ops, asserting which kernels ran: the direct implicit-GEMM
    instantiations (variant codes 321 / 131 / 311), the Winograd F(2, 3) / F(3, 2) kernels, and the F(2x2, 3x3) / F(3x3, 2x2)
    kernels the default path uses at these levels (the entry points called through the C ABI are recorded and checked)"""
    from segmentation3d import _ops, _engine as E
    N, C, D, H, W = shape
    assert E.query('seg3d_conv3d_k3_mfma_variant', N, D, H, W, C, C) == variant
    if form != 'direct' and not E.query('seg3d_conv3d_k3_{}_preferred'.format(form), N, D, H, W, C, C):
        pytest.skip('fewer than 192 items: the default path keeps the direct kernel here')
    monkeypatch.setattr(_ops, 'WINOGRAD', form != 'direct')
    monkeypatch.setattr(_ops, 'WINOGRAD2D', form == 'wino2d')
    called = []
    orig_call = E.call

    def spy(fn, *a):
        called.append(fn)
        return orig_call(fn, *a)
    monkeypatch.setattr(E, 'call', spy)
    winograd = form != 'direct'
    # (the direct kernel's variant code names the direct runs only; a Winograd run is labelled by its form)
    name = 'bench_{}_{}'.format('direct{}'.format(variant) if form == 'direct' else form, '_'.join(map(str, shape)))
    torch.set_num_threads(max(torch.get_num_threads(), 16))
    x = _t(15, name + 'x', (N, C, D, H, W)).requires_grad_(True)
    w = _t(16, name + 'w', (C, C, 3, 3, 3), std=(2.0 / (C * 27)) ** 0.5).requires_grad_(True)
    b = _t(17, name + 'b', (C,), std=0.1)
    g = _t(18, name + 'g', (N, C, D, H, W))
    ad = _t(19, name + 'a', (N, C, D, H, W))
    ref = F.conv3d(x, w, b, padding=1)
    rdx, rdw = torch.autograd.grad(ref, (x, w), g)
    xn = _ops.to_ndhwc(x.detach().to(hip_device))
    gn = _ops.to_ndhwc(g.to(hip_device))
    an = _ops.to_ndhwc(ad.to(hip_device))
    wd, bd = w.detach().to(hip_device), b.to(hip_device)
    y, part = _ops.conv_forward(xn, wd, bd, 'k3', want_stats=True)
    dx = _ops.conv_dgrad(gn, wd, 'k3', addend=an)
    dw = _ops.conv_wgrad(xn, gn, (C, C, 3, 3, 3), 'k3')
    base = _t(20, name + 'base', (C, C, 3, 3, 3)).to(hip_device)
    acc = base.clone()
    _ops.conv_wgrad(xn, gn, (C, C, 3, 3, 3), 'k3', out=acc)          # gradient-sink form (accumulate flag)
    torch.cuda.synchronize()
    rr = ref.detach().double().reshape(N, -1)
    st = part.double().sum(1).cpu()
    e = dict(out=max_err(_ops.from_ndhwc(y), ref), dx=rel_err(_ops.from_ndhwc(dx), rdx + ad), dw=rel_err(dw, rdw),
             dw_acc=rel_err(acc - base, rdw), stat_sum=float(((st[:, 0] - rr.sum(1)).abs() / rr.abs().sum(1)).max()),
             stat_sq=rel_err(st[:, 1], (rr * rr).sum(1)), variant=float(variant))
    report(name, **e)
    assert e['out'] < 1e-4 and e['dx'] < 1e-4 and e['dw'] < 2e-4 and e['dw_acc'] < 2e-4, e
    assert e['stat_sum'] < 1e-5 and e['stat_sq'] < 1e-5, e
    fwd_name = {'direct': 'seg3d_conv3d_k3_mfma_fwd', 'wino': 'seg3d_conv3d_k3_wino_fwd', 'wino2d': 'seg3d_conv3d_k3_wino2d_fwd_ws'}[form]
    assert called.count(fwd_name) == 2, called                        # forward + data-gradient
    wg = [c for c in called if c.endswith('_wgrad')]
    if form == 'direct':
        assert wg == ['seg3d_conv3d_k3_mfma_wgrad'] * 2, wg
    elif form == 'wino':
        assert wg == ['seg3d_conv3d_k3_wino_wgrad'] * 2, wg
    else:   # F(3x3, 2x2) where a workgroup gets enough tiles, else F(3, 2)
        want = 'seg3d_conv3d_k3_wino2d_wgrad' if E.query('seg3d_conv3d_k3_wino2d_wgrad_preferred', N, D, H, W, C, C) else 'seg3d_conv3d_k3_wino_wgrad'
        assert wg == [want] * 2, wg


@pytest.mark.parametrize('shape', [(1, 8, 32, 16, 24, 32), (2, 32, 32, 8, 16, 48), (1, 64, 96, 16, 16, 16), (3, 16, 64, 8, 8, 64),
                                   (1, 128, 128, 8, 16, 16),
                                   # whole 4^3 cells but not 8^3 tiles: the cell form of the F(2x2, 3x3) kernel (wino2d only) --
                                   # the 12^3 level, a last item with three of four cells, cells of one item in two samples
                                   # (two cells per item below 192 items; the 256-channel case has 216 items of four cells)
                                   (2, 16, 64, 12, 12, 12), (1, 8, 32, 4, 4, 12), (3, 24, 32, 4, 12, 8), (1, 64, 64, 12, 12, 20),
                                   (4, 8, 256, 12, 12, 12)])
@pytest.mark.parametrize('flip', [0, 1])
@pytest.mark.parametrize('form', ['wino', 'wino2d'])
def test_conv3d_k3_winograd(hip_device, shape, flip, form):
    """Winograd F(2, 3) along x (csrc/conv_wino.hip, T = 36 image) and F(2x2, 3x3) over (y, x) (csrc/conv_wino2d.hip, T = 48
    image) through the C ABI against the float64 convolution: several K chunks and
    column blocks, more items than workgroups and fewer, flipped taps (the data-gradient form), fused addend, no-bias /
    no-stats calls, per-wave statistics of its own output; and against the direct MFMA kernel on the same input"""
    from segmentation3d import _ops, _engine as E
    N, Cin, Cout, D, H, W = shape
    T = 36 if form == 'wino' else 48
    cells = bool(D % 8 or H % 8 or W % 8)
    if cells and form == 'wino':
        pytest.skip('F(2, 3) along x takes whole 8^3 tiles only')
    assert E.query('seg3d_conv3d_k3_{}_supported'.format(form), N, D, H, W, Cin, Cout) == 1
    assert E.query('seg3d_conv3d_k3_{}_supported'.format(form), N, D, H, W + (4 if form == 'wino' else 2), Cin, Cout) == 0   # not whole tiles / cells
    assert E.query('seg3d_conv3d_k3_{}_supported'.format(form), N, D, H, W, Cin + 4, Cout) == 0
    assert E.query('seg3d_conv3d_k3_{}_preferred'.format(form), 1, 8, 8, 8, Cin, Cout) == 0           # too few items: split-K kernel
    x = _t(31, 'wx', (N, Cin, D, H, W))
    w = _t(32, 'ww', (Cout, Cin, 3, 3, 3), std=(2.0 / (Cin * 27)) ** 0.5)
    b = _t(33, 'wb', (Cout,), std=0.1)
    ad = _t(34, 'wa', (N, Cout, D, H, W))
    xn = _ops.to_ndhwc(x.to(hip_device))
    an = _ops.to_ndhwc(ad.to(hip_device))
    wd, bd = w.to(hip_device), b.to(hip_device)
    wp = torch.full((E.query('seg3d_packed_mfma_floats', Cin, Cout, T),), float('nan'), device=hip_device)
    E.call('seg3d_pack_weights_mfma', E.ptr(wd), E.ptr(wp), Cin, Cout, T, 27, Cin * 27, flip, E.stream_ptr())
    y = torch.full((N, D, H, W, Cout), float('nan'), device=hip_device)
    st = torch.full((N, E.query('seg3d_conv3d_k3_{}_stats_count'.format(form), N, D, H, W, Cin, Cout), 2), float('nan'), device=hip_device)
    E.call('seg3d_conv3d_k3_{}_fwd'.format(form), E.ptr(xn), E.ptr(wp), E.ptr(bd), E.ptr(an), E.ptr(y), E.ptr(st), N, D, H, W, Cin, Cout,
           E.stream_ptr())
    wref = w.flip(2, 3, 4) if flip else w
    ref = F.conv3d(x.double(), wref.double(), b.double(), padding=1) + ad.double()
    got = _ops.from_ndhwc(y).double().cpu()
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max())
    # direct MFMA kernel on the same input: the yardstick for "error stays at the direct kernel's level"
    wp27 = torch.empty((E.query('seg3d_packed_mfma_floats', Cin, Cout, 27),), device=hip_device)
    E.call('seg3d_pack_weights_mfma', E.ptr(wd), E.ptr(wp27), Cin, Cout, 27, 27, Cin * 27, flip, E.stream_ptr())
    yd = torch.empty((N, D, H, W, Cout), device=hip_device)
    nws = E.query('seg3d_conv3d_k3_mfma_fwd_workspace_floats', N, D, H, W, Cin, Cout)
    ws = torch.empty(max(nws, 1), device=hip_device)
    E.call('seg3d_conv3d_k3_mfma_fwd', E.ptr(xn), E.ptr(wp27), E.ptr(bd), E.ptr(an), E.ptr(yd), None, E.ptr(ws), N, D, H, W, Cin,
           Cout, E.stream_ptr())
    err_direct = float((_ops.from_ndhwc(yd).double().cpu() - ref).abs().max())
    report('{}_{}_flip{}'.format(form, '_'.join(map(str, shape)), flip), max_abs_err=err, direct_kernel_err=err_direct,
           out_scale=scale)
    assert err < 1e-5 * scale and err < 4.0 * err_direct + 1e-6 * scale, (err, err_direct, scale)
    s = st.double().sum(1).cpu()
    rr = got.reshape(N, -1)
    assert float(((s[:, 0] - rr.sum(1)).abs() / rr.abs().sum(1)).max()) < 1e-5 and rel_err(s[:, 1], (rr * rr).sum(1)) < 1e-5
    y2 = torch.full((N, D, H, W, Cout), float('nan'), device=hip_device)
    E.call('seg3d_conv3d_k3_{}_fwd'.format(form), E.ptr(xn), E.ptr(wp), None, None, E.ptr(y2), None, N, D, H, W, Cin, Cout, E.stream_ptr())
    assert float((y2 + bd + an - y).abs().max()) < 2e-6 * scale
    # bias only / addend only (the other two instantiations), and run to run bitwise the same
    y3 = torch.full((N, D, H, W, Cout), float('nan'), device=hip_device)
    E.call('seg3d_conv3d_k3_{}_fwd'.format(form), E.ptr(xn), E.ptr(wp), E.ptr(bd), None, E.ptr(y3), None, N, D, H, W, Cin, Cout, E.stream_ptr())
    assert float((y3 + an - y).abs().max()) < 2e-6 * scale
    y4 = torch.full((N, D, H, W, Cout), float('nan'), device=hip_device)
    E.call('seg3d_conv3d_k3_{}_fwd'.format(form), E.ptr(xn), E.ptr(wp), None, E.ptr(an), E.ptr(y4), None, N, D, H, W, Cin, Cout, E.stream_ptr())
    assert float((y4 + bd - y).abs().max()) < 2e-6 * scale
    y5 = torch.full((N, D, H, W, Cout), float('nan'), device=hip_device)
    E.call('seg3d_conv3d_k3_{}_fwd'.format(form), E.ptr(xn), E.ptr(wp), E.ptr(bd), E.ptr(an), E.ptr(y5), None, N, D, H, W, Cin, Cout, E.stream_ptr())
    assert torch.equal(y5, y)


@pytest.mark.parametrize('shape', [(4, 128, 128, 24, 24, 24),    # 432 items of 32 chunks on 256 CUs: the 24^3 level of the train step
                                   (4, 32, 32, 48, 48, 48),       # 864 items of 8 chunks: ranges of 27 chunks, three or four items each
                                   (3, 32, 64, 32, 40, 24),       # 360 items: ranges of 12 chunks, the last workgroups empty
                                   (1, 64, 32, 64, 48, 48)])      # 288 items of 16 chunks, one column block
def test_conv3d_k3_wino2d_stream_k(hip_device, shape):
    """stream-K form of the F(2x2, 3x3) tile kernel (seg3d_conv3d_k3_wino2d_fwd_ws with its workspace: contiguous chunk ranges per
    workgroup, cut items finished by conv3d_k3_wino2d_sk_finish_kernel) against the float64 convolution, against the same entry
    point without a workspace (whole items per workgroup), the statistics slots of cut and uncut items, all four bias / addend
    instantiations, and run to run bitwise the same"""
    from segmentation3d import _ops, _engine as E
    N, Cin, Cout, D, H, W = shape
    nws = E.query('seg3d_conv3d_k3_wino2d_fwd_workspace_floats', N, D, H, W, Cin, Cout)
    assert nws > 0, 'shape was chosen to take the stream-K path'
    assert E.query('seg3d_conv3d_k3_wino2d_fwd_workspace_floats', 8, 96, 96, 96, 32, 32) == 0    # whole rounds: no stream-K
    assert E.query('seg3d_conv3d_k3_wino2d_fwd_workspace_floats', 4, 12, 12, 12, 128, 128) == 0   # cells
    x = _t(35, 'skx', (N, Cin, D, H, W))
    w = _t(36, 'skw', (Cout, Cin, 3, 3, 3), std=(2.0 / (Cin * 27)) ** 0.5)
    b = _t(37, 'skb', (Cout,), std=0.1)
    ad = _t(38, 'ska', (N, Cout, D, H, W))
    xn = _ops.to_ndhwc(x.to(hip_device))
    an = _ops.to_ndhwc(ad.to(hip_device))
    wd, bd = w.to(hip_device), b.to(hip_device)
    wp = torch.empty((E.query('seg3d_packed_mfma_floats', Cin, Cout, 48),), device=hip_device)
    E.call('seg3d_pack_weights_mfma', E.ptr(wd), E.ptr(wp), Cin, Cout, 48, 27, Cin * 27, 0, E.stream_ptr())
    nst = E.query('seg3d_conv3d_k3_wino2d_stats_count', N, D, H, W, Cin, Cout)

    def run(bias, addend, stats, workspace):
        y = torch.full((N, D, H, W, Cout), float('nan'), device=hip_device)
        st = torch.full((N, nst, 2), float('nan'), device=hip_device) if stats else None
        ws = torch.full((nws,), float('nan'), device=hip_device) if workspace else None
        E.call('seg3d_conv3d_k3_wino2d_fwd_ws', E.ptr(xn), E.ptr(wp), E.ptr(bd) if bias else None, E.ptr(an) if addend else None,
               E.ptr(y), E.ptr(st), E.ptr(ws), N, D, H, W, Cin, Cout, E.stream_ptr())
        return y, st

    y, st = run(True, True, True, True)
    y0, st0 = run(True, True, True, False)
    ref = F.conv3d(x.double(), w.double(), b.double(), padding=1) + ad.double()
    got = _ops.from_ndhwc(y).double().cpu()
    scale = float(ref.abs().max())
    err, err0 = float((got - ref).abs().max()), float((_ops.from_ndhwc(y0).double().cpu() - ref).abs().max())
    report('wino2d_stream_k_{}'.format('_'.join(map(str, shape))), max_abs_err=err, whole_items_err=err0, out_scale=scale,
           items_differ=float((y != y0).reshape(N, -1).any(1).sum()))
    assert err < 1e-5 * scale and err < 2.0 * err0 + 1e-6 * scale, (err, err0, scale)
    assert not torch.equal(y, y0), 'a cut item is summed in two pieces: the stream-K path did not run'
    s = st.double().sum(1).cpu()
    rr = got.reshape(N, -1)
    assert float(((s[:, 0] - rr.sum(1)).abs() / rr.abs().sum(1)).max()) < 1e-5 and rel_err(s[:, 1], (rr * rr).sum(1)) < 1e-5
    assert torch.isfinite(st).all()   # every slot of every item written exactly by one of the two kernels
    assert rel_err(st.double().sum(1), st0.double().sum(1)) < 1e-5
    for bias, addend in ((False, False), (True, False), (False, True)):
        y2, _ = run(bias, addend, False, True)
        full = y2 + (0 if bias else bd) + (0 if addend else an)
        assert float((full - y).abs().max()) < 2e-6 * scale, (bias, addend)
    y5, st5 = run(True, True, True, True)
    assert torch.equal(y5, y) and torch.equal(st5, st)


@pytest.mark.parametrize('shape', [(1, 32, 32, 8, 8, 16), (2, 16, 48, 4, 8, 8), (1, 64, 32, 12, 12, 12), (3, 8, 40, 8, 4, 24),
                                   (1, 32, 32, 16, 48, 32), (2, 96, 64, 4, 4, 4),
                                   # columns of four tiles and more: the F(3x3, 2x2) kernel with the transform inside the MFMA loop
                                   # (phantom steps at column starts, workgroup ranges that start inside a column, partial blocks)
                                   (2, 48, 40, 24, 8, 12), (1, 32, 64, 48, 16, 8), (3, 16, 32, 16, 12, 20)])
@pytest.mark.parametrize('form', ['wino', 'wino2d'])
def test_conv3d_k3_winograd_wgrad(hip_device, shape, form):
    """Winograd F(3, 2) along x (csrc/conv_wino.hip) and F(3x3, 2x2) over (y, x) (csrc/conv_wino2d.hip) weight gradients through the C ABI against the float64 weight gradient: both
    tile shapes (W % 8 == 0 and W % 4 == 0), partial 32-channel blocks, one and several tiles per workgroup, several block
    pairs, batch > 1; the accumulate flag; and against the 27-tap MFMA kernel on the same input"""
    from segmentation3d import _ops, _engine as E
    N, Cin, Cout, D, H, W = shape
    assert E.query('seg3d_conv3d_k3_{}_wgrad_supported'.format(form), N, D, H, W, Cin, Cout) == 1
    assert E.query('seg3d_conv3d_k3_{}_wgrad_supported'.format(form), N, D, H, W + 2, Cin, Cout) == 0     # not whole tiles
    assert E.query('seg3d_conv3d_k3_{}_wgrad_supported'.format(form), N, D, H, W, Cin + 2, Cout) == 0
    x = _t(41, 'gx', (N, Cin, D, H, W))
    dy = _t(42, 'gdy', (N, Cout, D, H, W))
    xd = x.double().requires_grad_(False)
    wz = torch.zeros((Cout, Cin, 3, 3, 3), dtype=torch.float64, requires_grad=True)
    F.conv3d(xd, wz, None, padding=1).backward(dy.double())
    ref = wz.grad
    xn, dyn = _ops.to_ndhwc(x.to(hip_device)), _ops.to_ndhwc(dy.to(hip_device))
    ws = torch.full((E.query('seg3d_conv3d_k3_{}_wgrad_workspace_floats'.format(form), N, D, H, W, Cin, Cout),), float('nan'), device=hip_device)
    dw = torch.full((Cout, Cin, 3, 3, 3), float('nan'), device=hip_device)
    E.call('seg3d_conv3d_k3_{}_wgrad'.format(form), E.ptr(xn), E.ptr(dyn), E.ptr(dw), E.ptr(ws), N, D, H, W, Cin, Cout, 0, E.stream_ptr())
    scale = float(ref.abs().max())
    err = float((dw.double().cpu() - ref).abs().max())
    ws27 = torch.empty((E.query('seg3d_conv3d_k3_mfma_wgrad_workspace_floats', N, D, H, W, Cin, Cout),), device=hip_device)
    dw27 = torch.empty((Cout, Cin, 3, 3, 3), device=hip_device)
    E.call('seg3d_conv3d_k3_mfma_wgrad', E.ptr(xn), E.ptr(dyn), E.ptr(dw27), E.ptr(ws27), N, D, H, W, Cin, Cout, 0, E.stream_ptr())
    err_direct = float((dw27.double().cpu() - ref).abs().max())
    report('{}_wgrad_{}'.format(form, '_'.join(map(str, shape))), max_abs_err=err, direct_kernel_err=err_direct, out_scale=scale)
    assert err < 1e-5 * scale and err < 4.0 * err_direct + 1e-6 * scale, (err, err_direct, scale)
    # accumulate: dw += gradient, bitwise the same gradient as the plain call
    base = _t(43, 'gbase', (Cout, Cin, 3, 3, 3)).to(hip_device)
    acc = base.clone()
    E.call('seg3d_conv3d_k3_{}_wgrad'.format(form), E.ptr(xn), E.ptr(dyn), E.ptr(acc), E.ptr(ws), N, D, H, W, Cin, Cout, 1, E.stream_ptr())
    assert torch.equal(acc, base + dw)
    # run to run: bitwise reproducible
    dw2 = torch.empty_like(dw)
    E.call('seg3d_conv3d_k3_{}_wgrad'.format(form), E.ptr(xn), E.ptr(dyn), E.ptr(dw2), E.ptr(ws), N, D, H, W, Cin, Cout, 0, E.stream_ptr())
    assert torch.equal(dw, dw2)


def test_conv_k3_mfma_addend_and_accumulate_flags(hip_device):
    """C-ABI flags of the second-generation kernels: `addend` of seg3d_conv3d_k3_mfma_fwd (y = conv + bias + addend, the
    fused residual-path gradient) and `accumulate` of seg3d_conv3d_k3_mfma_wgrad (dw += ..., gradient sinks), checked
    against the same entry points without the flag"""
    from segmentation3d import _ops, _engine as E
    for (N, D, H, W, Cin, Cout) in [(2, 8, 12, 16, 32, 32), (1, 6, 6, 6, 16, 24), (1, 12, 12, 12, 64, 64)]:
        x = _t(5, 'flagx', (N, D, H, W, Cin)).to(hip_device)
        dy = _t(6, 'flagdy', (N, D, H, W, Cout)).to(hip_device)
        ad = _t(7, 'flagad', (N, D, H, W, Cout)).to(hip_device)
        w = (_t(8, 'flagw', (Cout, Cin, 3, 3, 3)) * 0.05).to(hip_device)
        b = _t(9, 'flagb', (Cout,)).to(hip_device)
        y0, _ = _ops._conv_k3_generic(x, w, b, Cin, Cout, 27, Cin * 27, 0, False)
        y1, _ = _ops._conv_k3_generic(x, w, b, Cin, Cout, 27, Cin * 27, 0, False, addend=ad)
        assert max_err(y1, y0 + ad) < 1e-5
        dw0 = _ops.conv_wgrad(x, dy, (Cout, Cin, 3, 3, 3), 'k3')
        base = _t(10, 'flagbase', (Cout, Cin, 3, 3, 3)).to(hip_device)
        acc = base.clone()
        _ops.conv_wgrad(x, dy, (Cout, Cin, 3, 3, 3), 'k3', out=acc)
        assert rel_err(acc - base, dw0) < 1e-5
        report('flags_{}x{}x{}x{}_{}_{}'.format(N, D, H, W, Cin, Cout),
               variant=float(E.query('seg3d_conv3d_k3_mfma_variant', N, D, H, W, Cin, Cout)))


@pytest.mark.parametrize('shape', [(4, 256, 256, 6, 6, 6), (1, 256, 256, 12, 12, 12), (2, 128, 64, 5, 6, 7),
                                   (1, 64, 32, 3, 5, 6), (2, 256, 128, 8, 8, 8), (1, 72, 36, 4, 6, 6)])
def test_conv3d_k3_mfma_split_k(hip_device, shape):
    """spatially small levels cut K across work items (slabs summed by the finish pass): value, fused addend and
    statistics parity on whole and ragged tiles, K ranges that do not divide evenly included"""
    from segmentation3d import _ops, _engine as E
    N, Cin, Cout, D, H, W = shape
    x = _t(21, 'skx', (N, Cin, D, H, W)).to(hip_device)
    w = _t(22, 'skw', (Cout, Cin, 3, 3, 3), std=0.05).to(hip_device)
    b = _t(23, 'skb', (Cout,), std=0.5).to(hip_device)
    a = _t(24, 'ska', (N, Cout, D, H, W)).to(hip_device)
    an = _ops.to_ndhwc(a)
    y, part = _ops._conv_k3_generic(_ops.to_ndhwc(x), w, b, Cin, Cout, 27, Cin * 27, 0, True, addend=an)
    ref = F.conv3d(x.cpu().double(), w.cpu().double(), b.cpu().double(), padding=1) + a.cpu().double()
    assert max_err(_ops.from_ndhwc(y).double(), ref) < 1e-4
    s = part.double().sum(1).cpu()
    rr = ref.reshape(N, -1)
    assert float(((s[:, 0] - rr.sum(1)).abs() / rr.abs().sum(1)).max()) < 1e-5 and rel_err(s[:, 1], (rr * rr).sum(1)) < 1e-5
    nws = E.query('seg3d_conv3d_k3_mfma_fwd_workspace_floats', N, D, H, W, Cin, Cout)
    report('splitk_{}x{}x{}x{}_{}_{}'.format(N, D, H, W, Cin, Cout),
           variant=float(E.query('seg3d_conv3d_k3_mfma_variant', N, D, H, W, Cin, Cout)),
           ksplit=float(nws // (N * D * H * W * Cout)))


def test_bf16_conversion_bit_exact(hip_device):
    """seg3d_f32_to_bf16 rounds to nearest even exactly like torch's .bfloat16() (incl. ties, denormals, inf)"""
    from segmentation3d import _engine as E
    vals = _t(31, 'cv', (4099,)).float()
    special = torch.tensor([0.0, -0.0, 1.0, 1.00390625, 1.01171875, 3.3895314e38, float('inf'), -float('inf'), 1e-40, 65504.0])
    src = torch.cat([vals, special]).to(hip_device)
    pad = (-src.numel()) % 4
    dst = torch.empty(src.numel() + pad, dtype=torch.bfloat16, device=hip_device)
    E.call('seg3d_f32_to_bf16', E.ptr(src), E.ptr(dst), src.numel(), E.stream_ptr())
    got = dst[:src.numel()].view(torch.int16).cpu()
    want = src.cpu().bfloat16().view(torch.int16)
    assert torch.equal(got, want)
    back = torch.empty(src.numel(), device=hip_device)
    E.call('seg3d_bf16_to_f32', E.ptr(dst), E.ptr(back), src.numel(), E.stream_ptr())
    assert torch.equal(back.cpu(), src.cpu().bfloat16().float())


@pytest.mark.parametrize('shape', [(1, 32, 32, 16, 16, 16), (2, 16, 16, 6, 10, 12), (1, 64, 96, 4, 6, 6), (4, 256, 256, 6, 6, 6),
                                   (1, 128, 64, 5, 6, 7), (2, 48, 20, 4, 12, 20), (1, 32, 32, 24, 24, 48)])
@pytest.mark.parametrize('with_addend', [False, True])
def test_conv3d_k3_bf16_fwd(hip_device, shape, with_addend):
    """bf16-input conv (bf16 activations and weights, fp32 accumulate / bias / addend / output): equals the exact conv of
    the bf16-rounded operands to fp32 accumulation error (tolerance 2e-5 relative to the output scale), and its GN partial
    sums equal the statistics of its own output"""
    from segmentation3d import _ops, _engine as E
    N, Cin, Cout, D, H, W = shape
    x = _t(41, 'bx', (N, Cin, D, H, W))
    w = _t(42, 'bw', (Cout, Cin, 3, 3, 3), std=0.05)
    b = _t(43, 'bb', (Cout,), std=0.5)
    a = _t(44, 'ba', (N, Cout, D, H, W))
    xb = _ops.to_ndhwc(x.to(hip_device)).bfloat16()
    wd = w.to(hip_device)
    wp = torch.empty(E.query('seg3d_packed_mfma_bf16_elems', Cin, Cout, 27), dtype=torch.bfloat16, device=hip_device)
    E.call('seg3d_pack_weights_mfma_bf16', E.ptr(wd), E.ptr(wp), Cin, Cout, 27, 27, Cin * 27, 0, E.stream_ptr())
    y = torch.empty(N, D, H, W, Cout, device=hip_device)
    cnt = E.query('seg3d_conv3d_k3_bf16_stats_count', N, D, H, W, Cin, Cout)
    assert cnt > 0
    st = torch.zeros(N, cnt, 2, device=hip_device)
    nws = E.query('seg3d_conv3d_k3_bf16_fwd_workspace_floats', N, D, H, W, Cin, Cout)
    ws = torch.empty(max(nws, 1), device=hip_device)
    an = _ops.to_ndhwc(a.to(hip_device)) if with_addend else None
    bd = b.to(hip_device)
    E.call('seg3d_conv3d_k3_bf16_fwd', E.ptr(xb), E.ptr(wp), E.ptr(bd), E.ptr(an), E.ptr(y), E.ptr(st), E.ptr(ws),
           N, D, H, W, Cin, Cout, 0, E.stream_ptr())
    ref = F.conv3d(x.bfloat16().double(), w.bfloat16().double(), b.double(), padding=1)
    if with_addend:
        ref = ref + a.double()
    got = _ops.from_ndhwc(y).double().cpu()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) < 2e-5 * scale
    s = st.double().sum(1).cpu()
    rr = got.reshape(N, -1)
    assert float(((s[:, 0] - rr.sum(1)).abs() / rr.abs().sum(1)).max()) < 1e-5 and rel_err(s[:, 1], (rr * rr).sum(1)) < 1e-5
    report('bf16conv_{}x{}x{}x{}_{}_{}{}'.format(N, D, H, W, Cin, Cout, '_addend' if with_addend else ''),
           variant=float(E.query('seg3d_conv3d_k3_bf16_variant', N, D, H, W, Cin, Cout)),
           max_abs_err=float((got - ref).abs().max()), out_scale=scale,
           err_vs_fp32_conv=float((got - (F.conv3d(x.double(), w.double(), b.double(), padding=1) + (a.double() if with_addend else 0))).abs().max()))


@pytest.mark.parametrize('shape', [(1, 32, 2, 8, 8, 16), (2, 32, 2, 16, 16, 32), (1, 32, 3, 5, 9, 11), (2, 16, 1, 6, 10, 18),
                                   (1, 16, 3, 9, 7, 33), (1, 32, 2, 3, 2, 1), (1, 32, 2, 24, 24, 48)])
@pytest.mark.parametrize('flip', [0, 1])
def test_conv3d_k3_thin_out_mfma(hip_device, shape, flip):
    """head conv on the matrix cores (bf16 activations, x taps in K, (kz, ky) taps in the output rows, weights as bf16
    hi + lo): equals the conv of the bf16-rounded input with the fp32 weights (weight error 2^-17, tolerance 2e-5 of the
    output scale -- the bound of the bf16 kernels whose weights are rounded), ragged tiles, flipped taps (the adjoint
    form), no-bias / no-stats calls, and per-tile statistics equal to those of its own output"""
    from segmentation3d import _ops, _engine as E
    N, Cin, Cout, D, H, W = shape
    assert E.query('seg3d_conv3d_k3_thin_out_mfma_supported', Cin, Cout) == 1
    assert E.query('seg3d_conv3d_k3_thin_out_mfma_supported', 24, 2) == 0 and E.query('seg3d_conv3d_k3_thin_out_mfma_supported', 32, 4) == 0
    x = _t(51, 'hx', (N, Cin, D, H, W))
    w = _t(52, 'hw', (Cout, Cin, 3, 3, 3), std=0.05)
    b = _t(53, 'hb', (Cout,), std=0.5)
    xb = _ops.to_ndhwc(x.to(hip_device)).bfloat16()
    wd = w.to(hip_device)
    wp = torch.empty(E.query('seg3d_thin_out_mfma_packed_elems', Cin), dtype=torch.bfloat16, device=hip_device)
    E.call('seg3d_pack_weights_thin_out_mfma', E.ptr(wd), E.ptr(wp), Cin, Cout, 27, Cin * 27, flip, E.stream_ptr())
    y = torch.full((N, D, H, W, Cout), float('nan'), device=hip_device)
    cnt = E.query('seg3d_conv3d_k3_thin_out_stats_count', D, H, W)
    st = torch.full((N, cnt, 2), float('nan'), device=hip_device)
    bd = b.to(hip_device)
    E.call('seg3d_conv3d_k3_thin_out_mfma_fwd', E.ptr(xb), E.ptr(wp), E.ptr(bd), E.ptr(y), E.ptr(st), N, D, H, W, Cin, Cout,
           E.stream_ptr())
    wref = w.flip(2, 3, 4) if flip else w
    ref = F.conv3d(x.bfloat16().double(), wref.double(), b.double(), padding=1)
    got = _ops.from_ndhwc(y).double().cpu()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) < 2e-5 * scale
    s = st.double().sum(1).cpu()
    rr = got.reshape(N, -1)
    assert float(((s[:, 0] - rr.sum(1)).abs() / rr.abs().sum(1)).max()) < 1e-5 and rel_err(s[:, 1], (rr * rr).sum(1)) < 1e-5
    y2 = torch.full((N, D, H, W, Cout), float('nan'), device=hip_device)
    E.call('seg3d_conv3d_k3_thin_out_mfma_fwd', E.ptr(xb), E.ptr(wp), None, E.ptr(y2), None, N, D, H, W, Cin, Cout, E.stream_ptr())
    assert torch.equal(y2 + bd, y) or float((y2 + bd - y).abs().max()) < 1e-6 * scale
    report('thin_out_mfma_{}x{}x{}x{}_{}_{}_flip{}'.format(N, D, H, W, Cin, Cout, flip), max_abs_err=float((got - ref).abs().max()),
           out_scale=scale)


@pytest.mark.parametrize('shape', [(1, 32, 2, 8, 8, 16), (2, 32, 2, 16, 16, 32), (1, 32, 3, 5, 9, 11), (2, 16, 1, 6, 10, 18),
                                   (1, 16, 3, 9, 7, 33), (1, 32, 2, 3, 2, 1), (1, 32, 2, 40, 24, 48), (1, 32, 4, 7, 9, 17),
                                   (2, 32, 5, 6, 10, 18), (1, 16, 4, 4, 8, 16), (1, 16, 5, 12, 8, 16), (3, 16, 2, 33, 8, 16)])
@pytest.mark.parametrize('flip', [0, 1])
def test_conv3d_k3_thin_out_f32mfma(hip_device, shape, flip):
    """fp32 head conv on the fp32 matrix cores (x taps in K of v_mfma_f32_16x16x4_f32, (kz, ky) taps in the output rows,
    ninth tap on the VALU for 2 / 4 classes) against the float64 convolution: 1..5 classes, Cin 16 / 32, ragged tiles in
    every dimension, z marches longer than one tile, flipped taps (the adjoint form), no-bias / no-stats calls, and
    per-tile statistics equal to those of its own output"""
    from segmentation3d import _ops, _engine as E
    N, Cin, Cout, D, H, W = shape
    assert E.query('seg3d_conv3d_k3_thin_out_f32mfma_supported', Cin, Cout) == 1
    assert E.query('seg3d_conv3d_k3_thin_out_f32mfma_supported', 24, 2) == 0 and E.query('seg3d_conv3d_k3_thin_out_f32mfma_supported', 32, 6) == 0
    x = _t(61, 'fx', (N, Cin, D, H, W))
    w = _t(62, 'fw', (Cout, Cin, 3, 3, 3), std=0.05)
    b = _t(63, 'fb', (Cout,), std=0.5)
    xn = _ops.to_ndhwc(x.to(hip_device))
    wd = w.to(hip_device)
    wp = torch.full((E.query('seg3d_thin_out_f32mfma_packed_floats', Cin, Cout),), float('nan'), device=hip_device)
    E.call('seg3d_pack_weights_thin_out_f32mfma', E.ptr(wd), E.ptr(wp), Cin, Cout, 27, Cin * 27, flip, E.stream_ptr())
    y = torch.full((N, D, H, W, Cout), float('nan'), device=hip_device)
    cnt = E.query('seg3d_conv3d_k3_thin_out_f32mfma_stats_count', N, D, H, W)
    st = torch.full((N, cnt, 2), float('nan'), device=hip_device)
    bd = b.to(hip_device)
    E.call('seg3d_conv3d_k3_thin_out_f32mfma_fwd', E.ptr(xn), E.ptr(wp), E.ptr(bd), E.ptr(y), E.ptr(st), N, D, H, W, Cin, Cout,
           E.stream_ptr())
    wref = w.flip(2, 3, 4) if flip else w
    ref = F.conv3d(x.double(), wref.double(), b.double(), padding=1)
    got = _ops.from_ndhwc(y).double().cpu()
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max())
    assert err < 2e-6 * scale, (err, scale)
    s = st.double().sum(1).cpu()
    rr = got.reshape(N, -1)
    assert float(((s[:, 0] - rr.sum(1)).abs() / rr.abs().sum(1)).max()) < 1e-5 and rel_err(s[:, 1], (rr * rr).sum(1)) < 1e-5
    y2 = torch.full((N, D, H, W, Cout), float('nan'), device=hip_device)
    E.call('seg3d_conv3d_k3_thin_out_f32mfma_fwd', E.ptr(xn), E.ptr(wp), None, E.ptr(y2), None, N, D, H, W, Cin, Cout,
           E.stream_ptr())
    assert torch.equal(y2 + bd, y) or float((y2 + bd - y).abs().max()) < 1e-6 * scale
    report('thin_out_f32mfma_{}x{}x{}x{}_{}_{}_flip{}'.format(N, D, H, W, Cin, Cout, flip), max_abs_err=err, out_scale=scale)


@pytest.mark.parametrize('shape', [(1, 32, 2, 4, 8, 8), (2, 32, 2, 12, 16, 24), (1, 32, 3, 5, 9, 11), (2, 16, 1, 6, 10, 18),
                                   (1, 16, 5, 3, 7, 9), (1, 64, 8, 4, 8, 16), (1, 32, 2, 2, 1, 3)])
@pytest.mark.parametrize('accumulate', [False, True])
def test_k3_thin_wgrad_fatbf16(hip_device, shape, accumulate):
    """head weight gradient in bf16 mode (thin = fp32 dy with <= 8 channels, fat = the bf16 input) on the bf16 matrix
    cores with dy as a bf16 hi + lo pair: equals the exact weight gradient of the bf16 input and the fp32 dy to 2e-5 of
    the gradient scale; ragged tiles, accumulate flag"""
    from segmentation3d import _ops, _engine as E
    N, Cin, Cout, D, H, W = shape
    x = _t(61, 'gx', (N, Cin, D, H, W))
    dy = _t(62, 'gdy', (N, Cout, D, H, W))
    base = _t(63, 'gbase', (Cout, Cin, 3, 3, 3))
    xb = _ops.to_ndhwc(x.to(hip_device)).bfloat16()
    dyn = _ops.to_ndhwc(dy.to(hip_device))
    dw = base.to(hip_device).clone() if accumulate else torch.full((Cout, Cin, 3, 3, 3), float('nan'), device=hip_device)
    ws = torch.empty(E.query('seg3d_k3_thin_wgrad_workspace_floats', N, D, H, W, Cout, Cin), device=hip_device)
    E.call('seg3d_k3_thin_wgrad_fatbf16', E.ptr(dyn), E.ptr(xb), E.ptr(dw), E.ptr(ws), N, D, H, W, Cout, Cin, Cin * 27, 27, 1,
           int(accumulate), E.stream_ptr())
    wz = torch.zeros(Cout, Cin, 3, 3, 3, dtype=torch.double, requires_grad=True)
    (F.conv3d(x.bfloat16().double(), wz, None, padding=1) * dy.double()).sum().backward()
    ref = wz.grad + (base.double() if accumulate else 0)
    got = dw.double().cpu()
    scale = float(wz.grad.abs().max())
    assert float((got - ref).abs().max()) < 2e-5 * scale
    report('thin_wgrad_fatbf16_{}x{}x{}x{}_{}_{}{}'.format(N, D, H, W, Cin, Cout, '_acc' if accumulate else ''),
           max_abs_err=float((got - ref).abs().max()), grad_scale=scale)


@pytest.mark.parametrize('shape', [(2, 1, 16, 8, 8, 16), (1, 1, 16, 5, 9, 11), (1, 2, 32, 6, 10, 18), (2, 2, 32, 4, 8, 8),
                                   (1, 1, 64, 3, 7, 9), (1, 2, 20, 9, 5, 13), (1, 1, 16, 1, 2, 3)])
@pytest.mark.parametrize('flip', [0, 1])
def test_conv3d_k3_thin_in_mfma16(hip_device, shape, flip):
    """thin-input conv on the bf16 matrix cores (stem forward / head data-gradient in bf16 mode): fp32 input and
    weights as bf16 hi + lo pairs, bf16 output = one rounding of the fp32-grade result (tolerance: half a bf16 ulp at
    the output scale = 2^-8 of it, + 2e-5); statistics are taken from the unrounded values"""
    from segmentation3d import _ops, _engine as E
    N, CT, Cout, D, H, W = shape
    assert E.query('seg3d_conv3d_k3_thin_in_mfma16_supported', CT, Cout) == 1
    assert E.query('seg3d_conv3d_k3_thin_in_mfma16_supported', 3, 16) == 0
    x = _t(71, 'tx', (N, CT, D, H, W))
    w = _t(72, 'tw', (Cout, CT, 3, 3, 3), std=0.2)
    b = _t(73, 'tb', (Cout,), std=0.5)
    xn = _ops.to_ndhwc(x.to(hip_device))
    wd, bd = w.to(hip_device), b.to(hip_device)
    wq = torch.empty(E.query('seg3d_packed_thin_in16_elems', CT, Cout), dtype=torch.bfloat16, device=hip_device)
    E.call('seg3d_pack_weights_thin_in16', E.ptr(wd), E.ptr(wq), CT, Cout, 27, CT * 27, flip, E.stream_ptr())
    y = torch.full((N, D, H, W, Cout), float('nan'), dtype=torch.bfloat16, device=hip_device)
    cnt = E.query('seg3d_conv3d_k3_thin_stats_count', D, H, W, (Cout + 31) // 32)
    st = torch.full((N, cnt, 2), float('nan'), device=hip_device)
    E.call('seg3d_conv3d_k3_thin_in_mfma16_fwd', E.ptr(xn), E.ptr(wq), E.ptr(bd), E.ptr(y), E.ptr(st), N, D, H, W, CT, Cout,
           E.stream_ptr())
    wref = w.flip(2, 3, 4) if flip else w
    ref = F.conv3d(x.double(), wref.double(), b.double(), padding=1)
    got = _ops.from_ndhwc(y.float()).double().cpu()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) < (2.0 ** -8 + 2e-5) * scale
    s = st.double().sum(1).cpu()
    rr = ref.reshape(N, -1)
    assert float(((s[:, 0] - rr.sum(1)).abs() / rr.abs().sum(1)).max()) < 2e-5 and rel_err(s[:, 1], (rr * rr).sum(1)) < 2e-5
    report('thin_in_mfma16_{}x{}x{}x{}_{}_{}_flip{}'.format(N, D, H, W, CT, Cout, flip), max_abs_err=float((got - ref).abs().max()),
           out_scale=scale, stats_rel_err=rel_err(s[:, 1], (rr * rr).sum(1)))


@pytest.mark.parametrize('shape', [(2, 1, 16, 8, 8, 16), (1, 1, 16, 5, 9, 11), (1, 2, 32, 6, 10, 18), (2, 2, 32, 4, 8, 8),
                                   (1, 4, 16, 12, 16, 24), (1, 3, 20, 9, 5, 13), (1, 5, 48, 4, 8, 16), (3, 1, 16, 20, 24, 40),
                                   (1, 8, 16, 4, 8, 8), (1, 1, 16, 1, 2, 3)])
@pytest.mark.parametrize('flip', [0, 1])
@pytest.mark.parametrize('out_bf16', [0, 1])
def test_conv3d_k3_thin_in_persistent(hip_device, shape, flip, out_bf16):
    """persistent thin-input conv (stem forward / head data-gradient, fp32 arithmetic) against the float64 convolution:
    1..8 thin channels (even: immediate LDS offsets, odd: per-lane table), whole and ragged tiles, several tiles per
    workgroup, partial channel blocks, flipped taps, fp32 and bf16-storage outputs, no-bias / no-stats calls, per-wave
    statistics equal to those of the unrounded output"""
    from segmentation3d import _ops, _engine as E
    N, CT, Cout, D, H, W = shape
    x = _t(74, 'px', (N, CT, D, H, W))
    w = _t(75, 'pw', (Cout, CT, 3, 3, 3), std=0.2)
    b = _t(76, 'pb', (Cout,), std=0.5)
    xn = _ops.to_ndhwc(x.to(hip_device))
    wd, bd = w.to(hip_device), b.to(hip_device)
    wp = torch.full((E.query('seg3d_packed_thin_in_floats', CT, Cout),), float('nan'), device=hip_device)
    E.call('seg3d_pack_weights_thin_in', E.ptr(wd), E.ptr(wp), CT, Cout, 27, CT * 27, flip, E.stream_ptr())
    ydt = torch.bfloat16 if out_bf16 else torch.float32
    y = torch.full((N, D, H, W, Cout), float('nan'), dtype=ydt, device=hip_device)
    cnt = E.query('seg3d_conv3d_k3_thin_in_persistent_stats_count', D, H, W, (Cout + 31) // 32)
    st = torch.full((N, cnt, 2), float('nan'), device=hip_device)
    E.call('seg3d_conv3d_k3_thin_in_persistent_fwd', E.ptr(xn), E.ptr(wp), E.ptr(bd), E.ptr(y), E.ptr(st), N, D, H, W, CT, Cout,
           out_bf16, E.stream_ptr())
    wref = w.flip(2, 3, 4) if flip else w
    ref = F.conv3d(x.double(), wref.double(), b.double(), padding=1)
    got = _ops.from_ndhwc(y.float()).double().cpu()
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max())
    assert err < ((2.0 ** -8 + 2e-5) if out_bf16 else 2e-6) * scale, (err, scale)
    s = st.double().sum(1).cpu()
    rr = ref.reshape(N, -1)
    assert float(((s[:, 0] - rr.sum(1)).abs() / rr.abs().sum(1)).max()) < 2e-5 and rel_err(s[:, 1], (rr * rr).sum(1)) < 2e-5
    y2 = torch.full((N, D, H, W, Cout), float('nan'), dtype=ydt, device=hip_device)
    E.call('seg3d_conv3d_k3_thin_in_persistent_fwd', E.ptr(xn), E.ptr(wp), None, E.ptr(y2), None, N, D, H, W, CT, Cout,
           out_bf16, E.stream_ptr())
    assert float((y2.float() + bd - y.float()).abs().max()) < (2.0 ** -7 if out_bf16 else 1e-6) * scale
    report('thin_in_persistent_{}x{}x{}x{}_{}_{}_flip{}_bf{}'.format(N, D, H, W, CT, Cout, flip, out_bf16), max_abs_err=err,
           out_scale=scale)


@pytest.mark.parametrize('ncls', [3, 5, 7])
def test_head_dgrad_thin_channels_on_winograd(hip_device, ncls, monkeypatch):
    """data-gradient of a 3..7-class head (network/module/vnet_outblock.py:13) on a level big enough for the Winograd kernel:
    _ops zero-pads the thin side to 8 channels and runs conv3d_k3_wino2d_kernel (recorded and asserted) -- against torch"""
    from segmentation3d import _ops, _engine as E
    N, C, D = 1, 32, 48
    dy = _t(81, 'hdy', (N, ncls, D, D, D))
    w = _t(82, 'hw', (ncls, C, 3, 3, 3), std=0.1)
    x = torch.zeros((N, C, D, D, D), dtype=torch.float64, requires_grad=True)
    F.conv3d(x, w.double(), None, padding=1).backward(dy.double())
    called = []
    orig_call = E.call
    monkeypatch.setattr(E, 'call', lambda fn, *a: (called.append(fn), orig_call(fn, *a))[1])
    dx = _ops.conv_dgrad(_ops.to_ndhwc(dy.to(hip_device)), w.to(hip_device), 'k3')
    assert 'seg3d_conv3d_k3_wino2d_fwd' in called and 'seg3d_conv3d_k3_thin_in_persistent_fwd' not in called, called
    err = rel_err(_ops.from_ndhwc(dx), x.grad)
    report('head_dgrad_{}cls_wino2d'.format(ncls), rel_err=err)
    assert err < 1e-5, err


@pytest.mark.parametrize('shape', [(2, 1, 16, 8, 8, 16), (1, 4, 16, 5, 9, 11), (1, 2, 32, 6, 10, 18)])
def test_k3_thin_wgrad_stem_bf16_dy(hip_device, shape):
    """stem weight gradient in bf16 mode: thin = the fp32 image (hi + lo inside the kernel), fat = the bf16 dy"""
    from segmentation3d import _ops, _engine as E
    N, Cin, Cout, D, H, W = shape
    x = _t(64, 'sx', (N, Cin, D, H, W))
    dy = _t(65, 'sdy', (N, Cout, D, H, W))
    xn = _ops.to_ndhwc(x.to(hip_device))
    dyb = _ops.to_ndhwc(dy.to(hip_device)).bfloat16()
    dw = _ops.conv_wgrad(xn, dyb, (Cout, Cin, 3, 3, 3), 'k3')
    wz = torch.zeros(Cout, Cin, 3, 3, 3, dtype=torch.double, requires_grad=True)
    (F.conv3d(x.double(), wz, None, padding=1) * dy.bfloat16().double()).sum().backward()
    scale = float(wz.grad.abs().max())
    err = float((dw.double().cpu() - wz.grad).abs().max())
    assert err < 2e-5 * scale
    report('thin_wgrad_stem_bf16dy_{}x{}x{}x{}_{}_{}'.format(N, D, H, W, Cin, Cout), max_abs_err=err, grad_scale=scale)


@pytest.mark.parametrize('shape', [(2, 16, 32, 4, 8, 8), (1, 64, 16, 3, 5, 6), (2, 128, 256, 2, 4, 8), (1, 32, 32, 6, 6, 6),
                                   (1, 24, 40, 2, 4, 8)])
@pytest.mark.parametrize('transposed', [False, True])
def test_k2_bf16_wgrad(hip_device, shape, transposed):
    """weight gradient of the stride-2 2x2x2 conv / transposed conv on bf16 operands (bf16-MFMA kernel with transposing
    LDS reads; channel counts that are not multiples of 8 take the widening kernel): equals the exact gradient of the
    bf16-rounded operands to fp32 accumulation error"""
    from segmentation3d import _ops
    N, Cin, Cout, Dq, Hq, Wq = shape          # conv: x [N,Cin,2Dq,..] -> y [N,Cout,Dq,..]; transposed: x [N,Cin,Dq,..] -> 2x
    if transposed:
        x = _t(61, 'k2x', (N, Cin, Dq, Hq, Wq)).bfloat16()
        dy = _t(62, 'k2dy', (N, Cout, 2 * Dq, 2 * Hq, 2 * Wq)).bfloat16()
        w0 = torch.zeros(Cin, Cout, 2, 2, 2, dtype=torch.double, requires_grad=True)
        F.conv_transpose3d(x.double(), w0, None, stride=2).backward(dy.double())
        kind, wshape = 'convT', (Cin, Cout, 2, 2, 2)
    else:
        x = _t(61, 'k2x', (N, Cin, 2 * Dq, 2 * Hq, 2 * Wq)).bfloat16()
        dy = _t(62, 'k2dy', (N, Cout, Dq, Hq, Wq)).bfloat16()
        w0 = torch.zeros(Cout, Cin, 2, 2, 2, dtype=torch.double, requires_grad=True)
        F.conv3d(x.double(), w0, None, stride=2).backward(dy.double())
        kind, wshape = 'k2s2', (Cout, Cin, 2, 2, 2)
    nd = lambda t: t.to(hip_device).permute(0, 2, 3, 4, 1).contiguous()
    dw = _ops.conv_wgrad(nd(x), nd(dy), wshape, kind)
    err = float((dw.double().cpu() - w0.grad).abs().max()) / float(w0.grad.abs().max())
    report('bf16_k2wgrad_{}_{}x{}x{}x{}_{}_{}'.format(kind, N, Dq, Hq, Wq, Cin, Cout), rel_max=err)
    assert err < 2e-5, err


def test_bf16_multi_pack_equals_single_pack(hip_device):
    """PackedWeightCache.repack_all() refreshes every bf16 weight image with ONE launch (pack_mfma_bf16_multi_kernel,
    coalesced read / LDS transpose); the images equal those of the per-tensor pack kernel bit for bit, both weight
    orientations (forward: w[b][a][t]; data-gradient: flipped taps, w[a][b][t] strides), partial channel blocks included"""
    from segmentation3d import _ops, _engine as E
    cache = _ops.PackedWeightCache()
    cache.enabled = True
    specs = [(32, 32), (64, 16), (16, 48), (256, 128)]
    ws, images = [], []
    for k, (cin, cout) in enumerate(specs):
        w = _t(80 + k, 'pk', (cout, cin, 3, 3, 3), std=0.1).to(hip_device)
        ws.append(w)
        images.append(cache.get(w, cin, cout, 27, 27, cin * 27, 0, bf16=True))          # forward orientation
        images.append(cache.get(w, cout, cin, 27, cin * 27, 27, 1, bf16=True))          # data-gradient orientation
    for w in ws:
        w.mul_(1.5).add_(0.01)
    cache.repack_all()
    k = 0
    for w, (cin, cout) in zip(ws, specs):
        for (A, B, sa, sb, flip) in ((cin, cout, 27, cin * 27, 0), (cout, cin, cin * 27, 27, 1)):
            ref = torch.empty(E.query('seg3d_packed_mfma_bf16_elems', A, B, 27), dtype=torch.bfloat16, device=hip_device)
            E.call('seg3d_pack_weights_mfma_bf16', E.ptr(w), E.ptr(ref), A, B, 27, sa, sb, flip, E.stream_ptr())
            assert torch.equal(images[k].view(torch.int16).cpu(), ref.view(torch.int16).cpu()), (cin, cout, flip)
            k += 1


def test_fp32_multi_pack_equals_single_pack(hip_device):
    """PackedWeightCache.repack_all() in fp32: ONE launch (pack_mfma_multi_kernel) refreshes the 27-tap images, the Winograd
    F(2, 3) (T = 36) and F(2x2, 3x3) (T = 48: one thread per channel pair, twelve 16-byte words each) images and the 8-tap
    stride-2 images; every image equals the per-tensor pack kernel's bit for bit, both weight orientations (the data-gradient
    one with flipped taps), partial channel blocks included"""
    from segmentation3d import _ops, _engine as E
    cache = _ops.PackedWeightCache()
    cache.enabled = True
    specs = [(32, 32), (64, 16), (16, 48), (256, 128), (40, 72)]
    ws, images = [], []
    for k, (cin, cout) in enumerate(specs):
        w = _t(180 + k, 'pk32', (cout, cin, 3, 3, 3), std=0.1).to(hip_device)
        ws.append(w)
        for T in (27, 36, 48):
            images.append(cache.get(w, cin, cout, T, 27, cin * 27, 0))          # forward orientation
            images.append(cache.get(w, cout, cin, T, cin * 27, 27, 1))          # data-gradient orientation
    w8 = _t(190, 'pk8', (32, 16, 2, 2, 2), std=0.1).to(hip_device)               # stride-2 conv 16 -> 32
    img8 = cache.get(w8, 16, 32, 8, 8, 16 * 8, 0)
    for w in ws + [w8]:
        w.mul_(1.5).add_(0.01)
    cache.repack_all()
    k = 0
    for w, (cin, cout) in zip(ws, specs):
        for T in (27, 36, 48):
            for (A, B, sa, sb, flip) in ((cin, cout, 27, cin * 27, 0), (cout, cin, cin * 27, 27, 1)):
                ref = torch.full((E.query('seg3d_packed_mfma_floats', A, B, T),), float('nan'), device=hip_device)
                E.call('seg3d_pack_weights_mfma', E.ptr(w), E.ptr(ref), A, B, T, sa, sb, flip, E.stream_ptr())
                assert torch.equal(images[k].view(torch.int32).cpu(), ref.view(torch.int32).cpu()), (cin, cout, T, flip)
                k += 1
    ref = torch.empty((E.query('seg3d_packed_mfma_floats', 16, 32, 8),), device=hip_device)
    E.call('seg3d_pack_weights_mfma', E.ptr(w8), E.ptr(ref), 16, 32, 8, 8, 16 * 8, 0, E.stream_ptr())
    assert torch.equal(img8.view(torch.int32).cpu(), ref.view(torch.int32).cpu())


@pytest.mark.parametrize('shape', [(1, 32, 32, 16, 16, 16), (2, 48, 20, 4, 12, 20), (4, 256, 256, 6, 6, 6)])
def test_conv3d_k3_bf16_out_bf16_is_rounded_fp32(hip_device, shape):
    """out_bf16 = 1 (data-gradient outputs): the stored tensor is exactly the bf16 rounding of what the fp32-output
    launch stores (same accumulators, bias and addend; whole-K and split-K paths)"""
    from segmentation3d import _ops, _engine as E
    N, Cin, Cout, D, H, W = shape
    xb = _ops.to_ndhwc(_t(45, 'ox', (N, Cin, D, H, W)).to(hip_device)).bfloat16()
    wd = _t(46, 'ow', (Cout, Cin, 3, 3, 3), std=0.05).to(hip_device)
    an = _ops.to_ndhwc(_t(47, 'oa', (N, Cout, D, H, W)).to(hip_device))
    wp = torch.empty(E.query('seg3d_packed_mfma_bf16_elems', Cin, Cout, 27), dtype=torch.bfloat16, device=hip_device)
    E.call('seg3d_pack_weights_mfma_bf16', E.ptr(wd), E.ptr(wp), Cin, Cout, 27, 27, Cin * 27, 0, E.stream_ptr())
    ws = torch.empty(max(E.query('seg3d_conv3d_k3_bf16_fwd_workspace_floats', N, D, H, W, Cin, Cout), 1), device=hip_device)
    y32 = torch.empty(N, D, H, W, Cout, device=hip_device)
    y16 = torch.empty(N, D, H, W, Cout, device=hip_device, dtype=torch.bfloat16)
    for y, flag in ((y32, 0), (y16, 1)):
        E.call('seg3d_conv3d_k3_bf16_fwd', E.ptr(xb), E.ptr(wp), None, E.ptr(an), E.ptr(y), None, E.ptr(ws),
               N, D, H, W, Cin, Cout, flag, E.stream_ptr())
    assert torch.equal(y16.view(torch.int16).cpu(), y32.bfloat16().view(torch.int16).cpu())


@pytest.mark.parametrize('shape', [(1, 32, 32, 8, 8, 16), (2, 64, 32, 12, 12, 12), (1, 16, 48, 4, 8, 8), (2, 128, 128, 4, 4, 8),
                                   (1, 32, 32, 6, 6, 6), (1, 32, 16, 5, 8, 8), (4, 32, 32, 16, 16, 32), (2, 64, 64, 7, 9, 10),
                                   (4, 256, 256, 6, 6, 6), (1, 16, 16, 3, 3, 3), (1, 12, 20, 4, 8, 8)])
@pytest.mark.parametrize('accumulate', [False, True])
def test_conv3d_k3_bf16_wgrad(hip_device, shape, accumulate):
    """bf16 weight gradient (bf16 x and dy, fp32 dw): equals the exact weight gradient of the bf16-rounded operands to
    fp32 accumulation error -- the bf16-MFMA kernel with transposing LDS reads (tiles inside the volume and tiles that
    stick out of it), the register-staged fallback for channel counts that are not multiples of 8; `accumulate` adds into
    an existing gradient"""
    from segmentation3d import _ops, _engine as E
    N, Cin, Cout, D, H, W = shape
    x = _t(51, 'wx', (N, Cin, D, H, W)).bfloat16()
    dy = _t(52, 'wdy', (N, Cout, D, H, W)).bfloat16()
    xd = x.double()
    w0 = torch.zeros(Cout, Cin, 3, 3, 3, dtype=torch.double, requires_grad=True)
    F.conv3d(xd, w0, None, padding=1).backward(dy.double())
    ref = w0.grad
    xn = _ops.to_ndhwc(x.to(hip_device).permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3))
    dyn = _ops.to_ndhwc(dy.to(hip_device).permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3))
    base = _t(53, 'wbase', (Cout, Cin, 3, 3, 3)).to(hip_device)
    dw = base.clone() if accumulate else torch.empty_like(base)
    ws = torch.empty(E.query('seg3d_conv3d_k3_bf16_wgrad_workspace_floats', N, D, H, W, Cin, Cout), device=hip_device)
    E.call('seg3d_conv3d_k3_bf16_wgrad', E.ptr(xn), E.ptr(dyn), E.ptr(dw), E.ptr(ws), N, D, H, W, Cin, Cout,
           int(accumulate), E.stream_ptr())
    got = dw.double().cpu() - (base.double().cpu() if accumulate else 0)
    err = float((got - ref).abs().max()) / float(ref.abs().max())
    report('bf16wgrad_{}x{}x{}x{}_{}_{}{}'.format(N, D, H, W, Cin, Cout, '_acc' if accumulate else ''), rel_max=err)
    assert err < 2e-5, err


@pytest.mark.parametrize('shape', [(2, 16, 48, 6, 10, 20), (1, 128, 64, 4, 4, 4), (4, 256, 256, 6, 6, 6)])
def test_conv3d_k3_mfma_stats_partials(hip_device, shape):
    """the conv epilogue's per-workgroup (sum, sumsq) equal the statistics of its own output (incl. the split-K path
    used on deep, spatially tiny levels)"""
    from segmentation3d import _ops
    N, Cin, Cout, D, H, W = shape
    x = _t(5, 'stx', (N, Cin, D, H, W)).to(hip_device)
    w = _t(6, 'stw', (Cout, Cin, 3, 3, 3), std=0.1).to(hip_device)
    b = _t(7, 'stb', (Cout,), std=0.5).to(hip_device)
    y, part = _ops.conv_forward(_ops.to_ndhwc(x), w, b, 'k3', want_stats=True)
    assert part is not None
    s = part.double().sum(1).cpu()
    yy = y.double().reshape(N, -1).cpu()
    assert float(((s[:, 0] - yy.sum(1)).abs() / yy.abs().sum(1)).max()) < 1e-5 and rel_err(s[:, 1], (yy * yy).sum(1)) < 1e-5
    ref = F.conv3d(x.cpu(), w.cpu(), b.cpu(), padding=1)
    assert max_err(_ops.from_ndhwc(y), ref) < 1e-4


@pytest.mark.parametrize('force_direct', [False, True])
@pytest.mark.parametrize('kind,cin,cout,dims', [
    ('k2s2', 16, 32, (2, 8, 8, 8)), ('k2s2', 1, 16, (1, 4, 6, 8)), ('k2s2', 128, 256, (1, 2, 4, 4)),
    ('k2s2', 32, 64, (1, 6, 10, 12)), ('k2s2', 12, 20, (2, 2, 6, 14)),
    ('convT', 64, 16, (1, 4, 4, 6)), ('convT', 256, 128, (2, 2, 2, 2)), ('convT', 3, 5, (1, 3, 3, 3)),
    ('convT', 128, 32, (1, 3, 5, 7)), ('convT', 16, 24, (2, 5, 4, 9)), ('convT', 32, 8, (1, 3, 5, 7)), ('convT', 64, 16, (2, 6, 7, 9)),
    ('k1', 2, 2, (2, 8, 8, 8)), ('k1', 5, 5, (1, 4, 4, 4)), ('k1', 16, 8, (1, 4, 4, 4)),
])
def test_strided_convs_fwd_bwd(hip_device, kind, cin, cout, dims, force_direct):
    from segmentation3d import _ops
    N, D, H, W = dims
    if force_direct and (kind == 'k1' or cin < 8):
        pytest.skip('already the direct path')
    name = 'conv_{}_{}_{}_{}_{}'.format(kind, 'direct' if force_direct else 'auto', cin, cout, '_'.join(map(str, dims)))
    x = _t(11, name + 'x', (N, cin, D, H, W)).requires_grad_(True)
    if kind == 'convT':
        w = _t(12, name + 'w', (cin, cout, 2, 2, 2), std=(1.0 / cin) ** 0.5).requires_grad_(True)
        ref_fn = lambda a, ww, bb: F.conv_transpose3d(a, ww, bb, stride=2)
    elif kind == 'k2s2':
        w = _t(12, name + 'w', (cout, cin, 2, 2, 2), std=(1.0 / (8 * cin)) ** 0.5).requires_grad_(True)
        ref_fn = lambda a, ww, bb: F.conv3d(a, ww, bb, stride=2)
    else:
        w = _t(12, name + 'w', (cout, cin, 1, 1, 1), std=(1.0 / cin) ** 0.5).requires_grad_(True)
        ref_fn = lambda a, ww, bb: F.conv3d(a, ww, bb)
    b = _t(13, name + 'b', (cout,), std=0.1).requires_grad_(True)
    ref = ref_fn(x, w, b)
    g = _t(14, name + 'g', tuple(ref.shape))
    rdx, rdw, rdb = torch.autograd.grad(ref, (x, w, b), g)
    xd, wd, bd = (t.detach().to(hip_device).requires_grad_(True) for t in (x, w, b))
    _ops.FORCE_DIRECT = force_direct
    try:
        out = _ops.conv(xd, wd, bd, kind)
        dx, dw, db = torch.autograd.grad(out, (xd, wd, bd), g.to(hip_device))
        # the fused unit also consumes the conv-epilogue GroupNorm statistics of the strided kernels
        yn, part = _ops.conv_forward(_ops.to_ndhwc(xd.detach()), wd.detach(), bd.detach(), kind, want_stats=True)
    finally:
        _ops.FORCE_DIRECT = False
    e = dict(out=max_err(out, ref), dx=rel_err(dx, rdx), dw=rel_err(dw, rdw), db=rel_err(db, rdb))
    if part is not None:
        s = part.double().sum(1).cpu()
        yy = ref.detach().double().reshape(N, -1)
        e['stat_sum'] = float(((s[:, 0] - yy.sum(1)).abs() / (yy.abs().sum(1) + 1e-30)).max())
        e['stat_sq'] = rel_err(s[:, 1], (yy * yy).sum(1))
    report(name, **e)
    assert e['out'] < 1e-4 and e['dx'] < 1e-4 and e['dw'] < 2e-4 and e['db'] < 1e-4, e
    assert e.get('stat_sum', 0.0) < 1e-5 and e.get('stat_sq', 0.0) < 1e-5, e


def test_gn_backward_fused_finalize_equals_two_launches(hip_device):
    """GroupNorm backward with both finalize stages in one launch (last-ticket workgroup) is bit-identical to the
    two-launch path, repeatedly (the ticket counter returns to zero), with and without gradient sinks"""
    from segmentation3d import _ops
    N, C, D, H, W = 3, 64, 6, 10, 12
    y = _ops.to_ndhwc(_t(91, 'fy', (N, C, D, H, W)).to(hip_device))
    dout = _ops.to_ndhwc(_t(92, 'fd', (N, C, D, H, W)).to(hip_device))
    gamma, beta = _t(93, 'fg', (C,)).to(hip_device), _t(94, 'fb', (C,)).to(hip_device)
    mean_rstd = torch.stack([y.reshape(N, -1).mean(1), 1.0 / (y.reshape(N, -1).var(1, unbiased=False) + 1e-5).sqrt()], 1).contiguous()
    results = []
    was = _ops.GN_FUSED_FINALIZE
    for fused in (False, True, True, False, True):
        _ops.GN_FUSED_FINALIZE = fused
        try:
            sink = torch.ones(C, device=hip_device)
            dy, _, dgamma, dbeta, dbias = _ops.gn_backward(dout, None, y, mean_rstd, gamma, beta, True, want_dres=False,
                                                           want_dbias=True, sinks=(None, sink, None))
            results.append((dy, dgamma, sink, dbias))
        finally:
            _ops.GN_FUSED_FINALIZE = was
    assert int(_ops._gn_ticket(hip_device).item()) == 0
    for r in results[1:]:
        for a, b in zip(results[0], r):
            assert torch.equal(a, b)
    assert dbeta is None


def test_wgrad_reduce_many_chunks(hip_device):
    """1x1x1 conv weight gradient over enough voxels for the chunk-parallel reduce (>= 512 partial chunks, 4 outputs)"""
    from segmentation3d import _ops
    N, C, D, H, W = 2, 2, 48, 48, 48
    x = _t(95, 'rx', (N, C, D, H, W))
    dy = _t(96, 'rdy', (N, C, D, H, W))
    dw = _ops.conv_wgrad(_ops.to_ndhwc(x.to(hip_device)), _ops.to_ndhwc(dy.to(hip_device)), (C, C, 1, 1, 1), 'k1')
    ref = torch.einsum('nodhw,nidhw->oi', dy.double(), x.double()).reshape(C, C, 1, 1, 1)
    assert rel_err(dw.double().cpu(), ref) < 1e-5
    report('wgrad_reduce_many_chunks', rel_err=rel_err(dw.double().cpu(), ref))


@pytest.mark.parametrize('kind,shape', [
    ('k2s2', (4, 16, 64, 48, 48, 48)),     # gather, two column blocks per wave (top level: up_32.up_conv data-gradient)
    ('k2s2', (4, 16, 32, 48, 48, 48)),     # gather, top level forward: four steps, transposed stores
    ('k2s2', (4, 32, 64, 24, 24, 24)),     # gather, one column block per wave, 3 456 waves
    ('k2s2', (4, 64, 128, 12, 12, 12)),    # gather, K split over the four waves of a workgroup
    ('k2s2', (1, 32, 96, 5, 7, 9)),        # gather, ragged tiles, three column blocks
    ('k2s2', (2, 40, 40, 3, 4, 6)),        # gather, five chunks (an odd count: the staged kernel), a partial last column block
    ('convT', (4, 256, 128, 6, 6, 6)),     # scatter forward on the direct kernel (fewer than 256 staged workgroups)
    ('convT', (2, 64, 16, 5, 6, 7)),       # scatter forward, tap pairs in one accumulator (Cout = 16), ragged tiles
    ('convT', (1, 128, 32, 24, 24, 24)),   # scatter forward, staged kernel (one column block, 432 tiles)
])
def test_stride2_kernel_variants_full_grids(hip_device, kind, shape):
    """every variant of the stride-2 gather / scatter kernels at a grid that selects it (conv_k2_mfma.hip: direct kernels with one
    or two column blocks per wave, K split inside the workgroup, tap pairs; the staged kernels they fall back to) against a float64
    einsum over the 2^3 cells on the device; output and the per-sample statistics slots"""
    from segmentation3d import _ops
    N, Cin, Cout, D, H, W = shape          # D, H, W: the COARSE extent (output of the conv, input of the transposed conv)
    if kind == 'k2s2':
        x = _t(140, 'sx', (N, Cin, 2 * D, 2 * H, 2 * W)).to(hip_device)
        w = _t(141, 'sw', (Cout, Cin, 2, 2, 2), std=(1.0 / (8 * Cin)) ** 0.5).to(hip_device)
        cells = x.double().reshape(N, Cin, D, 2, H, 2, W, 2)
        ref = torch.einsum('nizaybxc,oiabc->nozyx', cells, w.double())
    else:
        x = _t(140, 'sx', (N, Cin, D, H, W)).to(hip_device)
        w = _t(141, 'sw', (Cin, Cout, 2, 2, 2), std=(1.0 / Cin) ** 0.5).to(hip_device)
        ref = torch.einsum('nizyx,ioabc->nozaybxc', x.double(), w.double()).reshape(N, Cout, 2 * D, 2 * H, 2 * W)
    b = _t(142, 'sb', (Cout,), std=0.1).to(hip_device)
    ref = ref + b.double().reshape(1, Cout, 1, 1, 1)
    yn, part = _ops.conv_forward(_ops.to_ndhwc(x), w, b, kind, want_stats=True)
    got = _ops.from_ndhwc(yn).double()
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max())
    rr = ref.reshape(N, -1)
    st = part.double().sum(1)
    e = dict(max_abs_err=err, out_scale=scale, stat_sum=float(((st[:, 0] - rr.sum(1)).abs() / rr.abs().sum(1)).max()),
             stat_sq=rel_err(st[:, 1].cpu(), (rr * rr).sum(1).cpu()))
    report('stride2_{}_{}'.format(kind, '_'.join(map(str, shape))), **e)
    assert err < 2e-6 * scale + 1e-6 and e['stat_sum'] < 1e-5 and e['stat_sq'] < 1e-5, e
    assert torch.isfinite(part).all()


@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
@pytest.mark.parametrize('shape', [(2, 32, 16, 4, 4, 8, 16), (1, 64, 32, 3, 5, 6, 0), (1, 16, 8, 2, 2, 2, 24)])
def test_k2s2_dgrad_with_skip_addend(hip_device, shape, mode):
    """stride-2 conv data-gradient with the skip gradient (a channel slice of a wider tensor) added in the epilogue equals
    the plain data-gradient + the slice, bit for bit in fp32 and to one bf16 rounding of the sum in bf16 mode"""
    from segmentation3d import _ops
    N, Cout, Cin, D, H, W, extra = shape
    bf = mode == 'bf16'
    dy = _t(81, 'kdy', (N, Cout, D, H, W)).to(hip_device)
    w = _t(82, 'kw', (Cout, Cin, 2, 2, 2), std=0.2).to(hip_device)
    wide = _t(83, 'kadd', (N, 2 * D, 2 * H, 2 * W, extra + Cin)).to(hip_device)
    dyn = _ops.to_ndhwc(dy)
    if bf:
        dyn, wide = dyn.bfloat16(), wide.bfloat16()
    addend = wide[..., extra:]
    plain = _ops.conv_dgrad(dyn, w, 'k2s2')
    fused = _ops.conv_dgrad(dyn, w, 'k2s2', addend=addend)
    assert fused.dtype == plain.dtype and fused.shape == plain.shape
    if bf and plain.dtype == torch.bfloat16:
        # plain was rounded once before the add; compare against the fp32 data-gradient of the same bf16 operands
        exact = _ops.conv_dgrad(dyn.float(), w, 'k2s2') + addend.float()
        scale = float(exact.abs().max())
        assert float((fused.float() - exact).abs().max()) < (2.0 ** -8 + 1e-3) * scale
    else:
        assert torch.equal(fused, plain + addend)
    report('k2s2_dgrad_addend_{}_{}'.format(mode, 'x'.join(str(v) for v in shape)), fused_is_bf16=float(fused.dtype == torch.bfloat16))


@pytest.mark.parametrize('C,dims,relu,with_res', [
    (16, (2, 8, 8, 8), True, False), (32, (1, 6, 10, 12), True, True), (256, (2, 2, 3, 3), False, True),
    (2, (2, 8, 8, 8), True, False), (5, (1, 4, 6, 8), False, False), (64, (1, 20, 20, 24), True, False),
    (128, (1, 4, 4, 4), False, False),
])
def test_fused_conv_gn_act(hip_device, C, dims, relu, with_res):
    """conv(k1, identity-free) + GroupNorm(1, C) [+res] [+ReLU] forward/backward against F.group_norm autograd"""
    from segmentation3d import _ops
    N, D, H, W = dims
    name = 'gn_{}_{}_{}_{}'.format(C, '_'.join(map(str, dims)), int(relu), int(with_res))
    x = _t(21, name + 'x', (N, C, D, H, W), std=2.0)
    x = (x + 0.7).requires_grad_(True)
    gamma = _t(22, name + 'ga', (C,), std=0.3).add(1.0).requires_grad_(True)
    beta = _t(23, name + 'be', (C,), std=0.3).requires_grad_(True)
    res = _t(24, name + 're', (N, C, D, H, W)).requires_grad_(True) if with_res else None
    ref = F.group_norm(x, 1, gamma, beta, 1e-5)
    if with_res:
        ref = ref + res
    if relu:
        ref = F.relu(ref)
    g = _t(25, name + 'g', tuple(ref.shape))
    ins = (x, gamma, beta) + ((res,) if with_res else ())
    rgr = torch.autograd.grad(ref, ins, g)
    xd, gd, bd = (t.detach().to(hip_device).requires_grad_(True) for t in (x, gamma, beta))
    rd = res.detach().to(hip_device).requires_grad_(True) if with_res else None

    # GroupNorm on its own
    if not with_res:
        out = _ops.group_norm(xd, gd, bd, relu=relu)
        gr = torch.autograd.grad(out, (xd, gd, bd), g.to(hip_device))
        e = dict(out=max_err(out, ref), dx=rel_err(gr[0], rgr[0]), dgamma=rel_err(gr[1], rgr[1]), dbeta=rel_err(gr[2], rgr[2]))
        report(name + '_gn_only', **e)
        assert e['out'] < 1e-4 and e['dx'] < 2e-4 and e['dgamma'] < 2e-4 and e['dbeta'] < 2e-4, e

    # fused unit with an identity 1x1x1 conv in front: exercises stats, apply, residual, and dbias through GN backward
    w = torch.eye(C).reshape(C, C, 1, 1, 1).contiguous().to(hip_device).requires_grad_(True)
    cb = torch.zeros(C, device=hip_device, requires_grad=True)
    out = _ops.conv_gn_act(xd, w, cb, gd, bd, residual=rd, kind='k1', relu=relu)
    outs = torch.autograd.grad(out, (xd, gd, bd, cb) + ((rd,) if with_res else ()), g.to(hip_device))
    e = dict(out=max_err(out, ref), dx=rel_err(outs[0], rgr[0]), dgamma=rel_err(outs[1], rgr[1]),
             dbeta=rel_err(outs[2], rgr[2]))
    # dbias of the conv feeding a GroupNorm: sum over voxels of dL/dy
    xr = x.detach().clone().requires_grad_(True)
    bz = torch.zeros(C, requires_grad=True)
    r2 = F.group_norm(xr + bz.view(1, C, 1, 1, 1), 1, gamma.detach(), beta.detach(), 1e-5)
    if with_res:
        r2 = r2 + res.detach()
    if relu:
        r2 = F.relu(r2)
    (rdb,) = torch.autograd.grad(r2, (bz,), g)
    e['dbias_abs'] = max_err(outs[3], rdb)
    if with_res:
        e['dres'] = rel_err(outs[4], rgr[3])
    report(name + '_fused', **e)
    assert e['out'] < 1e-4 and e['dx'] < 2e-4 and e['dgamma'] < 2e-4 and e['dbeta'] < 2e-4, e
    assert e['dbias_abs'] < 1e-3 + 2e-4 * float(rdb.abs().max()), e
    if with_res:
        assert e['dres'] < 1e-5, e


@pytest.mark.parametrize('C', [2, 5, 16])
def test_softmax_and_cat(hip_device, C):
    from segmentation3d import _ops
    x = _t(31, 'smx{}'.format(C), (2, C, 4, 6, 8), std=2.0).requires_grad_(True)
    g = _t(32, 'smg{}'.format(C), (2, C, 4, 6, 8))
    ref = F.softmax(x, dim=1)
    (rdx,) = torch.autograd.grad(ref, (x,), g)
    xd = x.detach().to(hip_device).requires_grad_(True)
    out = _ops.softmax_channels(xd)
    assert out.is_contiguous()
    (dx,) = torch.autograd.grad(out, (xd,), g.to(hip_device))
    e = dict(out=max_err(out, ref), dx=rel_err(dx, rdx))
    report('softmax_{}'.format(C), **e)
    assert e['out'] < 1e-6 and e['dx'] < 1e-5
    a = _t(33, 'cata', (2, C, 4, 4, 4)).to(hip_device).requires_grad_(True)
    b = _t(34, 'catb', (2, 3 * C, 4, 4, 4)).to(hip_device).requires_grad_(True)
    out = _ops.cat_channels(a, b)
    assert torch.equal(out.cpu(), torch.cat((a.detach().cpu(), b.detach().cpu()), 1))
    gg = _t(35, 'catg', tuple(out.shape)).to(hip_device)
    da, db = torch.autograd.grad(out, (a, b), gg)
    assert torch.equal(da.cpu(), gg[:, :C].cpu()) and torch.equal(db.cpu(), gg[:, C:].cpu())


def test_layout_roundtrip(hip_device):
    from segmentation3d import _ops
    x = _t(36, 'lay', (2, 5, 3, 4, 6)).to(hip_device)
    n = _ops.to_ndhwc(x)
    assert torch.equal(n.cpu(), x.cpu().permute(0, 2, 3, 4, 1).contiguous())
    back = _ops.to_ncdhw_contiguous(n)
    assert torch.equal(back.cpu(), x.cpu())


@pytest.mark.parametrize('C,shape', [(2, (2, 2, 16, 16, 20)), (5, (1, 5, 8, 12, 12)), (3, (3, 3, 4, 4, 4))])
def test_losses_against_oracle(hip_device, C, shape):
    from segmentation3d.loss.multi_dice_loss import MultiDiceLoss
    from segmentation3d.loss.focal_loss import FocalLoss
    logits = _t(41, 'll{}'.format(C), shape, std=2.0)
    probs = F.softmax(logits, dim=1)
    target = torch.from_numpy(detgen.labels(42, 'lt{}'.format(C), (shape[0], 1) + shape[2:], C))
    w = [1.0 + 0.5 * i for i in range(C)]
    p = probs.clone().requires_grad_(True)
    ref = torch_ref.multi_dice_loss(p, target, w)
    ref.backward()
    pd = probs.to(hip_device).requires_grad_(True)
    out = MultiDiceLoss(w, C, use_gpu=True)(pd, target.to(hip_device))
    out.backward()
    e = dict(loss=abs(float(out) - float(ref)), grad=rel_err(pd.grad, p.grad))
    report('dice_{}'.format(C), **e)
    assert e['loss'] < 1e-5 and e['grad'] < 1e-4, e
    for gamma, alpha, avg in ((2, None, True), (0, None, True), (1.5, w, True), (2, None, False)):
        p = probs.clone().requires_grad_(True)
        ref = torch_ref.focal_loss(p, target, C, alpha, gamma, avg)
        ref.backward()
        pd = probs.to(hip_device).requires_grad_(True)
        out = FocalLoss(C, alpha=alpha, gamma=gamma, size_average=avg, use_gpu=True)(pd, target.to(hip_device))
        out.backward()
        e = dict(loss=abs(float(out) - float(ref)) / max(1.0, abs(float(ref))), grad=rel_err(pd.grad, p.grad))
        report('focal_{}_{}_{}'.format(C, gamma, int(avg)), **e)
        assert e['loss'] < 1e-5 and e['grad'] < 1e-4, e
    # 2-D [sample, class] form (focal_loss.py:32)
    p2 = probs.movedim(1, -1).reshape(-1, C).contiguous()
    ref = torch_ref.focal_loss(p2, target.reshape(-1), C, None, 2)
    out = FocalLoss(C, use_gpu=True)(p2.to(hip_device), target.reshape(-1).to(hip_device))
    assert abs(float(out) - float(ref)) < 1e-5


def test_losses_golden_ties(hip_device):
    """threshold / tie cases pinned by the reference itself (tests/golden/loss_ties2, loss_allbg4)"""
    from conftest import golden_npz
    from segmentation3d.loss.multi_dice_loss import MultiDiceLoss
    from segmentation3d.loss.focal_loss import FocalLoss
    for name in ('ties2', 'allbg4', 'rand2', 'rand5'):
        gold = golden_npz('loss_' + name)
        C = gold['probs'].shape[1]
        t = torch.from_numpy(gold['target']).to(hip_device)
        for wname, w in (('uniform', [1.0] * C), ('ramp', [1.0 + i for i in range(C)])):
            p = torch.from_numpy(gold['probs']).to(hip_device).requires_grad_(True)
            loss = MultiDiceLoss(w, C, use_gpu=True)(p, t)
            loss.backward()
            assert abs(float(loss) - float(gold['dice_' + wname])) < 1e-6, (name, wname)
            assert max_err(p.grad, gold['dice_{}_grad'.format(wname)]) < 1e-6 * max(1.0, float(np.abs(gold['dice_{}_grad'.format(wname)]).max()) * 1e2)
        for gname, gamma, alpha in (('g2', 2, None), ('g0', 0, None), ('g1p5_alpha', 1.5, [1.0 + i for i in range(C)])):
            p = torch.from_numpy(gold['probs']).to(hip_device).requires_grad_(True)
            loss = FocalLoss(C, alpha=alpha, gamma=gamma, use_gpu=True)(p, t)
            loss.backward()
            assert abs(float(loss) - float(gold['focal_' + gname])) < 1e-6, (name, gname)
            assert rel_err(p.grad, gold['focal_{}_grad'.format(gname)]) < 1e-5


def test_small_losses_match_reference_fixtures(hip_device):
    """BinaryDiceLoss on its own (HIP kernels seg3d_binary_dice_*) and the CrossEntropyLoss plugin (stock torch op on the
    device, double soft-max quirk kept) against values + gradients produced by the reference's own modules"""
    from conftest import golden_npz
    from segmentation3d.loss.binary_dice_loss import BinaryDiceLoss
    from segmentation3d.loss.cross_entropy_loss import CrossEntropyLoss
    gold = golden_npz('small_losses')
    for name in ('b2', 'b3_ties'):
        p = torch.from_numpy(gold['bdice_{}/probs'.format(name)]).to(hip_device).requires_grad_(True)
        t = torch.from_numpy(gold['bdice_{}/target'.format(name)]).to(hip_device)
        loss = BinaryDiceLoss()(p, t)
        loss.backward()
        e = dict(loss=abs(float(loss) - float(gold['bdice_{}/loss'.format(name)])),
                 grad=max_err(p.grad, gold['bdice_{}/grad'.format(name)]))
        report('binary_dice_' + name, **e)
        assert e['loss'] < 1e-6 and e['grad'] < 1e-7, e
    with pytest.raises(ValueError):
        BinaryDiceLoss()(torch.zeros(1, 3, 4, 4, 4, device=hip_device), torch.zeros(1, 1, 4, 4, 4, device=hip_device))
    for C in (2, 5):
        p = torch.from_numpy(gold['ce{}/probs'.format(C)]).to(hip_device).requires_grad_(True)
        loss = CrossEntropyLoss()(p, torch.from_numpy(gold['ce{}/target'.format(C)]).to(hip_device))
        loss.backward()
        e = dict(loss=abs(float(loss) - float(gold['ce{}/loss'.format(C)])), grad=max_err(p.grad, gold['ce{}/grad'.format(C)]))
        report('cross_entropy_on_probs_{}'.format(C), **e)
        assert e['loss'] < 1e-5 and e['grad'] < 1e-6, e


def test_patch_kernels_against_reference_fixtures(hip_device):
    """seg3d_patch_gather_normalize against the reference's OWN FixedNormalizer / AdaptiveNormalizer outputs and
    seg3d_patch_scatter_accumulate against its add_image_region / add_image_value loop (tests/golden/normalizers.npz,
    produced by executing the reference's function bodies)"""
    from conftest import golden_npz
    from segmentation3d.core.seg_infer import SlidingWindowBatcher
    gold = golden_npz('normalizers')
    for name, roi, kind, params in detgen.normalizer_cases():
        d = {'type': 0, 'mean': params['mean'], 'stddev': params['stddev'], 'clip': params['clip']} if kind == 'fixed' \
            else {'type': 1, 'clip_sigma': params['clip_sigma']}
        Z, Y, X = roi.shape
        # the ROI sits inside a larger volume (other voxels must not leak into the adaptive statistics)
        vol = np.full((Z + 3, Y + 5, X + 2), 1e4, dtype=np.float32)
        vol[2:2 + Z, 1:1 + Y, 1:1 + X] = roi
        b = SlidingWindowBatcher(torch.from_numpy(vol).to(hip_device), [[1, 1, 2]], (X, Y, Z), 2, d, max_batch=2)
        out = b.gather([0])[0, 0]
        want = gold['norm/' + name]
        scale = max(1.0, float(np.abs(want).max()))
        e = max_err(out, want)
        report('gather_normalize_' + name, err=e)
        assert e <= 4e-6 * scale, (name, e)
        assert float(out.min()) >= float(want.min()) and float(out.max()) <= float(want.max())   # clip bounds exact
    vol_shape, patches = detgen.accumulate_case()
    box = tuple(patches[0][1][d] - patches[0][0][d] for d in range(3))
    b = SlidingWindowBatcher(torch.zeros(vol_shape, device=hip_device), [p[0] for p in patches], box, 3, None, max_batch=4)
    for i in range(0, len(patches), 4):
        idx = list(range(i, min(i + 4, len(patches))))
        probs = torch.stack([torch.stack([torch.from_numpy(detgen.uniform(71, 'acc/p{}c{}'.format(k, c), box[::-1]).astype(np.float32))
                                          for c in range(3)]) for k in idx]).to(hip_device)
        b.scatter(idx, probs)
    assert np.array_equal(b.acc.cpu().numpy(), gold['acc/sum'])            # list order, no atomics: bit-identical
    assert np.array_equal(b.count.cpu().numpy(), gold['acc/count'])
    # finalize on a z-slab (what a rank of the sharded sliding window does) and the zero-count rule
    ref_p, ref_m = numpy_ref.finalize(gold['acc/sum'].copy(), gold['acc/count'])
    b.count[:2] = 0
    b.acc[:, :2] = 0
    probs_d, mask_d = b.finalize((0, 9))
    got = probs_d.cpu().numpy()
    assert np.isfinite(got).all() and (got[:, :2] == 0).all() and (mask_d[:2] == 0).all()
    assert np.array_equal(got[:, 2:9], ref_p[:, 2:9]) and np.array_equal(mask_d[2:9].cpu().numpy(), ref_m[2:9])
    assert np.array_equal(got[:, 9:], gold['acc/sum'][:, 9:]) and (mask_d[9:] == 0).all()    # outside the slab: untouched


def test_fused_adam_matches_torch_adam(hip_device):
    from segmentation3d.optim.fused_adam import FusedAdam
    shapes = [(16, 1, 3, 3, 3), (16,), (33,), (5, 7), (1,)]
    ps_ref = [_t(51, 'ap{}'.format(i), s).requires_grad_(True) for i, s in enumerate(shapes)]
    ps_dev = [torch.nn.Parameter(p.detach().clone().to(hip_device)) for p in ps_ref]
    ref = torch.optim.Adam(ps_ref, lr=1e-3, betas=(0.9, 0.999))
    opt = FusedAdam(ps_dev, lr=1e-3, betas=(0.9, 0.999))
    for step in range(5):
        ref.zero_grad()
        opt.zero_grad()
        for i, (a, b) in enumerate(zip(ps_ref, ps_dev)):
            g = _t(52 + step, 'ag{}'.format(i), shapes[i])
            a.grad = g.clone()
            b.grad.copy_(g.to(hip_device))
        ref.step()
        opt.step()
    err = max(max_err(b, a) for a, b in zip(ps_ref, ps_dev))
    report('fused_adam', err=err)
    assert err < 1e-6
    sd = opt.state_dict()
    assert sorted(sd['state'][0].keys()) == ['exp_avg', 'exp_avg_sq', 'step']
    assert max_err(sd['state'][0]['exp_avg'], ref.state_dict()['state'][0]['exp_avg']) < 1e-6


def test_patch_batcher_against_numpy_oracle(hip_device):
    from segmentation3d.core.seg_infer import SlidingWindowBatcher
    rng = np.random.RandomState(0)
    Z, Y, X = 48, 64, 80
    vol = (rng.randn(Z, Y, X) * 300 - 200).astype(np.float32)
    starts, ends = numpy_ref.partition_by_fixed_size((X, Y, Z), (1.0, 1.0, 1.0), [0, 0, 0], [X, Y, Z], (32, 32, 32),
                                                     (24, 24, 16), 16)
    C = 3
    for normalizer in ({'type': 1, 'clip_sigma': 2.5}, {'type': 0, 'mean': -150.0, 'stddev': 280.0, 'clip': True}, None):
        batcher = SlidingWindowBatcher(torch.from_numpy(vol).to(hip_device), starts, (32, 32, 32), C, normalizer)
        acc = np.zeros((C, Z, Y, X), np.float32)
        cnt = np.zeros((Z, Y, X), np.float32)
        P = 5
        for i in range(0, len(starts), P):
            idx = list(range(i, min(i + P, len(starts))))
            batch = batcher.gather(idx)
            for j, k in enumerate(idx):
                s, e = starts[k], ends[k]
                roi = numpy_ref.apply_normalizer(vol[s[2]:e[2], s[1]:e[1], s[0]:e[0]].copy(), normalizer)
                assert max_err(batch[j, 0], roi) < 2e-5, normalizer
            probs = torch.softmax(torch.stack([batch[:, 0] * (c + 1) for c in range(C)], 1), 1).contiguous()
            batcher.scatter(idx, probs)
            pc = probs.cpu().numpy()
            for j, k in enumerate(idx):
                numpy_ref.accumulate_patch(acc, cnt, starts[k], ends[k], pc[j])
        probs_d, mask_d = batcher.finalize()
        rp, rm = numpy_ref.finalize(acc, cnt)
        assert np.array_equal(batcher.count.cpu().numpy(), cnt)
        assert max_err(probs_d, rp) == 0.0          # same summation order as the sequential reference loop
        assert np.array_equal(mask_d.cpu().numpy(), rm)


# ---------------------------------------------------------------------------------------------------------------------
# geometry around the patch path
# ---------------------------------------------------------------------------------------------------------------------
RESAMPLE_CASES = [
    # src size (x, y, z), src frame, dst size, dst frame
    ((14, 12, 10), ((0.8, 1.1, 2.0), (1.0, 2.0, 3.0), np.eye(3).ravel()), (16, 16, 16), ((1.0, 1.0, 1.0), (1.0, 2.0, 3.0), np.eye(3).ravel())),
    ((20, 16, 12), ((1.0, 1.0, 1.0), (0.0, 0.0, 0.0), np.eye(3).ravel()), (32, 32, 16), ((0.5, 0.5, 1.5), (0.0, 0.0, 0.0), np.eye(3).ravel())),
    ((16, 16, 16), ((1.0, 1.0, 1.0), (0.0, 0.0, 0.0), np.eye(3).ravel()), (16, 16, 16), ((1.0, 1.0, 1.0), (0.0, 0.0, 0.0), np.eye(3).ravel())),
    ((12, 18, 9), ((0.7, 0.9, 1.3), (-3.0, 4.0, 1.0), np.diag([-1.0, 1.0, 1.0]).ravel()), (11, 13, 17),
     ((0.9, 0.6, 0.8), (-9.0, 3.5, 0.5), np.diag([-1.0, 1.0, 1.0]).ravel())),
]


@pytest.mark.parametrize('interp', ['LINEAR', 'NN'])
@pytest.mark.parametrize('case', RESAMPLE_CASES)
def test_resample_against_oracle(hip_device, case, interp):
    """seg3d_resample_affine (trilinear / nearest with ITK's inside test and padding) against oracle/numpy_ref"""
    from oracle import numpy_ref
    from segmentation3d.utils import image_tools
    src_size, src_frame, dst_size, dst_frame = case
    src = detgen.normal(401, 'rs/{}'.format(src_size), (src_size[2], src_size[1], src_size[0]))
    M = numpy_ref.index_affine(src_frame, dst_frame)
    ref = numpy_ref.resample_affine(src, M, dst_size, interp == 'LINEAR', pad=-2.5)
    out = image_tools.resample_device(torch.from_numpy(src).to(hip_device), src_frame, dst_size, dst_frame, interp, -2.5)
    e = max_err(out, ref)
    report('resample_{}_{}'.format(interp, 'x'.join(map(str, src_size))), err=e, padded=float((ref == -2.5).mean()))
    assert e < (1e-5 if interp == 'LINEAR' else 1e-7), e
    assert tuple(out.shape) == (dst_size[2], dst_size[1], dst_size[0])


def test_resample_spacing_and_back_host_api(hip_device):
    """resample_spacing (size rounded up to the stride multiple, zero beyond the input) and resample onto a reference
    image, through the Image3d API; identity resampling returns the input bit for bit"""
    from oracle import numpy_ref
    from segmentation3d.utils import image_tools
    from segmentation3d.utils.image3d import Image3d
    arr = detgen.normal(402, 'rs/img', (20, 30, 26))
    img = Image3d(arr, (0.8, 0.8, 1.6), (5.0, -3.0, 2.0), np.eye(3).ravel())
    iso = image_tools.resample_spacing(img, [1.0, 1.0, 1.0], 16, 'LINEAR')
    want = numpy_ref.resampled_size(img.GetSize(), img.GetSpacing(), [1.0, 1.0, 1.0], 16)
    assert list(iso.GetSize()) == want and all(v % 16 == 0 for v in want)
    M = numpy_ref.index_affine((img.GetSpacing(), img.GetOrigin(), img.GetDirection()), ([1.0] * 3, img.GetOrigin(), img.GetDirection()))
    assert max_err(iso.array, numpy_ref.resample_affine(arr, M, want, True, 0.0)) < 1e-5
    back = image_tools.resample(iso, img, 'LINEAR', 1.0)
    assert back.array.shape == arr.shape
    same = image_tools.resample(img, img, 'LINEAR', 0.0)
    assert np.array_equal(same.array, arr)


def test_connected_components_and_bounding_box(hip_device):
    """26-connected component selection (largest with the raster-order tie rule, size threshold, multi-label
    composition) and the bounding box, bit-exact against the scipy-based oracle"""
    from oracle import numpy_ref
    from segmentation3d.utils import image_tools
    shape = (28, 36, 44)
    u = detgen.uniform(403, 'cc/u', tuple((s + 3) // 4 for s in shape))
    blobs = np.zeros(u.shape, dtype=np.int8)
    blobs[u > 0.80] = 1
    blobs[u < 0.12] = 2
    blobs[(u > 0.45) & (u < 0.50)] = 3
    mask = np.repeat(np.repeat(np.repeat(blobs, 4, 0), 4, 1), 4, 2)[:shape[0], :shape[1], :shape[2]].copy()
    mask[0, 0, 0:3] = 1                       # small stray components
    mask[27, 35, 41:44] = 2
    mask[5, 5, 5] = 3                         # diagonal neighbours join under 26-connectivity only
    mask[6, 6, 6] = 3
    md = torch.from_numpy(mask).to(hip_device)
    for labels in ([1, 2, 3], [2], [3, 1], [4]):
        got = image_tools.connected_component_filter_device(md, labels, 'largest').cpu().numpy()
        assert np.array_equal(got, numpy_ref.connected_component_filter(mask, labels, 'largest')), labels
        for thr in (2, 30, 100000):
            got = image_tools.connected_component_filter_device(md, labels, 'min_size', thr).cpu().numpy()
            assert np.array_equal(got, numpy_ref.connected_component_filter(mask, labels, 'min_size', thr)), (labels, thr)
    # tie: two components of equal size -> the one met first in raster order
    tie = np.zeros((6, 6, 6), dtype=np.int8)
    tie[4, 4, 0:3] = 1
    tie[1, 1, 2:5] = 1
    got = image_tools.connected_component_filter_device(torch.from_numpy(tie).to(hip_device), [1], 'largest').cpu().numpy()
    assert np.array_equal(got, numpy_ref.connected_component_filter(tie, [1], 'largest')) and got[1, 1, 3] == 1 and got[4, 4, 1] == 0
    for sel in (None, [2], [1, 3], [9]):
        assert image_tools.get_bounding_box_device(md, sel) == numpy_ref.get_bounding_box(mask, sel), sel


def test_winograd_kernels_random_shapes_against_direct_kernels(hip_device):
    """40 random supported shapes (batch 1-5, extents in whole tiles, channel counts incl. partial 32-blocks, item counts that
    are / are not multiples of 8 or exceed the workgroup count): the F(2x2,3x3) / F(2,3) forward kernels and the F(3x3,2x2) /
    F(3,2) weight gradients against the direct MFMA kernels on the same device tensors (relative error of the difference)"""
    from segmentation3d import _engine as E
    g = torch.Generator().manual_seed(2024)

    def ri(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=g))
    worst = {}
    for case in range(40):
        N = ri(1, 5)
        D, H, W = 8 * ri(1, 3), 8 * ri(1, 4), 8 * ri(1, 5)
        Cin, Cout = 8 * ri(1, 9), 32 * ri(1, 3)
        x = torch.randn((N, D, H, W, Cin), generator=g).to(hip_device)
        w = (torch.randn((Cout, Cin, 3, 3, 3), generator=g) * (2.0 / (Cin * 27)) ** 0.5).to(hip_device)
        b = torch.randn((Cout,), generator=g).to(hip_device)
        dy = torch.randn((N, D, H, W, Cout), generator=g).to(hip_device)
        wp27 = torch.empty((E.query('seg3d_packed_mfma_floats', Cin, Cout, 27),), device=hip_device)
        E.call('seg3d_pack_weights_mfma', E.ptr(w), E.ptr(wp27), Cin, Cout, 27, 27, Cin * 27, 0, E.stream_ptr())
        yd = torch.empty((N, D, H, W, Cout), device=hip_device)
        ws = torch.empty((max(1, E.query('seg3d_conv3d_k3_mfma_fwd_workspace_floats', N, D, H, W, Cin, Cout)),), device=hip_device)
        E.call('seg3d_conv3d_k3_mfma_fwd', E.ptr(x), E.ptr(wp27), E.ptr(b), None, E.ptr(yd), None, E.ptr(ws), N, D, H, W, Cin, Cout,
               E.stream_ptr())
        for form, T in (('wino', 36), ('wino2d', 48)):
            assert E.query('seg3d_conv3d_k3_{}_supported'.format(form), N, D, H, W, Cin, Cout) == 1
            wp = torch.empty((E.query('seg3d_packed_mfma_floats', Cin, Cout, T),), device=hip_device)
            E.call('seg3d_pack_weights_mfma', E.ptr(w), E.ptr(wp), Cin, Cout, T, 27, Cin * 27, 0, E.stream_ptr())
            y = torch.full((N, D, H, W, Cout), float('nan'), device=hip_device)
            E.call('seg3d_conv3d_k3_{}_fwd'.format(form), E.ptr(x), E.ptr(wp), E.ptr(b), None, E.ptr(y), None, N, D, H, W, Cin, Cout,
                   E.stream_ptr())
            e = float((y - yd).abs().max() / yd.abs().max())
            worst[form] = max(worst.get(form, 0.0), e)
            assert e < 2e-5, (form, case, (N, D, H, W, Cin, Cout), e)
        dwd = torch.empty((Cout, Cin, 3, 3, 3), device=hip_device)
        wsd = torch.empty((E.query('seg3d_conv3d_k3_mfma_wgrad_workspace_floats', N, D, H, W, Cin, Cout),), device=hip_device)
        E.call('seg3d_conv3d_k3_mfma_wgrad', E.ptr(x), E.ptr(dy), E.ptr(dwd), E.ptr(wsd), N, D, H, W, Cin, Cout, 0, E.stream_ptr())
        for form in ('wino', 'wino2d'):
            dw = torch.full((Cout, Cin, 3, 3, 3), float('nan'), device=hip_device)
            wsw = torch.empty((E.query('seg3d_conv3d_k3_{}_wgrad_workspace_floats'.format(form), N, D, H, W, Cin, Cout),), device=hip_device)
            E.call('seg3d_conv3d_k3_{}_wgrad'.format(form), E.ptr(x), E.ptr(dy), E.ptr(dw), E.ptr(wsw), N, D, H, W, Cin, Cout, 0,
                   E.stream_ptr())
            e = float((dw - dwd).abs().max() / dwd.abs().max())
            worst[form + '_wgrad'] = max(worst.get(form + '_wgrad', 0.0), e)
            assert e < 2e-5, (form + '_wgrad', case, (N, D, H, W, Cin, Cout), e)
    report('winograd_random_shapes_vs_direct', **worst)


def test_channel_slice_inputs_equal_their_packed_copies(hip_device):
    """A logical NCDHW tensor that is a CHANNEL SLICE of a wider NDHWC buffer (what UpCatFunction.backward hands back as the skip
    gradient when no link is set, what an inference forward keeps in a decoder's concatenated buffer) must give every operator the
    same result as its packed copy: only the stride-2 conv of a no-grad forward and up_cat's skip read a slice in place
    (_ops.to_ndhwc(allow_slice=True)), all others repack it.  Forward and gradients, bit for bit."""
    from segmentation3d import _ops
    N, D, H, W, Ca, Cb = 2, 8, 8, 16, 16, 32
    wide = _t(71, 'slice/wide', (N, D, H, W, Ca + Cb)).to(hip_device)
    sl = _ops.from_ndhwc(wide[..., Ca:])                     # [N, Cb, D, H, W], rows of Cb floats at stride Ca + Cb
    packed = _ops.from_ndhwc(wide[..., Ca:].contiguous())
    assert _ops._is_channel_slice(sl.permute(0, 2, 3, 4, 1)) and not sl.permute(0, 2, 3, 4, 1).is_contiguous()
    assert _ops.to_ndhwc(sl).is_contiguous()                  # the default is a packed copy ...
    assert not _ops.to_ndhwc(sl, allow_slice=True).is_contiguous()   # ... the in-place view only on request

    def both(fn, params, pair=None):
        outs = []
        for src in (pair or (sl, packed)):
            x = src.detach().requires_grad_(True)
            ps = [p.detach().clone().requires_grad_(True) for p in params]
            y = fn(x, *ps)
            g = torch.autograd.grad(y, [x] + ps, torch.ones_like(y) * 0.5 + y.detach() * 0.25)
            outs.append([y.detach()] + [t.contiguous() for t in g])
        for a, b in zip(*outs):
            assert a.shape == b.shape and torch.equal(a.contiguous(), b.contiguous())

    w3 = _t(72, 'slice/w3', (Cb, Cb, 3, 3, 3), std=0.05).to(hip_device)
    w2 = _t(73, 'slice/w2', (2 * Cb, Cb, 2, 2, 2), std=0.1).to(hip_device)
    b3, b2 = _t(74, 'slice/b3', (Cb,)).to(hip_device), _t(75, 'slice/b2', (2 * Cb,)).to(hip_device)
    g3, be3 = _t(76, 'slice/g', (Cb,)).to(hip_device), _t(77, 'slice/be', (Cb,)).to(hip_device)
    g2, be2 = _t(78, 'slice/g2', (2 * Cb,)).to(hip_device), _t(79, 'slice/be2', (2 * Cb,)).to(hip_device)
    other = _ops.from_ndhwc(_t(80, 'slice/other', (N, D, H, W, Ca)).to(hip_device))
    both(lambda x, w, b: _ops.conv(x, w, b, 'k3'), [w3, b3])
    both(lambda x, w, b: _ops.conv(x, w, b, 'k2s2'), [w2, b2])
    both(lambda x, g, b: _ops.group_norm(x, g, b, relu=True), [g3, be3])
    both(lambda x, w, b, g, be: _ops.conv_gn_act(x, w, b, g, be, kind='k3'), [w3, b3, g3, be3])
    both(lambda x, w, b, g, be: _ops.conv_gn_act(x, w, b, g, be, kind='k2s2'), [w2, b2, g2, be2])   # training: the slice is repacked
    both(lambda x, w, b, g, be: _ops.conv_gn_act(x, w, b, g, be, residual=x, kind='k3'), [w3, b3, g3, be3])
    both(lambda x: _ops.cat_channels(other, x), [])
    both(lambda x: _ops.cat_channels(x, other), [])
    sl8 = _ops.from_ndhwc(wide[..., Ca:Ca + 8])              # (the softmax kernel takes up to 16 classes)
    both(lambda x: _ops.softmax_channels(x), [], pair=(sl8, _ops.from_ndhwc(wide[..., Ca:Ca + 8].contiguous())))
    with torch.no_grad():                                     # inference: the stride-2 conv reads the slice in place
        a = _ops.conv_gn_act(sl, w2, b2, g2, be2, kind='k2s2')
        b = _ops.conv_gn_act(packed, w2, b2, g2, be2, kind='k2s2')
    assert torch.equal(a, b)


def test_out_slot_is_refused_when_a_gradient_is_wanted(hip_device):
    """out_slot writes a unit's output into a slice of another buffer and saves nothing a backward could use: inference only"""
    from segmentation3d import _ops
    N, D, H, W, C = 1, 4, 4, 8, 16
    x = _ops.from_ndhwc(_t(81, 'slot/x', (N, D, H, W, C)).to(hip_device))
    w = _t(82, 'slot/w', (C, C, 3, 3, 3), std=0.05).to(hip_device).requires_grad_(True)
    b, g, be = (torch.zeros(C, device=hip_device), torch.ones(C, device=hip_device), torch.zeros(C, device=hip_device))
    buf = torch.zeros((N, D, H, W, 2 * C), device=hip_device)
    with pytest.raises(RuntimeError):
        _ops.conv_gn_act(x, w, b, g, be, kind='k3', out_slot=buf[..., C:])
    with torch.no_grad():
        out = _ops.conv_gn_act(x, w, b, g, be, kind='k3', out_slot=buf[..., C:])
        ref = _ops.conv_gn_act(x, w, b, g, be, kind='k3')
    assert torch.equal(out, ref) and torch.equal(buf[..., C:], ref.permute(0, 2, 3, 4, 1))
