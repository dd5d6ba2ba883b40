"""bf16 mode (BASELINE config 5: bf16 activations / weights, fp32 accumulation, fp32 GroupNorm statistics, fp32 master
weights).  bf16 cannot meet the 1e-4 fp32 bar; the tolerances here are stated per test:
  * a fused unit against the EXACT arithmetic of its bf16-rounded operands (double precision on the CPU): the only
    differences left are fp32 accumulation order and bf16 output rounding (<= 1 bf16 ulp = 2^-8 relative);
  * whole networks against the fp32 engine (itself pinned to the reference fixtures at 1e-4 in test_gpu_parity.py) and
    against the reference fixtures directly.
"""
import importlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden_npz
from gpu_util import report, max_err, mask_flips, rel_err
from oracle import detgen

pytestmark = pytest.mark.gpu


def _load(module, seed):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = detgen.state_dict_like(shapes, seed)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return sd


def _bf(t):
    return t.bfloat16().double()


@pytest.mark.parametrize('kind,cin,cout,shape,residual', [
    ('k3', 32, 32, (2, 8, 8, 16), False), ('k3', 16, 16, (1, 6, 10, 12), True), ('k3', 64, 16, (1, 4, 6, 6), False),
    ('k2s2', 16, 32, (2, 8, 8, 8), False), ('convT', 64, 16, (1, 4, 4, 6), False)])
@pytest.mark.parametrize('y16', [True, False])
def test_fused_unit_bf16_exact_operands(hip_device, kind, cin, cout, shape, residual, y16):
    """out = bf16(relu(GN(conv(x_bf16, w) + b) [+ res_bf16])): equals the double-precision evaluation of the same
    expression on the bf16-rounded operands up to one bf16 rounding of the result; input / weight gradients equal the
    double-precision gradients of that expression with the same roundings of the intermediate gradient (dy -> bf16).
    y16: the conv output itself is stored as bf16 (rounded after the fp32 statistics were taken) -- the default -- or fp32"""
    from segmentation3d import _ops
    N, D, H, W = shape
    x = torch.from_numpy(detgen.normal(71, 'u/x', (N, cin, D, H, W))).bfloat16()
    wshape = (cin, cout, 2, 2, 2) if kind == 'convT' else ((cout, cin, 3, 3, 3) if kind == 'k3' else (cout, cin, 2, 2, 2))
    w = torch.from_numpy(detgen.normal(72, 'u/w', wshape, std=0.08))
    b = torch.from_numpy(detgen.normal(73, 'u/b', (cout,), std=0.3))
    g = torch.from_numpy(detgen.normal(74, 'u/g', (cout,), std=0.3)) + 1.0
    be = torch.from_numpy(detgen.normal(75, 'u/be', (cout,), std=0.3))
    # exact expression in double on the CPU
    xd = x.double().requires_grad_(True)
    wq = _bf(w).requires_grad_(True)   # the bf16 MFMA kernels (3x3x3 and stride-2, Cin % 16 == 0) use bf16 weight images
    if kind == 'k3':
        y = F.conv3d(xd, wq, b.double(), padding=1)
    elif kind == 'k2s2':
        y = F.conv3d(xd, wq, b.double(), stride=2)
    else:
        y = F.conv_transpose3d(xd, wq, b.double(), stride=2)
    res = None
    if residual:
        res = torch.from_numpy(detgen.normal(76, 'u/r', tuple(y.shape))).bfloat16()
    # GroupNorm(1, C) with the statistics of the exact y; with y16 the normalised tensor is bf16(y) (straight-through)
    dims = (1, 2, 3, 4)
    mu = y.mean(dims, keepdim=True)
    rstd = 1.0 / torch.sqrt(y.var(dims, unbiased=False, keepdim=True) + 1e-5)
    y_used = y + (y.bfloat16().double() - y).detach() if y16 else y
    o = (y_used - mu) * rstd * g.double().view(1, -1, 1, 1, 1) + be.double().view(1, -1, 1, 1, 1)
    if res is not None:
        o = o + res.double()
    o = F.relu(o)
    dout = torch.from_numpy(detgen.normal(77, 'u/do', tuple(o.shape))).bfloat16()
    o.backward(dout.double())

    dev = hip_device
    y_was = _ops.BF16_CONV_OUTPUT
    _ops.BF16_CONV_OUTPUT = y16
    with _ops.activation_dtype('bf16'):
        xg = x.to(dev).permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3).requires_grad_(True)   # NDHWC memory
        wg = w.to(dev).requires_grad_(True)
        bg, gg, beg = (t.to(dev).requires_grad_(True) for t in (b, g, be))
        rg = None if res is None else res.to(dev).permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3)
        out = _ops.conv_gn_act(xg, wg, bg, gg, beg, residual=rg, kind=kind, relu=True)
        assert out.dtype == torch.bfloat16
        out.backward(dout.to(dev))
    _ops.BF16_CONV_OUTPUT = y_was
    torch.cuda.synchronize()
    ref = o.detach()
    got = out.detach().double().cpu()
    scale = float(ref.abs().max())
    e_out = float((got - ref).abs().max()) / scale
    # gradients: dy is rounded to bf16 before the dgrad / wgrad kernels -> relative error ~2^-9 per element, averaged
    # down in the sums; compare in relative L2
    e_dx = rel_err(xg.grad.double().cpu(), xd.grad)
    e_dw = rel_err(wg.grad.double().cpu(), wq.grad)
    report('bf16_unit_{}_{}_{}{}{}'.format(kind, cin, cout, '_res' if residual else '', '_y16' if y16 else ''),
           out_rel_max=e_out, dx_rel=e_dx, dw_rel=e_dw)
    assert e_out <= 2.0 ** -8 * 1.01 + 1e-6, e_out          # one bf16 rounding of the result
    assert e_dx < 5e-3 and e_dw < 5e-3, (e_dx, e_dw)


@pytest.mark.parametrize('plugin,cin,ncls', [('vnet', 1, 2), ('vnet', 4, 4), ('vbnet', 1, 2)])
def test_network_bf16_vs_fp32(hip_device, plugin, cin, ncls):
    """whole net at 32^3, forward + Dice loss + backward: bf16 mode against the fp32 engine on identical inputs and
    against the reference fixture.  Tolerances (bf16 has 8 significant bits): probabilities max 3e-2 / mean 3e-3,
    loss 3e-3; per-parameter gradient relative L2 error: median within 1.5x the yardstick described below, cosine > 0.8"""
    from segmentation3d import _ops
    from segmentation3d.loss.multi_dice_loss import MultiDiceLoss
    tag = '{}_{}_{}'.format(plugin, cin, ncls)
    gold = golden_npz('net_' + tag)
    mod = importlib.import_module('segmentation3d.network.' + plugin)
    net = mod.SegmentationNet(cin, ncls)
    _load(net, 21)
    net = net.to(hip_device)
    x = torch.from_numpy(detgen.normal(22, tag + '/x', (1, cin, 32, 32, 32))).to(hip_device)
    t = torch.from_numpy(detgen.labels(23, tag + '/t', (1, 1, 32, 32, 32), ncls)).to(hip_device)
    loss_fn = MultiDiceLoss([1.0 + 0.5 * i for i in range(ncls)], ncls, use_gpu=True)
    res = {}
    for mode in ('fp32', 'bf16'):
        with _ops.activation_dtype(mode):
            net.zero_grad()
            probs = net(x)
            loss = loss_fn(probs, t)
            loss.backward()
        torch.cuda.synchronize()
        res[mode] = (probs.detach().clone(), float(loss), {k: p.grad.detach().clone() for k, p in net.named_parameters()})
    # yardstick for the gradient comparison: this random-init net is ill-conditioned (in fp32 a 1e-7 rounding difference
    # already moves gradients by ~1e-2, test_gpu_parity.py), so the bf16 error is set against the change the fp32 engine
    # itself shows when only its INPUT and conv weights are rounded to bf16 once (no rounding anywhere inside)
    saved = {k: p.detach().clone() for k, p in net.named_parameters()}
    with torch.no_grad():
        for k, p in net.named_parameters():
            if p.dim() == 5:
                p.copy_(p.bfloat16().float())
    from segmentation3d import _ops as _o
    _o.PACK_CACHE.invalidate()
    net.zero_grad()
    loss_fn(net(x.bfloat16().float()), t).backward()
    torch.cuda.synchronize()
    gyard = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    with torch.no_grad():
        for k, p in net.named_parameters():
            p.copy_(saved[k])
    _o.PACK_CACHE.invalidate()
    p32, l32, g32 = res['fp32']
    p16, l16, g16 = res['bf16']
    yard = np.array([float((gyard[k] - g32[k]).double().norm() / (g32[k].double().norm() + 1e-30)) for k in g32])
    assert p16.dtype == torch.float32 and tuple(p16.shape) == (1, ncls, 32, 32, 32)
    gerr = np.array([float((g16[k] - g32[k]).double().norm() / (g32[k].double().norm() + 1e-30)) for k in g32])
    gcos = np.array([float((g16[k].double() * g32[k].double()).sum() /
                           (g16[k].double().norm() * g32[k].double().norm() + 1e-30)) for k in g32])
    worst = sorted(zip(gerr, g32.keys()))[-3:]
    print('worst gradient tensors:', worst)
    e = dict(probs_max=max_err(p16, p32), probs_mean=float((p16 - p32).abs().mean()), loss=abs(l16 - l32),
             probs_vs_fixture=max_err(p16, gold['probs']), loss_vs_fixture=abs(l16 - float(gold['loss_dice'])),
             grad_rel_median=float(np.median(gerr)), grad_rel_max=float(gerr.max()),
             grad_cos_min=float(gcos.min()), grad_cos_median=float(np.median(gcos)),
             yard_rel_median=float(np.median(yard)), yard_rel_max=float(yard.max()))
    report('bf16_net_' + tag, **e)
    assert e['probs_max'] < 3e-2 and e['probs_mean'] < 3e-3, e
    assert e['loss'] < 3e-3 and e['loss_vs_fixture'] < 3e-3, e
    assert e["grad_rel_median"] < 1.5 * e["yard_rel_median"] + 1e-2 and e["grad_cos_min"] > 0.8, e


def test_train_steps_bf16_track_fp32(hip_device):
    """five train steps of vnet(1,2) on a [2,1,48,48,32] batch (FusedAdam with gradient sinks, packed-weight cache, bf16
    weight images refreshed by the multi-pack kernel): the bf16 loss curve stays within 5e-3 of the fp32 curve"""
    from segmentation3d import _ops
    from segmentation3d.network import vnet
    from segmentation3d.loss.multi_dice_loss import MultiDiceLoss
    from segmentation3d.optim.fused_adam import FusedAdam
    x = torch.from_numpy(detgen.normal(61, 'ts/x', (2, 1, 48, 48, 32))).to(hip_device)
    t = torch.from_numpy(detgen.labels(62, 'ts/t', (2, 1, 48, 48, 32), 2)).to(hip_device)
    curves = {}
    prev_cache = _ops.weight_cache(True)
    try:
        for mode in ('fp32', 'bf16'):
            net = vnet.SegmentationNet(1, 2)
            _load(net, 21)
            net = net.to(hip_device)
            _ops.PACK_CACHE.invalidate()
            opt = FusedAdam(net.parameters(), lr=1e-3, betas=(0.9, 0.999), direct_grads=True)
            loss_fn = MultiDiceLoss([0.5, 0.5], 2, use_gpu=True)
            losses = []
            with _ops.activation_dtype(mode):
                for step in range(5):
                    opt.zero_grad()
                    loss = loss_fn(net(x), t)
                    loss.backward()
                    opt.step()
                    losses.append(float(loss))
            opt.release_grad_sinks()
            curves[mode] = losses
    finally:
        _ops.weight_cache(prev_cache)
    diffs = [abs(a - b) for a, b in zip(curves['fp32'], curves['bf16'])]
    report('bf16_train_curve', **{'fp32_{}'.format(i): v for i, v in enumerate(curves['fp32'])},
           **{'bf16_{}'.format(i): v for i, v in enumerate(curves['bf16'])})
    assert curves['bf16'][-1] < curves['bf16'][0], curves
    assert max(diffs) < 5e-3, (curves, diffs)


def test_sliding_window_bf16_vs_fp32(hip_device, tmp_path):
    """whole-volume path (device patch batcher, captured graph, two half-batch streams) in bf16 mode against the same
    model in fp32 on a 64x80x96 volume: mean probability difference < 3e-3, max < 3e-2, the arg-max masks agree on
    > 99 % of the voxels (the disagreements sit where the two class probabilities are within the bf16 error of 0.5)"""
    from segmentation3d import _ops
    from segmentation3d.network import vnet
    from segmentation3d.core.seg_infer import load_single_model, segmentation_volume
    from segmentation3d.utils.image3d import Image3d
    net = vnet.SegmentationNet(1, 2)
    _load(net, 21)
    folder = tmp_path / 'model' / 'fine'
    chk = folder / 'checkpoints' / 'chk_7'
    chk.mkdir(parents=True)
    norm = {'type': 1, 'clip_sigma': 3}
    torch.save({'epoch': 7, 'batch': 1, 'net': 'vnet', 'max_stride': 16, 'state_dict': dict(net.state_dict()),
                'spacing': [1.0, 1.0, 1.0], 'interpolation': 'LINEAR', 'in_channels': 1, 'out_channels': 2,
                'crop_normalizers': [norm]}, str(chk / 'params.pth'))
    vol = (detgen.normal(71, 'sw/vol', (64, 80, 96)) * 200 - 300).astype(np.float32)

    class Cfg(object):
        partition_type = 'SIZE'
        partition_size = [48, 48, 32]
        partition_stride = [32, 32, 16]
    res = {}
    for mode in ('fp32', 'bf16'):
        with _ops.activation_dtype(mode):
            _ops.PACK_CACHE.clear()
            model = load_single_model(str(folder), 0)
            probs, mask = segmentation_volume(model, Cfg, Image3d(vol), None, None, True, batch_size=4)
        res[mode] = (np.stack([p.array for p in probs]), mask.array.copy())
    _ops.PACK_CACHE.clear()
    d = np.abs(res['bf16'][0] - res['fp32'][0])
    # the masks are integer work: the bf16 arg-max may differ from the fp32 engine's only where the fp32 engine's two largest
    # class probabilities are closer than twice the mode's own probability bar (2 x 3e-2: each probability may move by 3e-2),
    # and the returned mask must be exactly the arg-max of the bf16 probabilities it came with (no post-processing here)
    flips, gap = mask_flips(res['bf16'][1], res['fp32'][1], res['fp32'][0])
    own = bool(np.array_equal(res['bf16'][1], np.argmax(res['bf16'][0], axis=0).astype(np.int8)))
    e = dict(probs_max=float(d.max()), probs_mean=float(d.mean()), mask_mismatch=float(np.mean(res['bf16'][1] != res['fp32'][1])),
             argmax_flips=float(flips), worst_fp32_gap_at_a_flip=gap, mask_is_own_argmax=float(own))
    report('bf16_sliding_window_64x80x96', **e)
    assert e['probs_mean'] < 3e-3 and e['probs_max'] < 3e-2, e
    assert gap < 2 * 3e-2, 'a bf16 arg-max flip at an fp32 top1 - top2 gap of {} (bar: 2 x the 3e-2 probability tolerance)'.format(gap)
    assert own, 'bf16 mask differs from the arg-max of its own probabilities'
