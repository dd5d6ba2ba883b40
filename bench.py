#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X-native segmentation engine.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

Metric (BASELINE.json): patches/sec of a 96^3, 1-modality V-Net fp32 TRAIN STEP (forward + Dice loss + backward + Adam),
batch 4 per GPU, weak scaling over the GPUs of one node; plus the whole-volume sliding-window inference seconds
(512x512x400, 96^3 patches, stride 48, hipGraph-replayed batches) reported in the `infer` object at N = 1.
A "step" = one optimisation step on one synthetic batch that is already resident in HBM.

The JSON line also carries
  roofline     : the dominant kernel (fp32 MFMA 3x3x3 convolution, forward + data-gradient launches), measured live with
                 events on the launch stream: algorithmic FLOPs of its launches / their summed duration vs 157.3 TFLOP/s
  cpu_baseline : the oracle (stock torch CPU ops, op-for-op the reference's path) timed on this box's host cores
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(REPO, 'medical-segmentation3d-toolkit_amd'), REPO):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 MFMA, dense (v_mfma_f32_32x32x16_bf16)
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=4, help='patches per GPU per step')
    ap.add_argument('--patch', type=int, default=96)
    ap.add_argument('--net', default='vnet')
    ap.add_argument('--in-channels', type=int, default=1)
    ap.add_argument('--classes', type=int, default=2)
    ap.add_argument('--loss', default='Dice', choices=['Dice', 'Focal'])
    ap.add_argument('--no-infer', action='store_true', help='skip the whole-volume inference measurement')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--infer-volume', default='512,512,400', help='X,Y,Z of the synthetic inference volume')
    ap.add_argument('--infer-batch', type=int, default=16)
    ap.add_argument('--infer-timeout', type=float, default=240.0,
                    help='N > 1: seconds the sharded inference leg may take before the ranks give up (exit code 3)')
    ap.add_argument('--kernel-report', default='', help='write the per-shape kernel timing table to this json file')
    ap.add_argument('--dtype', default='fp32', choices=['fp32', 'bf16'],
                    help="fp32 = the headline metric (BASELINE configs 2-4); bf16 = BASELINE config 5's mode (bf16 "
                         "activations / packed weights, fp32 accumulate, statistics and master weights) -- reported "
                         "under its own dtype, never as the fp32 headline")
    ap.add_argument('--graph', action='store_true', help='(default for one GPU; kept for compatibility)')
    ap.add_argument('--no-graph', action='store_true',
                    help='one GPU: issue the ~420 launches of a step eagerly instead of replaying the step from one '
                         'hipGraph (core/seg_train.TrainStep use_graph; same kernels, bit-identical losses).  With several '
                         'GPUs the step is always eager: the gradient all-reduce stays outside hipGraphs')
    ap.add_argument('--step-mode', default='auto', choices=['auto', 'graph', 'eager'],
                    help='one GPU: how the ~420 launches of a step are issued.  graph = replay of the captured hipGraph; eager '
                         '= plain launches (what every rank does with several GPUs); auto (default) = both are timed for a few '
                         'steps after the capture and the faster one runs the warm-up and the timed steps (same kernels, same '
                         'losses: on a fast host the eager launches overlap the weight-gradient side stream better, on a slow '
                         'host or with short kernels -- bf16 mode -- the replay wins)')
    ap.add_argument('--no-wgrad-overlap', action='store_true',
                    help='enqueue weight-gradient kernels on the main stream (no second HIP stream): use this under '
                         'rocprofv3 --kernel-trace, which serialises concurrent dispatches and distorts their durations')
    return ap.parse_args()


def synthetic_batch(batch, cin, ncls, patch, device, seed):
    """N(0,1) clipped to +-3 (post-AdaptiveNormalizer distribution) and blobby labels, generated on the device"""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    x = torch.randn((batch, cin, patch, patch, patch), generator=g, device=device).clamp_(-3, 3)
    coarse = torch.rand((batch, 1, patch // 8, patch // 8, patch // 8), generator=g, device=device)
    lab = torch.nn.functional.interpolate(coarse, scale_factor=8, mode='nearest')
    t = torch.floor(lab * ncls).clamp_(0, ncls - 1)
    return x.contiguous(), t.contiguous()


class KernelTimer(object):
    """brackets every launch of the MFMA 3x3x3 convolution entry point with events on torch's current stream (the
    stream the C ABI launches on) and books its algorithmic FLOPs"""

    def __init__(self):
        from segmentation3d import _engine
        self.E = _engine
        self.records = []
        self.wgrad_records = []
        self._orig = None

    def __enter__(self):
        E = self.E
        self._orig = E.call
        timer = self

        def call(name, *args):
            if name in WGRAD_ENTRY_POINTS:
                N, D, H, W, Cin, Cout = args[4:10]
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                rc = timer._orig(name, *args)
                b.record()
                timer.wgrad_records.append(((N, D, H, W, Cin, Cout, name), a, b))
                return rc
            if name not in ('seg3d_conv3d_k3_mfma_fwd', 'seg3d_conv3d_k3_bf16_fwd', 'seg3d_conv3d_k3_wino_fwd',
                            'seg3d_conv3d_k3_wino2d_fwd', 'seg3d_conv3d_k3_wino2d_fwd_ws'):
                return timer._orig(name, *args)
            wino = name in ('seg3d_conv3d_k3_wino_fwd', 'seg3d_conv3d_k3_wino2d_fwd', 'seg3d_conv3d_k3_wino2d_fwd_ws')
            N, D, H, W, Cin, Cout = args[6:12] if (wino and not name.endswith('_ws')) else args[7:13]
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            rc = timer._orig(name, *args)
            b.record()
            ma = ((WINO2D_CELL_VARIANT if (D % 8 or H % 8 or W % 8) else WINO2D_VARIANT) if '2d' in name else WINO_VARIANT) if wino else E.query(
                'seg3d_conv3d_k3_bf16_variant' if 'bf16' in name else 'seg3d_conv3d_k3_mfma_variant', N, D, H, W, Cin, Cout)
            timer.records.append(((N, D, H, W, Cin, Cout, ma), a, b, 2 if 'bf16' in name else 4))
            return rc
        E.call = call
        return self

    def __exit__(self, *exc):
        self.E.call = self._orig

    def summary(self):
        torch.cuda.synchronize()
        table = {}
        for key, a, b, elem_bytes in self.records:
            N, D, H, W, Cin, Cout, ma = key
            ms = a.elapsed_time(b)
            flops = 2.0 * N * D * H * W * 27 * Cin * Cout
            e = table.setdefault(key, {'launches': 0, 'ms': 0.0, 'flops': 0.0, 'bytes': 0.0})
            e['launches'] += 1
            e['ms'] += ms
            e['flops'] += flops
            # algorithmic bytes: input + output once (+ the packed weights); bf16 mode moves 2-byte elements both ways
            e['bytes'] += elem_bytes * (float(N) * D * H * W * (Cin + Cout) + 27.0 * Cin * Cout)
        return table

    def wgrad_summary(self):
        """weight-gradient launches of the 3x3x3 C -> C layers (entry point + the reduce kernel it ends with), by entry point:
        {name: {'launches', 'ms', 'flops' (algorithmic), 'executed' (issued to the matrix cores)}}"""
        torch.cuda.synchronize()
        table = {}
        for key, a, b in self.wgrad_records:
            N, D, H, W, Cin, Cout, name = key
            flops = 2.0 * N * D * H * W * 27 * Cin * Cout
            e = table.setdefault(name, {'launches': 0, 'ms': 0.0, 'flops': 0.0, 'executed': 0.0})
            e['launches'] += 1
            e['ms'] += a.elapsed_time(b)
            e['flops'] += flops
            e['executed'] += flops * WGRAD_ENTRY_POINTS[name][1]
        return table


WINO_VARIANT = 500   # the Winograd F(2,3) kernel (csrc/conv_wino.hip) in the variant column of the launch tables
WINO_EXECUTED = 2.0 / 3.0   # it executes 4 multiplies per 2 outputs x 3 taps: 2/3 of the algorithmic FLOPs
WINO2D_VARIANT = 600   # the Winograd F(2x2,3x3) kernel (csrc/conv_wino2d.hip)
WINO2D_EXECUTED = 4.0 / 9.0   # 16 multiplies per 4 outputs x 9 (ky, kx) taps
WINO2D_CELL_VARIANT = 601   # the same on 4^3 cells (conv3d_k3_wino2d_c4_kernel: levels that are multiples of 4 but not of 8)
# weight-gradient entry points of the 3x3x3 C -> C layers: (kernel symbol, executed share of the algorithmic FLOPs)
WGRAD_ENTRY_POINTS = {'seg3d_conv3d_k3_wino2d_wgrad': ('conv3d_k3_wgrad_wino2d_kernel', WINO2D_EXECUTED),   # F(3x3, 2x2)
                      'seg3d_conv3d_k3_wino_wgrad': ('conv3d_k3_wgrad_wino_kernel', WINO_EXECUTED),          # F(3, 2)
                      'seg3d_conv3d_k3_mfma_wgrad': ('conv3d_k3_wgrad2_kernel', 1.0),
                      'seg3d_conv3d_k3_bf16_wgrad': ('conv3d_k3_wgrad3_bf16_kernel', 1.0)}


def executed_share(variant):
    return WINO_EXECUTED if variant == WINO_VARIANT else WINO2D_EXECUTED if variant in (WINO2D_VARIANT, WINO2D_CELL_VARIANT) else 1.0


def variant_kernel_name(v):
    """seg3d_conv3d_k3_mfma_variant code -> kernel symbol as rocprofv3 prints it"""
    if v == WINO_VARIANT:
        return 'conv3d_k3_wino_kernel'
    if v == WINO2D_VARIANT:
        return 'conv3d_k3_wino2d_kernel'
    if v == WINO2D_CELL_VARIANT:
        return 'conv3d_k3_wino2d_c4_kernel'
    if v >= 400:
        return 'conv3d_k3_mfma2w8_bf16_kernel<{}, {}, true>'.format((v - 400) // 10, v % 10)
    if v >= 300:   # two waves per SIMD, MA row blocks per wave
        return 'conv3d_k3_mfma2w8_kernel<{}, {}>'.format((v - 300) // 10, v % 10)
    if v >= 200:
        # third template argument: bf16 output -- every data-gradient, and every forward launch
        return 'conv3d_k3_mfma2_bf16_kernel<{}, {}, true>'.format((v - 200) // 10, v % 10)
    if v >= 100:
        return 'conv3d_k3_mfma2_kernel<{}, {}>'.format((v - 100) // 10, v % 10)
    return 'conv3d_k3_mfma_kernel<{}>'.format(v)


def mfma_roofline_fields(variant, algorithmic_tflops, peak):
    """`achieved` / `frac` price the multiply-adds the kernel ISSUES to the matrix cores against the dense MFMA peak, so
    frac <= 1 by construction.  A Winograd kernel issues only 4/9 (F(2x2,3x3)) or 2/3 (F(2,3)) of the ALGORITHMIC FLOPs
    (2 x 27 x Cin x Cout per output voxel); the algorithmic rate -- what the layer is worth to the step -- is reported in
    its own fields and may exceed the peak."""
    ex = executed_share(variant)
    out = {'achieved': round(algorithmic_tflops * ex, 2), 'peak': peak, 'unit': 'TFLOP/s',
           'frac': round(algorithmic_tflops * ex / peak, 4),
           'algorithmic_tflops': round(algorithmic_tflops, 2), 'frac_algorithmic': round(algorithmic_tflops / peak, 4),
           'executed_share_of_algorithmic_flops': round(ex, 4)}
    if ex != 1.0:
        what = ('F(2,3) along x: 4 multiplies per 2 outputs x 3 taps' if variant == WINO_VARIANT
                else 'F(2x2,3x3) over (y, x): 16 multiplies per 4 outputs x 9 taps')
        out['flops_note'] = ('Winograd ' + what + ': `achieved` / `frac` = FLOPs issued to the matrix cores / dense fp32 MFMA '
                             'peak; algorithmic_tflops = 2*27*Cin*Cout per output voxel / time')
    return out


def pmc_traffic_gb(kernel_name):
    """HBM-side bytes per launch of a kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/r02_pmc_fetch_write_per_kernel.json: FETCH_SIZE and WRITE_SIZE in KB, collected in separate --pmc
    passes).  gfx950 correction per MI355X_MICROARCH.md section HBM: FETCH_SIZE under-reports wide coalesced reads
    by 2x, WRITE_SIZE is exact.  Counters cannot be collected from inside bench.py; returns None when absent."""
    prof = os.path.join(REPO, 'profiles')
    cands = (['r04_bf16_pmc_fetch_write_per_kernel.json', 'r03_bf16_pmc_fetch_write_per_kernel.json',
              'r02_bf16_pmc_fetch_write_per_kernel.json', 'r01_i_bf16_pmc_fetch_write_per_kernel.json'] if 'bf16' in kernel_name
             else ['r04_pmc_fetch_write_per_kernel.json', 'r03_pmc_fetch_write_per_kernel.json', 'r02_pmc_fetch_write_per_kernel.json',
                   'r01_pmc_fetch_write_per_kernel.json'])
    path = next((os.path.join(prof, c) for c in cands if os.path.isfile(os.path.join(prof, c))), None)
    if path is None:
        return None
    with open(path) as f:
        table = json.load(f)
    base = kernel_name.replace('void ', '', 1)
    # exact name first (non-template kernels print without 'void'); else every template instantiation of the kernel
    # (`void name<...>`), averaged by launches -- the F(2x2,3x3) kernel has <BIAS, ADD> instantiations since round 3
    hits = [e for k, e in table.items() if k in (kernel_name, base)]
    if not hits:
        hits = [e for k, e in table.items() if k.replace('void ', '', 1).startswith(base + '<')]
    hits = [e for e in hits if 'FETCH_SIZE_KB_per_launch' in e and 'WRITE_SIZE_KB_per_launch' in e]
    if not hits:
        return None
    n = sum(e.get('launches', 1) for e in hits)
    kb = sum((2.0 * e['FETCH_SIZE_KB_per_launch'] + e['WRITE_SIZE_KB_per_launch']) * e.get('launches', 1) for e in hits) / n
    return round(kb * 1024 / 1e9, 3)


def host_cpu():
    """(model name, physical core count) of the host from /proc/cpuinfo; (None, None) where it cannot be read"""
    try:
        model, cores = None, set()
        phys = core = None
        with open('/proc/cpuinfo') as f:
            for line in f:
                k, _, v = line.partition(':')
                k, v = k.strip(), v.strip()
                if k == 'model name' and model is None:
                    model = v
                elif k == 'physical id':
                    phys = v
                elif k == 'core id':
                    core = v
                elif not k and phys is not None and core is not None:
                    cores.add((phys, core))
                    phys = core = None
        if phys is not None and core is not None:
            cores.add((phys, core))
        return model, (len(cores) or None)
    except OSError:
        return None, None


def time_cpu_baseline(net_name, cin, ncls, patch, loss_name):
    """the CPU side of every number this file reports, on this box's host cores, with
    oracle/torch_ref.py + oracle/numpy_ref.py = the stock torch CPU ops and host loops the reference composes:
      * train step (fwd + loss + bwd + torch.optim.Adam) at batch 1 (3 timed steps after a warm-up) and
        batch 4 (1 step);
      * forward only, batch 1 (2 timed);
      * BASELINE config 1: whole-volume sliding window over a synthetic 128^3 volume (96^3 boxes, stride 48 => 8 patches,
        adaptive normaliser, accumulate, divide, arg-max) with ONE forward per patch and with the reference's TWO.
    A bounded sample (~1 minute of CPU work on the 8-core build container class of host, less on the GPU box)."""
    from oracle import numpy_ref, torch_ref
    from segmentation3d.network import vnet, vbnet
    plugin = {'vnet': vnet, 'vbnet': vbnet}[net_name]
    torch.manual_seed(0)
    net = plugin.SegmentationNet(cin, ncls)
    plugin.parameters_kaiming_init(net)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    opt = torch.optim.Adam(list(sd.values()), lr=1e-4, betas=(0.9, 0.999))
    x4, t4 = synthetic_batch(4, cin, ncls, patch, torch.device('cpu'), 123)
    x, t = x4[:1].contiguous(), t4[:1].contiguous()
    kw = {'weights': [1.0 / ncls] * ncls} if loss_name == 'Dice' else {'class_num': ncls, 'alpha': None, 'gamma': 2}
    times = []
    for i in range(4):
        t0 = time.time()
        torch_ref.train_step(sd, opt, x, t, net_name, loss_name, kw)
        times.append(time.time() - t0)
    steady = times[1:]
    t0 = time.time()
    torch_ref.train_step(sd, opt, x4, t4, net_name, loss_name, kw)
    step_b4 = time.time() - t0
    fwd = []
    sd_eval = {k: v.detach() for k, v in sd.items()}
    with torch.no_grad():
        for i in range(3):
            t0 = time.time()
            torch_ref.segmentation_net(x, sd_eval, net_name)
            fwd.append(time.time() - t0)
    fwd_steady = fwd[1:]
    cpu_model, physical = host_cpu()
    out = {'value': round(1.0 / (sum(steady) / len(steady)), 4), 'unit': 'patches/s', 'cores': torch.get_num_threads(),
           'cpu_model': cpu_model, 'physical_cores': physical, 'logical_cpus': os.cpu_count(),
           'batch4_patches_per_s': round(4.0 / step_b4, 4),
           'forward_only_patches_per_s': round(1.0 / (sum(fwd_steady) / len(fwd_steady)), 4),
           'kind': 'port',
           'sample': '{} timed train steps (fwd+{}+bwd+Adam) of {}({},{}) on one {}^3 patch, batch 1, after 1 warm-up, plus 1 '
                     'step at batch 4 and 2 forward-only passes; oracle/torch_ref.py = the stock torch CPU ops the '
                     'reference composes'.format(len(steady), loss_name, net_name, cin, ncls, patch)}
    if cin == 1:
        vol = torch.randn((128, 128, 128), generator=torch.Generator().manual_seed(11)).numpy()

        def net_fn(a):
            with torch.no_grad():
                return torch_ref.segmentation_net(torch.from_numpy(a), sd_eval, net_name).numpy()
        secs = {}
        for label, double in (('single_forward', False), ('double_forward_as_reference', True)):
            t0 = time.time()
            _, _, (starts, _) = numpy_ref.sliding_window_inference(vol, net_fn, ncls, (1.0, 1.0, 1.0), [float(patch)] * 3,
                                                                    [patch / 2.0] * 3, 16, {'type': 1, 'clip_sigma': 3},
                                                                    double_forward=double)
            secs[label] = round(time.time() - t0, 3)
        out['config1_whole_volume_128'] = {'patches': len(starts), 'seconds_' + 'single_forward': secs['single_forward'],
                                           'seconds_double_forward_as_reference': secs['double_forward_as_reference'],
                                           'what': 'oracle/numpy_ref.sliding_window_inference on a synthetic 128^3 volume, '
                                                   '{0}^3 boxes at stride {1}, adaptive normaliser, host accumulate / '
                                                   'divide / arg-max (BASELINE config 1 without file IO)'.format(
                                                       patch, patch // 2)}
    return out


def time_inference(net, volume_xyz, patch, stride, ncls, batch, device, two_streams=True, world=1, dtype='fp32'):
    """whole-volume sliding-window job.  world > 1: every rank holds the volume, the patch list is sharded into z-contiguous
    chunks (core/seg_infer.SlabShardPlan), only the halo planes travel point to point, every rank divides / arg-maxes its own
    slab and the mask slabs are replicated (gather='mask'); seconds = MAX over ranks of the job time."""
    from segmentation3d.core.seg_infer import sliding_window_inference
    from segmentation3d.utils.image_tools import image_partition_by_fixed_size
    X, Y, Z = volume_xyz
    g = torch.Generator(device=device)
    g.manual_seed(7)
    host = torch.randn((Z, Y, X), generator=torch.Generator().manual_seed(7))
    starts, ends = image_partition_by_fixed_size(((X, Y, Z), (1.0, 1.0, 1.0)), [0, 0, 0], [X, Y, Z], [patch] * 3,
                                                 [stride] * 3, 16)
    net.eval()
    torch.cuda.synchronize()
    t0 = time.time()
    vol = host.to(device)
    torch.cuda.synchronize()
    t_h2d = time.time() - t0
    # the complete job (accumulator allocation, warm-up batch, graph capture, all replays, halo merge, divide + arg-max,
    # mask replication): one untimed warm-up job -- the first job of a process also pays the one-time graph-pool allocation,
    # code-object loading and (N > 1) communicator set-up for point-to-point transfers; it is reported as first_job_seconds,
    # like the warm-up steps of the train loop it is not part of `seconds` -- then three timed jobs, median reported
    runs = []
    first_job = None
    for it in range(4):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.time()
        probs, mask, batcher = sliding_window_inference(net, vol, starts, (patch,) * 3, ncls, {'type': 1, 'clip_sigma': 3},
                                                        batch_size=batch, use_graph=True, two_streams=two_streams,
                                                        shard=world > 1, gather='mask')
        torch.cuda.synchronize()
        dt = time.time() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        if it == 0:
            first_job = dt
        else:
            runs.append(dt)
        if it < 3:
            del probs, mask
    t_dev = sorted(runs)[1]
    t0 = time.time()
    mask_host = mask.cpu()
    t_d2h = time.time() - t0
    shard_info = None
    if world > 1:
        plan = batcher.shard_plan
        halo = sum((ncls + 1) * (z1 - z0) * Y * X * 4 for _, _, z0, z1 in plan.transfers())
        shard_info = {'ranks': world, 'patches_per_rank': [len(q) for q in plan.patches],
                      'owned_z_slabs': [list(plan.owned(r)) for r in range(world)],
                      'halo_bytes_point_to_point': int(halo), 'mask_bytes_replicated': int(Z * Y * X),
                      'full_allreduce_bytes_avoided': int((ncls + 1) * Z * Y * X * 4),
                      'seconds_is': 'max over ranks of each complete job, median of 3 jobs'}
    # roofline of the job: one eager, single-stream forward of a half batch (what one stream of a replay runs) with every
    # launch of the MFMA 3x3x3 convolution bracketed by events -> its dominant instantiation against the fp32 MFMA peak,
    # and the whole job against the FLOP floor (algorithmic forward FLOPs of all patches / peak)
    roof = None
    try:
        from segmentation3d import _ops
        half = max(1, batch // 2)
        xb = torch.randn((half, 1, patch, patch, patch), device=device)
        cache_was = _ops.weight_cache(True)
        try:
            with torch.no_grad():
                net(xb)
                with KernelTimer() as kt:
                    net(xb)
                table = kt.summary()
        finally:
            _ops.PACK_CACHE.enabled = cache_was
        by_variant = {}
        for key, e in table.items():
            v = by_variant.setdefault(key[6], {'launches': 0, 'ms': 0.0, 'flops': 0.0})
            for k in v:
                v[k] += e[k]
        dom = max(by_variant, key=lambda m: by_variant[m]['ms'])
        d = by_variant[dom]
        conv_flops_per_patch = sum(e['flops'] for e in table.values()) / half
        achieved = d['flops'] / (d['ms'] * 1e-3) / 1e12
        job_tflop = conv_flops_per_patch * len(starts) / 1e12
        job_exec = sum(e['flops'] * executed_share(k[6]) for k, e in table.items()) / half * len(starts) / 1e12
        peak = BF16_MFMA_PEAK_TFLOPS if dtype == 'bf16' else FP32_MFMA_PEAK_TFLOPS
        roof = {'kernel': variant_kernel_name(dom), 'bound': 'mfma', **mfma_roofline_fields(dom, achieved, peak),
                'launches_per_forward': d['launches'], 'avg_launch_ms': round(d['ms'] / d['launches'], 4),
                'job': {'algorithmic_tflop_mfma_convs': round(job_tflop, 2), 'executed_tflop_mfma_convs': round(job_exec, 2),
                        'executed_flop_floor_seconds': round(job_exec / peak, 4),
                        'frac_of_mfma_peak_executed': round(job_exec / t_dev / peak, 4),
                        'frac_of_mfma_peak_algorithmic': round(job_tflop / t_dev / peak, 4)},
                'note': 'one eager single-stream forward of {} patches after the timed jobs; the job runs two such half '
                        'batches on two streams inside each hipGraph replay'.format(half)}
    except Exception as exc:   # the roofline is a report, never a reason to lose the measurement
        roof = {'error': repr(exc)}
    return {'roofline': roof, 'workload': 'sliding-window inference, {}x{}x{} volume, {}^3 patches stride {}, {} patches, batch {} per '
                        'hipGraph replay (two half batches on two streams), 1 forward/patch{}'.format(
                            X, Y, Z, patch, stride, len(starts), batch,
                            ', patch list sharded over {} GPUs in z-contiguous chunks'.format(world) if world > 1 else ''),
            'seconds': round(t_dev, 4), 'seconds_all_runs': [round(v, 4) for v in runs], 'first_job_seconds': round(first_job, 4),
            'h2d_seconds': round(t_h2d, 4),
            'd2h_mask_seconds': round(t_d2h, 4), 'patches_per_s': round(len(starts) / t_dev, 2),
            'mask_nonzero': int((mask_host != 0).sum()), 'n_gpus': world, 'sharding': shard_info}


def time_config1_gpu(net, net_name, cin, ncls, patch, device):
    """BASELINE config 1 on the GPU, beside cpu_baseline.config1_whole_volume_128: one synthetic 128^3 volume, 96^3 boxes at
    stride 48 (8 patches), adaptive normaliser.  Two readings: `segmentation_volume` on a resident image (what the CPU figure
    covers: patch loop, accumulate, divide, arg-max; no file IO), and the reference's CLI entry `segmentation()` on a .mha
    file with a reference-layout model folder (model load + file read + inference, mask returned, nothing saved)."""
    import shutil
    import tempfile
    from segmentation3d.core.seg_infer import load_models, segmentation, segmentation_volume
    from segmentation3d.utils.image3d import Image3d
    from segmentation3d.utils.mha_io import write_mha
    root = tempfile.mkdtemp(prefix='seg3d_cfg1_')
    try:
        chk = os.path.join(root, 'model', 'fine', 'checkpoints', 'chk_1')
        os.makedirs(chk)
        norm = {'type': 1, 'clip_sigma': 3}
        torch.save({'epoch': 1, 'batch': 1, 'net': net_name, 'max_stride': 16,
                    'state_dict': {'module.' + k: v.detach().cpu() for k, v in net.state_dict().items()},
                    'spacing': [1.0, 1.0, 1.0], 'interpolation': 'LINEAR', 'in_channels': cin, 'out_channels': ncls,
                    'crop_normalizers': [norm]}, os.path.join(chk, 'params.pth'))
        import segmentation3d
        cfg_src = open(os.path.join(os.path.dirname(segmentation3d.__file__), 'config', 'infer_config.py')).read()
        cfg_src = cfg_src.replace("__C.general.single_scale = 'DISABLE'", "__C.general.single_scale = 'fine'")
        cfg_src = cfg_src.replace('__C.fine.pick_largest_cc = True', '__C.fine.pick_largest_cc = False')
        cfg_src += '\n__C.fine.partition_size = [{0}.0, {0}.0, {0}.0]\n__C.fine.partition_stride = [{1}.0, {1}.0, {1}.0]\n'.format(
            patch, patch // 2)
        with open(os.path.join(root, 'model', 'infer_config.py'), 'w') as f:
            f.write(cfg_src)
        vol = torch.randn((128, 128, 128), generator=torch.Generator().manual_seed(11)).numpy()
        frame = ((1.0, 1.0, 1.0), (0.0, 0.0, 0.0), (1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0))
        img = Image3d(vol, *frame)
        path = os.path.join(root, 'case.mha')
        write_mha(img, path)
        models = load_models(os.path.join(root, 'model'), device.index or 0)
        vols = []
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.time()
            _, mask = segmentation_volume(models['fine_model'], models['infer_cfg'].fine, img, None, None, True)
            torch.cuda.synchronize()
            vols.append(time.time() - t0)
        import contextlib
        import io
        cli = []
        for _ in range(2):
            t0 = time.time()
            with contextlib.redirect_stdout(io.StringIO()):
                segmentation(path, os.path.join(root, 'model'), os.path.join(root, 'out'), 'seg.mha', device.index or 0,
                             True, False, False, False)
            torch.cuda.synchronize()
            cli.append(time.time() - t0)
        return {'patches': 8, 'seconds_segmentation_volume': round(sorted(vols[1:])[1], 4),
                'seconds_segmentation_volume_all_runs': [round(v, 4) for v in vols],
                'seconds_segmentation_cli_entry': round(min(cli), 4), 'seconds_segmentation_cli_entry_all_runs': [round(v, 4) for v in cli],
                'mask_nonzero': int((mask.array != 0).sum()),
                'what': 'synthetic 128^3 volume at 1 mm, {0}^3 boxes at stride {1} (8 patches, one hipGraph-free batch), adaptive '
                        'normaliser, device accumulate / divide / arg-max, probabilities and mask copied to the host; '
                        'segmentation_volume = resident image (comparable with cpu_baseline.config1_whole_volume_128, which has no '
                        'file IO either; the reference\'s second, identical forward per patch is not repeated), cli_entry = '
                        'segmentation() on the .mha incl. model-folder load and file read'.format(patch, patch // 2)}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no rendezvous in the environment: this process (which has not touched the
    GPU and never will) starts N ranks, one per GPU, through torch.distributed.run on the loopback address, relays their
    output (rank 0 prints the JSON line) and exits with their exit code.  The ranks re-enter main() with WORLD_SIZE set."""
    import socket
    import subprocess
    share = os.environ.get('SEG3D_BENCH_SHARE_GPU', '0') == '1'   # rehearsal only: several gloo ranks on one GPU
    visible = torch.cuda.device_count()                            # (does not initialise the GPU)
    if visible < args.gpus and not share:
        sys.stderr.write('bench.py: --gpus {} requested but only {} GPU(s) are visible; refusing to report a {}-GPU number '
                         'from fewer devices\n'.format(args.gpus, visible, args.gpus))
        return 2
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // args.gpus)))
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        if rank == 0:
            sys.stderr.write('bench.py: --gpus {} does not match WORLD_SIZE {} of the launcher\n'.format(args.gpus, world))
        sys.exit(2)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm device (the engine has no CPU path)')
    # one process per GPU over RCCL ("nccl" on ROCm).  SEG3D_DIST_BACKEND=gloo allows rehearsing the N > 1 code path
    # with several ranks sharing one GPU (RCCL refuses two ranks on one device); never used for reported numbers.
    backend = os.environ.get('SEG3D_DIST_BACKEND', 'nccl')
    ndev = torch.cuda.device_count()
    if world > ndev and not (backend != 'nccl' and os.environ.get('SEG3D_BENCH_SHARE_GPU', '0') == '1'):
        raise SystemExit('bench.py: {} ranks but {} visible GPU(s)'.format(world, ndev))
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)

    if args.no_graph:
        args.step_mode = 'eager'
    args.graph = (world == 1) and args.step_mode != 'eager'
    from segmentation3d.core.seg_train import TrainStep
    from segmentation3d import _ops as _ops_mode
    _ops_mode.set_activation_dtype(args.dtype)
    weights = [1.0 / args.classes] * args.classes
    if args.no_wgrad_overlap:
        from segmentation3d import _ops as _ops_cfg
        _ops_cfg.WGRAD_SIDE_STREAM = False
    step = TrainStep(args.net, args.in_channels, args.classes, args.loss, weights if args.loss == 'Dice' else None,
                     device=device, seed=0, use_graph=args.graph and world == 1)
    x, t = synthetic_batch(args.batch, args.in_channels, args.classes, args.patch, device, 1000 + rank)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if step.use_graph:
        # capture of the step's hipGraph (two eager steps, then the capture) is setup, not warm-up: it must never fall
        # into the timed region, whatever --warmup says
        for _ in range(4):
            if step._graph is not None or not step.use_graph:
                break
            step(x, t)
    run = step
    probe = None
    freeze = os.environ.get('SEG3D_BENCH_GC_FREEZE', '1') != '0'
    if step.use_graph and args.step_mode == 'auto':
        # time a few steps both ways (after the capture, before the warm-up) and keep the faster issue mode
        if freeze:
            import gc
            gc.collect()
            gc.freeze()
        probe = {}
        for name, fn in (('graph', step), ('eager', step._eager), ('graph', step), ('eager', step._eager)):
            for _ in range(2):
                fn(x, t)
            sync()
            tp = time.perf_counter()
            for _ in range(6):
                fn(x, t)
            sync()
            ms = 1e3 * (time.perf_counter() - tp) / 6
            probe[name] = min(probe.get(name, 1e9), ms)
        if probe['eager'] < probe['graph']:
            run = step._eager
        probe = {k: round(v, 3) for k, v in probe.items()}
    graph_used = bool(step.use_graph and run is step)
    for _ in range(args.warmup):
        loss = run(x, t)
    if not graph_used and freeze:
        # eager steps (the multi-GPU path): the objects that live for the whole run leave the collector's young generations,
        # so the collections triggered by a step's short-lived autograd objects stay short (standard training-loop practice)
        import gc
        gc.collect()
        gc.freeze()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = run(x, t)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    final_loss = float(loss.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * args.batch * args.steps / elapsed

    roofline, kernels = None, None
    if not args.no_roofline:
        # two extra, instrumented steps AFTER the timed region; every rank runs them (the gradient all-reduce is a
        # collective), rank 0 reports
        # Per-kernel durations are taken with the weight-gradient side stream OFF: with it, an event pair around a launch on
        # the main stream also spans the time the kernel waits for (or shares the chip with) a weight-gradient kernel of
        # the other stream, which says nothing about the kernel itself.  `value` above is measured with the overlap on.
        from segmentation3d import _ops
        overlap_was = _ops.WGRAD_SIDE_STREAM
        _ops.WGRAD_SIDE_STREAM = False
        # An event pair also spans any time the launch itself arrives late (the event executes as soon as the queue is empty),
        # so the host must stay ahead of the GPU here: one eager step refills the queue after the timed loop, and the cyclic
        # garbage collector -- whose pauses drained the queue in the middle of a step and added 0.2 ms to individual launches
        # -- is parked for the two bracketed steps.
        import gc
        gc.collect()
        gc.disable()
        try:
            step._eager(x, t)
            with KernelTimer() as kt:
                for _ in range(2):
                    step._eager(x, t)      # eager launches (a captured step cannot be bracketed launch by launch)
            table = kt.summary()
        finally:
            gc.enable()
            _ops.WGRAD_SIDE_STREAM = overlap_was
    if not args.no_roofline and rank == 0:
        by_variant = {}
        for key, e in table.items():
            v = by_variant.setdefault(key[6], {'launches': 0, 'ms': 0.0, 'flops': 0.0, 'bytes': 0.0})
            for k in v:
                v[k] += e[k]
        dom = max(by_variant, key=lambda m: by_variant[m]['ms'])
        kname = variant_kernel_name(dom)
        traffic = pmc_traffic_gb('void ' + kname)
        d = by_variant[dom]
        achieved = d['flops'] / (d['ms'] * 1e-3) / 1e12
        peak = BF16_MFMA_PEAK_TFLOPS if args.dtype == 'bf16' else FP32_MFMA_PEAK_TFLOPS
        gbps = d['bytes'] / (d['ms'] * 1e-3) / 1e9
        hbm_frac = gbps / HBM_PEAK_GBPS
        roofline = {'kernel': kname, 'bound': 'mfma', **mfma_roofline_fields(dom, achieved, peak),
                    'traffic': traffic, 'traffic_unit': 'GB per launch (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes, see profiles/)',
                    'launches_per_step': d['launches'] // 2,
                    'avg_launch_ms': round(d['ms'] / d['launches'], 4),
                    'gflop_per_launch': round(d['flops'] / d['launches'] / 1e9, 2),
                    'share_of_step_ms': round(d['ms'] / 2, 3),
                    'note': 'launch durations from 2 instrumented eager steps with the weight-gradient side stream off '
                            '(kernels back to back on one stream, garbage collector parked); value/ms_per_step are '
                            'measured with the side stream on'}
        # the weight gradients of the same layers, measured the same way (entry point = main kernel + its slab reduce)
        wg = kt.wgrad_summary()
        if wg:
            wdom = max(wg, key=lambda n: wg[n]['ms'])
            w = wg[wdom]
            walg = w['flops'] / (w['ms'] * 1e-3) / 1e12
            wex = WGRAD_ENTRY_POINTS[wdom][1]
            roofline['wgrad'] = {'kernel': WGRAD_ENTRY_POINTS[wdom][0] + ' (+ its slab reduce)', 'bound': 'mfma',
                                 'achieved': round(walg * wex, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(walg * wex / peak, 4),
                                 'algorithmic_tflops': round(walg, 2), 'executed_share_of_algorithmic_flops': round(wex, 4),
                                 'launches_per_step': w['launches'] // 2, 'avg_launch_ms': round(w['ms'] / w['launches'], 4),
                                 'share_of_step_ms': round(w['ms'] / 2, 3)}
        # the whole step against the MFMA roof, everything measured: algorithmic and executed FLOPs of the 3x3x3 C -> C
        # convolutions (forward + data-gradient + weight-gradient launches bracketed above) / ms_per_step / peak
        fwd_alg = sum(e['flops'] for e in table.values()) / 2
        fwd_exec = sum(e['flops'] * executed_share(k[6]) for k, e in table.items()) / 2
        wg_alg = sum(e['flops'] for e in wg.values()) / 2
        wg_exec = sum(e['executed'] for e in wg.values()) / 2
        step_tflop, step_exec = (fwd_alg + wg_alg) / 1e12, (fwd_exec + wg_exec) / 1e12
        roofline['step'] = {'algorithmic_tflop_mfma_convs': round(step_tflop, 3), 'executed_tflop_mfma_convs': round(step_exec, 3),
                            'executed_flop_floor_ms': round(step_exec / peak * 1e3, 2),
                            'frac_of_peak': round(step_exec / (ms_per_step * 1e-3) / peak, 4),
                            'frac_of_peak_algorithmic': round(step_tflop / (ms_per_step * 1e-3) / peak, 4),
                            'mfma_conv_kernel_ms': round((sum(e['ms'] for e in table.values()) + sum(e['ms'] for e in wg.values())) / 2, 3),
                            'note': 'frac_of_peak = FLOPs issued to the matrix cores by the bracketed 3x3x3 convolution launches '
                                    '(forward, data-gradient, weight-gradient; all measured) / ms_per_step / peak, <= 1'}
        if hbm_frac > roofline['frac']:
            # (bf16 mode) the same launches priced against HBM: algorithmic bytes = input + output once
            roofline.update({'bound': 'hbm', 'achieved': round(gbps, 1), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                             'frac': round(hbm_frac, 4), 'gb_per_launch': round(d['bytes'] / d['launches'] / 1e9, 4),
                             'mfma_tflops': round(achieved, 2), 'mfma_frac': round(achieved / peak, 4)})
        else:
            roofline['hbm_frac_of_algorithmic_bytes'] = round(hbm_frac, 4)
        kernels = [{'N_D_H_W_Cin_Cout_variant': list(k), 'launches_per_step': e['launches'] // 2,
                    'avg_ms': round(e['ms'] / e['launches'], 4),
                    'tflops': round(e['flops'] / (e['ms'] * 1e-3) / 1e12, 2)} for k, e in sorted(table.items())]
        if args.kernel_report:
            os.makedirs(os.path.dirname(os.path.abspath(args.kernel_report)), exist_ok=True)
            with open(args.kernel_report, 'w') as f:
                json.dump({'mfma_conv_launches': kernels, 'ms_per_step': ms_per_step}, f, indent=1)

    infer, infer_timed_out = None, False
    if not args.no_infer:
        # every rank takes part (N > 1: the patch list is sharded, the halo merge and the mask replication are collectives)
        vx = tuple(int(v) for v in args.infer_volume.split(','))
        if world == 1:
            infer = time_inference(step.net, vx, args.patch, args.patch // 2, args.classes, args.infer_batch, device,
                                   world=world, dtype=args.dtype)
        else:
            # N > 1: the train-step measurement above is complete, and the sharded inference leg -- whose point-to-point halo
            # exchange has only ever been rehearsed over gloo -- must not cost the line: it runs in a worker thread; an
            # exception is reported in its place, and if the leg has not returned after --infer-timeout seconds (default 240;
            # every rank would be stuck in the same collective) rank 0 prints the line with infer.timed_out = true and every
            # rank leaves the process with exit code 3 without waiting for the stuck thread: a hang is a finding, never rc 0.
            import threading
            box = {}

            def leg():
                try:
                    torch.cuda.set_device(device)
                    box['infer'] = time_inference(step.net, vx, args.patch, args.patch // 2, args.classes, args.infer_batch,
                                                  device, world=world, dtype=args.dtype)
                except Exception as exc:
                    box['infer'] = {'error': repr(exc), 'n_gpus': world}
            th = threading.Thread(target=leg, daemon=True)
            th.start()
            th.join(args.infer_timeout)
            if th.is_alive():
                infer_timed_out = True
                infer = {'error': 'sharded inference leg did not return within {:.0f} s'.format(args.infer_timeout),
                         'timed_out': True, 'n_gpus': world}
            else:
                infer = box['infer']
        if world == 1 and args.in_channels == 1 and 'error' not in infer:
            try:
                infer['config1_whole_volume_128_gpu'] = time_config1_gpu(step.net, args.net, args.in_channels, args.classes,
                                                                          args.patch, device)
            except Exception as exc:   # a report beside the headline, never a reason to lose it
                infer['config1_whole_volume_128_gpu'] = {'error': repr(exc)}

    cpu_baseline = None
    if not args.no_cpu_baseline and world == 1 and rank == 0:
        cpu_baseline = time_cpu_baseline(args.net, args.in_channels, args.classes, args.patch, args.loss)

    collective_lib = 'none'
    if world > 1:
        collective_lib = backend
        if backend == 'nccl':
            try:
                collective_lib = 'RCCL {}'.format('.'.join(str(v) for v in torch.cuda.nccl.version()))
            except Exception:
                collective_lib = 'RCCL'
    if rank == 0:
        out = {
            'metric': 'patches/sec (96^3, 1-mod V-Net) train-step', 'value': round(value, 3), 'unit': 'patches/s',
            'n_gpus': dist.get_world_size() if world > 1 else 1, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if args.dtype == 'fp32' else 'bf16 (activations + packed k3 weights; f32 accumulate, GN statistics, master weights)',
            'data': 'synthetic',
            'config': {'workload': '{}({},{}) train step (fwd + {} loss + bwd + fused Adam), {} patches of {}^3 per GPU, '
                                   '{}, random-init weights'.format(args.net, args.in_channels, args.classes, args.loss,
                                                                    args.batch, args.patch, args.dtype),
                       'global_batch': world * args.batch, 'patch': args.patch,
                       'parallelism': 'dp{} (one process per GPU, bucketed {} all-reduce overlapped with backward)'.format(
                           world, collective_lib) if world > 1 else 'single GPU',
                       'wgrad_overlap': not args.no_wgrad_overlap, 'train_step_hipgraph': graph_used,
                       'step_mode': args.step_mode, 'step_mode_probe_ms': probe},
            'final_loss': round(final_loss, 6),
            'roofline': roofline, 'cpu_baseline': cpu_baseline, 'infer': infer,
        }
        print(json.dumps(out), flush=True)
    if infer_timed_out:
        sys.stdout.flush()
        sys.stderr.write('bench.py: rank {}: the sharded inference leg is stuck in a collective; leaving with exit code 3\n'.format(rank))
        sys.stderr.flush()
        os._exit(3)      # no clean shutdown is possible with a stuck collective; the line (rank 0) is out, the launcher sees rc != 0
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
