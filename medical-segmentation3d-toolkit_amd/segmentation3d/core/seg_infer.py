"""Sliding-window whole-volume inference -- MI355X-native replacement of the patch path of the reference's
core/seg_infer.py (segmentation_voi :208-246, segmentation_volume :249-350, load_single_model :99-164,
load_models :167-205).

Reference behaviour: for every partition box, slice the ROI on the host, normalise it, run the net TWICE on a batch
of one, copy every class map back to the host and `+=` it into a SimpleITK volume through full-volume numpy round
trips (utils/image_tools.py:446-450); finally multiply by 1/count and arg-max.

Here the volume stays resident in HBM.  `SlidingWindowBatcher` crops + normalises P patches per launch
(seg3d_patch_gather_normalize), the net runs ONCE per patch (the reference's two forwards are identical in eval mode,
so their mean is the single result bit for bit), and the class maps are accumulated on the device in list order
(seg3d_patch_scatter_accumulate, no atomics => the same float summation order as the reference's sequential loop).
gather -> net -> scatter is captured once into a hipGraph (torch.cuda.CUDAGraph drives hipStreamBeginCapture) and
replayed per batch; only a 4*(3P+7)-byte control block changes between replays.
"""
import importlib
import os
import threading
import time

import numpy as np
import torch

from segmentation3d import _engine as E
from segmentation3d import _ops
from segmentation3d.utils.image3d import Image3d
from segmentation3d.utils.image_tools import image_partition_by_fixed_size
from segmentation3d.utils.model_io import get_checkpoint_folder, strip_module_prefix
from segmentation3d.utils.normalizer import normalizer_from_dict


class SlidingWindowBatcher(object):
    """device-side crop/normalise + accumulate for one resident volume.

    :param volume: float32 device tensor [Z, Y, X]
    :param starts: list of [x, y, z] patch start voxels (all patches share `box`)
    :param box: (bx, by, bz) patch size in voxels
    :param num_classes: C
    :param normalizer: checkpoint-style dict {'type': 0|1, ...} or None (utils/normalizer.py:36-39,78-81)
    """

    def __init__(self, volume, starts, box, num_classes, normalizer, max_batch=16):
        E.require_device(volume)
        if volume.dim() != 3 or volume.dtype != torch.float32:
            raise ValueError('volume must be a float32 [Z, Y, X] tensor')
        self.volume = volume.contiguous()
        self.Z, self.Y, self.X = (int(s) for s in volume.shape)
        self.box = tuple(int(b) for b in box)
        self.C = int(num_classes)
        self.starts = [[int(v) for v in s] for s in starts]
        bx, by, bz = self.box
        for s in self.starts:
            if s[0] < 0 or s[1] < 0 or s[2] < 0 or s[0] + bx > self.X or s[1] + by > self.Y or s[2] + bz > self.Z:
                raise ValueError('patch {} with box {} leaves the volume {}'.format(s, self.box, (self.X, self.Y, self.Z)))
        self.normalizer = normalizer
        if normalizer is None:
            self._norm = (-1, 0.0, 1.0, 0, 1.0)
        elif normalizer['type'] == 0:
            self._norm = (0, float(normalizer['mean']), float(normalizer['stddev']), int(bool(normalizer['clip'])), 1.0)
        elif normalizer['type'] == 1:
            self._norm = (1, 0.0, 1.0, 1, float(normalizer['clip_sigma']))
        else:
            raise ValueError('Unsupported normalization type.')
        dev = volume.device
        self.acc = torch.zeros((self.C, self.Z, self.Y, self.X), dtype=torch.float32, device=dev)
        self.count = torch.zeros((self.Z, self.Y, self.X), dtype=torch.float32, device=dev)
        self.max_batch = int(max_batch)
        nblk = E.query('seg3d_patch_stats_blocks', bx, by, bz)
        self._stat_ws = torch.empty((self.max_batch * nblk * 2,), dtype=torch.float64, device=dev)
        self._mean_std = torch.empty((self.max_batch, 2), dtype=torch.float32, device=dev)
        # control block on the device: [P][3] starts then {lo xyz, extent xyz, n_valid}
        self._ctl = torch.zeros((3 * self.max_batch + 7,), dtype=torch.int32, device=dev)
        self._plan = None

    # ---- control block ---------------------------------------------------------------------------------------------
    def _control_words(self, idx):
        P = self.max_batch
        if not 0 < len(idx) <= P:
            raise ValueError('batch of {} patches exceeds max_batch={}'.format(len(idx), P))
        bx, by, bz = self.box
        sel = np.array([self.starts[k] for k in idx], dtype=np.int32)
        lo = sel.min(0)
        hi = sel.max(0) + np.array([bx, by, bz], dtype=np.int32)
        host = np.zeros((3 * P + 7,), dtype=np.int32)
        host[:3 * len(idx)] = sel.reshape(-1)
        host[3 * len(idx):3 * P] = np.tile(sel[0], P - len(idx))  # padding patches: gathered, never scattered
        host[3 * P:3 * P + 3] = lo
        host[3 * P + 3:3 * P + 6] = hi - lo
        host[3 * P + 6] = len(idx)
        return host

    def set_batch(self, idx):
        """upload starts / bounding box / valid count of the patches `idx` (synchronous 4*(3P+7)-byte copy)"""
        self._ctl.copy_(torch.from_numpy(self._control_words(idx)))

    def plan(self, batches):
        """upload the control blocks of ALL batches in one copy; `select(b)` then switches batch with a stream-ordered
        device-to-device copy, so the host never has to wait for the GPU between batches"""
        words = np.stack([self._control_words(idx) for idx in batches])
        self._plan = torch.from_numpy(words).to(self._ctl.device)

    def select(self, b):
        self._ctl.copy_(self._plan[b], non_blocking=True)

    def _starts_ptr(self):
        return E.ptr(self._ctl)

    def _ctl_ptr(self):
        import ctypes
        return ctypes.c_void_p(self._ctl.data_ptr() + 4 * 3 * self.max_batch)

    # ---- kernels -----------------------------------------------------------------------------------------------------
    def gather_current(self, out=None):
        """crop + normalise the max_batch patches described by the control block -> [P, 1, bz, by, bx]"""
        bx, by, bz = self.box
        P = self.max_batch
        if out is None:
            out = torch.empty((P, 1, bz, by, bx), dtype=torch.float32, device=self.volume.device)
        ntype, mean, std, clip, sigma = self._norm
        E.call('seg3d_patch_gather_normalize', E.ptr(self.volume), self._starts_ptr(), E.ptr(out), E.ptr(self._stat_ws),
               E.ptr(self._mean_std), self.Z, self.Y, self.X, bx, by, bz, P, ntype, mean, std, clip, sigma, E.stream_ptr())
        return out

    def scatter_current(self, probs):
        """accumulate probs [P, C, bz, by, bx] of the control block's valid patches into acc / count"""
        bx, by, bz = self.box
        P = self.max_batch
        if tuple(probs.shape) != (P, self.C, bz, by, bx) or not probs.is_contiguous():
            raise ValueError('probs must be contiguous [{}, {}, {}, {}, {}], got {}'.format(P, self.C, bz, by, bx,
                                                                                            tuple(probs.shape)))
        max_box = min(self.X * self.Y * self.Z, min(self.X, bx * P) * min(self.Y, by * P) * min(self.Z, bz * P))
        E.call('seg3d_patch_scatter_accumulate', E.ptr(probs), self._starts_ptr(), self._ctl_ptr(), E.ptr(self.acc),
               E.ptr(self.count), self.Z, self.Y, self.X, bx, by, bz, self.C, max_box, E.stream_ptr())

    # convenience (eager) forms used by tests and the non-graph path
    def gather(self, idx):
        self.set_batch(idx)
        return self.gather_current()[:len(idx)]

    def scatter(self, idx, probs):
        self.set_batch(idx)
        P = self.max_batch
        if probs.shape[0] != P:
            pad = torch.zeros((P - probs.shape[0],) + tuple(probs.shape[1:]), dtype=probs.dtype, device=probs.device)
            probs = torch.cat((probs, pad), 0)
        self.scatter_current(probs.contiguous())

    def finalize(self, z_range=None):
        """acc *= 1/count (in place) and arg-max -> (probs [C,Z,Y,X], mask int8 [Z,Y,X]).
        z_range = (z0, z1): only that slab of planes (a rank of the sharded sliding window finalizes the slab it owns);
        the mask is zero outside it"""
        import ctypes
        plane = self.Y * self.X
        if z_range is None:
            mask = torch.empty((self.Z, self.Y, self.X), dtype=torch.int8, device=self.volume.device)
            z0, z1 = 0, self.Z
        else:
            mask = torch.zeros((self.Z, self.Y, self.X), dtype=torch.int8, device=self.volume.device)
            z0, z1 = int(z_range[0]), int(z_range[1])
            if not 0 <= z0 <= z1 <= self.Z:
                raise ValueError('z range {} outside the volume'.format(z_range))
        if z1 > z0:
            E.call('seg3d_finalize_argmax', ctypes.c_void_p(self.acc.data_ptr() + 4 * z0 * plane),
                   ctypes.c_void_p(self.count.data_ptr() + 4 * z0 * plane), ctypes.c_void_p(mask.data_ptr() + z0 * plane),
                   self.C, (z1 - z0) * plane, self.Z * plane, E.stream_ptr())
        return self.acc, mask


def shard_batches(batches, rank, world_size):
    """round-robin deal of patch batches (kept for callers that merge with a full all-reduce; the sliding window itself
    uses SlabShardPlan below)"""
    return [b for i, b in enumerate(batches) if i % world_size == rank]


class SlabShardPlan(object):
    """Patch inference over several GPUs: patches are independent, so the patch list is cut into `world`
    chunks that are CONTIGUOUS IN Z (sorted by z start, ties in list order; equal patch counts).  Rank r then only ever
    touches the planes [touch_lo[r], touch_hi[r]) of its accumulators, and the volume is cut into disjoint OWNED slabs
    [bounds[r], bounds[r + 1]) with bounds[r] = the smallest z start of rank r.  A lower rank's patches reach at most
    box_z - stride_z planes into the slabs above it, so the merge moves only those halo planes (rank q -> owner r,
    q < r) instead of all-reducing (C + 1) full volumes; every rank finalizes (divide + arg-max) its own slab.
    Pure host index arithmetic: the same plan is computed by every rank."""

    def __init__(self, starts, box, volume_zyx, world_size):
        self.world = int(world_size)
        self.Z = int(volume_zyx[0])
        bz = int(box[2])
        n = len(starts)
        order = sorted(range(n), key=lambda k: (int(starts[k][2]), k))
        cuts = [(n * r) // self.world for r in range(self.world + 1)]
        self.patches = [sorted(order[cuts[r]:cuts[r + 1]]) for r in range(self.world)]   # list order within a rank
        self.touch_lo, self.touch_hi, self.bounds = [], [], []
        prev = 0
        for r in range(self.world):
            zs = [int(starts[k][2]) for k in self.patches[r]]
            lo = min(zs) if zs else prev
            hi = max(zs) + bz if zs else prev
            self.touch_lo.append(lo)
            self.touch_hi.append(hi)
            self.bounds.append(0 if r == 0 else max(lo, prev))
            prev = self.bounds[-1]
        self.bounds.append(self.Z)

    def owned(self, rank):
        return self.bounds[rank], self.bounds[rank + 1]

    def transfers(self):
        """[(src rank, dst rank, z0, z1)]: planes of src's accumulators that belong to dst's slab, in a fixed global order"""
        out = []
        for r in range(self.world):
            o0, o1 = self.owned(r)
            for q in range(self.world):
                if q == r or not self.patches[q]:
                    continue
                z0, z1 = max(o0, self.touch_lo[q]), min(o1, self.touch_hi[q])
                if z1 > z0:
                    out.append((q, r, z0, z1))
        return out


def _stage_on_host(t, group):
    """gloo (the CPU rehearsal backend; also used when several test ranks share one GPU) moves host memory only: device
    buffers are staged through the host there.  RCCL ("nccl") takes device pointers directly."""
    import torch.distributed as dist
    return t.is_cuda and dist.get_backend(group) == 'gloo'


def merge_slabs(acc, count, plan, rank, group=None):
    """exchange the halo planes of a SlabShardPlan: afterwards acc [C,Z,Y,X] / count [Z,Y,X] of rank r are complete
    inside plan.owned(r) (point-to-point sends over RCCL / xGMI; nothing is exchanged for planes only one rank touched).
    The owner adds the incoming partial sums in ascending source-rank order (fixed => reproducible)."""
    import torch.distributed as dist
    C = acc.shape[0]
    stage = _stage_on_host(acc, group)
    ops, recvs, keep = [], [], []
    for q, r, z0, z1 in plan.transfers():
        if rank == q:
            buf = torch.cat((acc[:, z0:z1], count[z0:z1].unsqueeze(0)), 0).contiguous()
            if stage:
                buf = buf.cpu()
            keep.append(buf)
            peer = r if group is None else dist.get_global_rank(group, r)
            ops.append(dist.P2POp(dist.isend, buf, peer, group))
        elif rank == r:
            buf = torch.empty((C + 1, z1 - z0) + tuple(acc.shape[2:]), dtype=acc.dtype,
                              device='cpu' if stage else acc.device)
            peer = q if group is None else dist.get_global_rank(group, q)
            ops.append(dist.P2POp(dist.irecv, buf, peer, group))
            recvs.append((z0, z1, buf))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for z0, z1, buf in recvs:
        buf = buf.to(acc.device)
        acc[:, z0:z1] += buf[:C]
        count[z0:z1] += buf[C]


def gather_slabs(probs, mask, plan, group=None, with_probs=True):
    """replicate the finalized slabs on every rank: each owner broadcasts its slab of every class map (contiguous planes,
    no staging) and of the mask -- the output's own size, once"""
    import torch.distributed as dist
    stage = _stage_on_host(mask, group)

    def bcast(view, src):
        if not stage:
            dist.broadcast(view, src=src, group=group)
            return
        host = view.cpu()
        dist.broadcast(host, src=src, group=group)
        view.copy_(host)
    for r in range(plan.world):
        z0, z1 = plan.owned(r)
        if z1 <= z0:
            continue
        src = r if group is None else dist.get_global_rank(group, r)
        bcast(mask[z0:z1], src)
        if with_probs:
            for c in range(probs.shape[0]):
                bcast(probs[c, z0:z1], src)


def _forward_two_streams(net, batch, side):
    """forward of one patch batch as two half batches on two HIP streams: while one half runs an MFMA-bound conv, the
    other half's HBM-bound GroupNorm / stride-2 / head kernels share the chip (patches are independent; GroupNorm(1, C)
    is per sample, so the result is identical).  Works eagerly and under hipGraph capture (fork / join by events)."""
    P = batch.shape[0]
    if side is None or P < 2:
        return net(batch).contiguous()
    h = P // 2
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with _ops.two_forwards():
        with torch.cuda.stream(side):
            out_b = net(batch[h:])
        out_a = net(batch[:h])
    cur.wait_stream(side)
    out_b.record_stream(cur)
    return torch.cat([out_a, out_b], 0)


def sliding_window_inference(net, volume, starts, box, num_classes, normalizer, batch_size=8, use_graph=True,
                             process_group=None, shard=False, two_streams=True, gather='all'):
    """run `net` over all patches of a device-resident volume; returns (probs [C,Z,Y,X], mask int8 [Z,Y,X], batcher).
    With shard=True under an initialised torch.distributed group the patch list is cut into z-contiguous chunks, one per
    rank (SlabShardPlan); after the patch loop only the halo planes are exchanged (merge_slabs), every rank divides and
    arg-maxes the slab it owns, and the slabs are replicated (gather='all': probabilities and mask, 'mask': mask only,
    'none': each rank keeps just its slab; batcher.shard_plan tells which).  Float summation order inside a halo then
    differs from the sequential reference loop by rounding only."""
    with torch.cuda.device(volume.device):
        return _sliding_window_inference(net, volume, starts, box, num_classes, normalizer, batch_size, use_graph,
                                         process_group, shard, two_streams, gather)


# hipGraph capture for the per-volume graphs.  `with torch.cuda.graph(g)` empties the caching allocator before every capture
# and gives every graph a private pool that is handed back to the driver when the graph dies: one volume = ~50 hipMalloc +
# ~50 hipFree calls (measured: torch.cuda.memory_stats num_device_alloc / num_device_free grow by 51-54 per job), which
# cost 30-60 ms on a healthy box and far more on a box where the driver is slow at it (whole jobs read 1.4-1.8 s instead
# of 0.96 s; with 50 patches per replay the calls alone made every job after the first take 2.8 s).  All volume graphs of a
# device therefore capture into ONE pool that a tiny keep-alive graph holds for the life of the process: the segments a
# dead graph leaves behind are reused by the next capture, and nothing is emptied.
# The caching allocator keeps free blocks per (pool, stream): the warm-up stream, the capture stream and the second forward
# stream are therefore created once per device as well -- with fresh streams per volume no block was ever reused (reserved
# memory grew by 20 GB per job).
# Retained footprint: the pool keeps every segment a volume graph of this device ever needed (the peak intermediates of the
# largest box / batch seen, ~20 GB for 16 patches of 96^3) reserved until release_graph_pool(device) is called --
# torch.cuda.empty_cache() cannot return memory of a live graph pool.
_GRAPH_POOLS = {}
_JOB_STREAMS = {}
_GRAPH_POOL_LOCK = threading.Lock()


def _job_stream(device, which):
    key = (device.index if device.index is not None else torch.cuda.current_device(), which)
    st = _JOB_STREAMS.get(key)
    if st is None:
        st = _JOB_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


def _capture_in_shared_pool(fn, device):
    """capture fn() into a CUDAGraph whose memory comes from the device's shared volume-graph pool.  Graphs that share a pool
    must not be alive at the same time unless they replay in capture order (the pool hands a dead temporary's block to the
    next capture): the shared pool therefore serves ONE live volume graph per device; a second concurrent job (another
    thread, a nested call) captures into a private pool of its own, as torch.cuda.graph would."""
    import weakref
    key = device.index if device.index is not None else torch.cuda.current_device()
    cur = torch.cuda.current_stream()
    # the check of `live`, the claim of the shared pool and the creation of the pool itself are one critical section: two
    # threads that both saw live == 0 would capture into the same pool while both graphs are alive
    with _GRAPH_POOL_LOCK:
        entry = _GRAPH_POOLS.get(key)
        if entry is None:
            stream = _job_stream(device, 'capture')
            pool = torch.cuda.graph_pool_handle()
            keep = torch.cuda.CUDAGraph()
            anchor = torch.zeros(1, device=device)
            stream.wait_stream(cur)
            with torch.cuda.stream(stream):
                keep.capture_begin(pool=pool)
                anchor.add_(1.0)
                keep.capture_end()
            cur.wait_stream(stream)
            entry = _GRAPH_POOLS[key] = {'pool': pool, 'keep': keep, 'anchor': anchor, 'live': 0}
        shared = entry['live'] == 0
        if shared:
            entry['live'] += 1
    stream = _job_stream(device, 'capture') if shared else torch.cuda.Stream(device=device)
    graph = torch.cuda.CUDAGraph()
    try:
        stream.wait_stream(cur)
        with torch.cuda.stream(stream):
            graph.capture_begin(pool=entry['pool'] if shared else torch.cuda.graph_pool_handle())
            try:
                fn()
            finally:
                graph.capture_end()
        cur.wait_stream(stream)
    except BaseException:
        if shared:
            with _GRAPH_POOL_LOCK:
                entry['live'] -= 1
        raise
    if shared:
        def _released(e=entry):
            with _GRAPH_POOL_LOCK:
                e['live'] -= 1
        weakref.finalize(graph, _released)
    return graph


def release_graph_pool(device=None):
    """hand the shared volume-graph pool of `device` (default: the current one) back to the caching allocator -- for a process
    that ran inference and now wants the memory for training, or shares the GPU.  Refuses (returns False) while a volume
    graph of that device is alive; the next sliding-window job simply builds a new pool."""
    if device is None:
        key = torch.cuda.current_device()
    else:
        device = torch.device(device)
        key = device.index if device.index is not None else torch.cuda.current_device()
    with _GRAPH_POOL_LOCK:
        entry = _GRAPH_POOLS.get(key)
        if entry is None:
            return True
        if entry['live'] != 0:
            return False
        del _GRAPH_POOLS[key]
    entry.clear()                 # drops the keep-alive graph: the pool's segments become ordinary cached blocks
    torch.cuda.synchronize(key)
    torch.cuda.empty_cache()
    return True


def _sliding_window_inference(net, volume, starts, box, num_classes, normalizer, batch_size, use_graph, process_group,
                              shard, two_streams, gather):
    if gather not in ('all', 'mask', 'none'):
        raise ValueError("gather must be 'all', 'mask' or 'none'")
    batcher = SlidingWindowBatcher(volume, starts, box, num_classes, normalizer, max_batch=batch_size)
    P = batcher.max_batch
    sharded = shard and torch.distributed.is_available() and torch.distributed.is_initialized() and \
        torch.distributed.get_world_size(process_group) > 1
    plan, rank = None, 0
    if sharded:
        rank = torch.distributed.get_rank(process_group)
        plan = SlabShardPlan(batcher.starts, batcher.box, (batcher.Z, batcher.Y, batcher.X),
                             torch.distributed.get_world_size(process_group))
        mine = plan.patches[rank]
    else:
        mine = list(range(len(starts)))
    batcher.shard_plan = plan
    batches = [mine[i:i + P] for i in range(0, len(mine), P)]
    if batches:
        batcher.plan(batches)
    graph, first = None, 0
    # the weights do not change during a volume: keep their packed (MFMA-layout) images across batches, so neither
    # the eager batches nor the captured graph re-pack 26 tensors per forward
    cache_was_on = _ops.weight_cache(True)
    side = _job_stream(volume.device, 'side') if (two_streams and P >= 2) else None
    try:
        with torch.no_grad():
            if use_graph and len(batches) > 2:
                # warm-up on a side stream (allocator, lazy code-object loading, packed weights), then capture
                # gather -> net -> scatter once.  The warm-up batch is accumulated for real: it simply is the first
                # batch of the job.
                stream = _job_stream(volume.device, 'warmup')
                stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(stream):
                    batcher.select(0)
                    static_in = batcher.gather_current()
                    batcher.scatter_current(_forward_two_streams(net, static_in, side))
                torch.cuda.current_stream().wait_stream(stream)
                first = 1
                # capture with n_valid = 0 in the control block so the captured launch itself accumulates nothing
                batcher._ctl.zero_()
                torch.cuda.synchronize()
                graph = _capture_in_shared_pool(
                    lambda: batcher.scatter_current(_forward_two_streams(net, batcher.gather_current(out=static_in), side)),
                    volume.device)
            for b in range(first, len(batches)):
                batcher.select(b)
                if graph is not None:
                    graph.replay()
                else:
                    batcher.scatter_current(_forward_two_streams(net, batcher.gather_current(), side))
            if sharded:
                merge_slabs(batcher.acc, batcher.count, plan, rank, process_group)
                probs, mask = batcher.finalize(plan.owned(rank))
                if gather != 'none':
                    gather_slabs(probs, mask, plan, process_group, with_probs=(gather == 'all'))
            else:
                probs, mask = batcher.finalize()
    finally:
        if not cache_was_on:
            torch.cuda.synchronize()   # the images were allocated on the warm-up stream: nothing may still read them
            _ops.weight_cache(False)
    return probs, mask, batcher


# ---------------------------------------------------------------------------------------------------------------------
# reference-shaped API
# ---------------------------------------------------------------------------------------------------------------------
class _Model(dict):
    """attribute dict (the reference uses easydict here, core/seg_infer.py:107)"""
    __getattr__ = dict.get
    __setattr__ = dict.__setitem__


def load_single_model(model_folder, gpu_id=0):
    """load one model folder `<folder>/checkpoints/chk_<latest>/params.pth` (reference: seg_infer.py:99-164).
    gpu_id must be >= 0: this engine has no CPU path (the reference's gpu_id = -1 branch is served by stock torch)."""
    assert os.path.isdir(model_folder), 'Model folder does not exist: {}'.format(model_folder)
    if gpu_id is None or int(gpu_id) < 0:
        raise E.Seg3dEngineError('segmentation3d HIP engine needs gpu_id >= 0 (no CPU inference path)')
    device = torch.device('cuda:{}'.format(int(gpu_id)))
    # the reference pins the GPU with CUDA_VISIBLE_DEVICES (seg_infer.py:104-105); here the chosen device becomes the
    # CURRENT device: every kernel of this engine is launched on the current device's stream
    torch.cuda.set_device(device)
    chk_dir = get_checkpoint_folder(os.path.join(model_folder, 'checkpoints'), -1)
    state = torch.load(os.path.join(chk_dir, 'params.pth'), map_location='cpu', weights_only=True)
    net_module = importlib.import_module('segmentation3d.network.' + state['net'])
    net = net_module.SegmentationNet(state['in_channels'], state['out_channels'])
    net.load_state_dict(strip_module_prefix(state['state_dict']))
    net.eval()
    net = net.to(device)
    model = _Model()
    model.net = net
    model.device = device
    model.spacing, model.max_stride, model.interpolation = state['spacing'], state['max_stride'], state['interpolation']
    model.in_channels, model.out_channels = state['in_channels'], state['out_channels']
    model.crop_normalizers = [normalizer_from_dict(d) for d in state['crop_normalizers']]
    model.crop_normalizer_dicts = list(state['crop_normalizers'])
    return model


def load_models(model_folder, gpu_id=0):
    """load `infer_config.py` plus the coarse / fine models it names (reference: seg_infer.py:167-205)"""
    from segmentation3d.utils.file_io import load_config
    assert os.path.isdir(model_folder), 'Model folder does not exist: {}'.format(model_folder)
    infer_cfg = load_config(os.path.join(model_folder, 'infer_config.py'))
    models = _Model()
    models.infer_cfg = infer_cfg
    scale = infer_cfg.general.single_scale
    if scale not in ('coarse', 'fine', 'DISABLE'):
        raise ValueError('Unsupported single scale type!')
    models.coarse_model = models.fine_model = None
    if scale in ('coarse', 'DISABLE'):
        models.coarse_model = load_single_model(os.path.join(model_folder, infer_cfg.coarse.model_name), gpu_id)
    if scale in ('fine', 'DISABLE'):
        models.fine_model = load_single_model(os.path.join(model_folder, infer_cfg.fine.model_name), gpu_id)
    return models


def segmentation_voi(model, iso_image, start_voxel, end_voxel, use_gpu=True):
    """segment one volume of interest (reference: seg_infer.py:208-246); returns the list of per-class Image3d maps.
    Kept for API parity; whole volumes should go through segmentation_volume, which batches patches on the device."""
    vol = torch.from_numpy(np.ascontiguousarray(iso_image.array, dtype=np.float32)).to(model['device'])
    box = [int(end_voxel[d] - start_voxel[d]) for d in range(3)]
    norm = model['crop_normalizer_dicts'][0] if model['crop_normalizer_dicts'] else None
    probs, _, _ = sliding_window_inference(model['net'], vol, [list(start_voxel)], box, model['out_channels'], norm,
                                           batch_size=1, use_graph=False)
    z0, y0, x0 = start_voxel[2], start_voxel[1], start_voxel[0]
    maps = []
    for c in range(model['out_channels']):
        roi = probs[c, z0:z0 + box[2], y0:y0 + box[1], x0:x0 + box[0]].cpu().numpy()
        maps.append(Image3d(roi, iso_image.GetSpacing(), iso_image.GetOrigin(), iso_image.GetDirection()))
    return maps


def _physical_to_index(frame, point):
    """sitk TransformPhysicalPointToIndex: nearest index (round half up per axis)"""
    spacing, origin, direction = (np.asarray(v, dtype=np.float64) for v in frame)
    c = np.diag(1.0 / spacing) @ np.linalg.inv(direction.reshape(3, 3)) @ (np.asarray(point, dtype=np.float64) - origin)
    return [int(np.floor(v + 0.5)) for v in c]


def _index_to_physical(frame, index):
    spacing, origin, direction = (np.asarray(v, dtype=np.float64) for v in frame)
    return origin + direction.reshape(3, 3) @ (spacing * np.asarray(index, dtype=np.float64))


def segmentation_volume(model, cfg, image, bbox_start_voxel, bbox_end_voxel, use_gpu=True, batch_size=8):
    """segment a whole volume (reference: seg_infer.py:249-350), everything between the upload of the image and the
    download of the probability maps / mask on the device:
      resample to the model spacing, size up to a multiple of max_stride (image_tools.py:348-377)  -> seg3d_resample_affine
      partition (host index arithmetic), one batched gather -> net -> scatter per hipGraph replay, divide by overlap
      resample the class probabilities back onto the image grid, padding 1.0 for class 0 and 0.0 otherwise (:330-333)
      arg-max -> int8 mask (:336-339); largest component / small-component removal (:342-348)
    `use_gpu` is kept for signature compatibility (the reference shrinks spacing / partitions on the CPU path only).
    Returns (mean_probs: list of Image3d, mask: Image3d int8).
    """
    assert isinstance(image, Image3d)
    with torch.cuda.device(model['device']):
        return _segmentation_volume(model, cfg, image, bbox_start_voxel, bbox_end_voxel, batch_size)


def _segmentation_volume(model, cfg, image, bbox_start_voxel, bbox_end_voxel, batch_size):
    from segmentation3d.utils import image_tools
    dev = model['device']
    ms = int(model['max_stride'])
    num_classes = int(model['out_channels'])
    spacing = [float(s) for s in model['spacing']]
    img_frame = (image.GetSpacing(), image.GetOrigin(), image.GetDirection())
    iso_frame = (spacing, image.GetOrigin(), image.GetDirection())
    X, Y, Z = image.GetSize()
    Xp, Yp, Zp = image_tools.resampled_size((X, Y, Z), image.GetSpacing(), spacing, ms)
    src = torch.from_numpy(np.array(image.array, dtype=np.float32, order='C')).to(dev)
    interp = model.get('interpolation', 'LINEAR') or 'LINEAR'
    vol = image_tools.resample_device(src, img_frame, (Xp, Yp, Zp), iso_frame, interp, 0.0)
    if cfg.partition_type == 'DISABLE':
        starts, box = [[0, 0, 0]], (Xp, Yp, Zp)
    elif cfg.partition_type == 'SIZE':
        if bbox_start_voxel is not None and bbox_end_voxel is not None:
            # bounding box given in image voxels -> iso grid (seg_infer.py:292-303)
            s0 = _physical_to_index(iso_frame, _index_to_physical(img_frame, [float(v) for v in bbox_start_voxel]))
            e0 = _physical_to_index(iso_frame, _index_to_physical(img_frame, [float(v) for v in bbox_end_voxel]))
            s0 = [max(0, v) for v in s0]
            e0 = [min(v, lim) for v, lim in zip(e0, (Xp, Yp, Zp))]
        else:
            s0, e0 = [0, 0, 0], [Xp, Yp, Zp]
        starts, ends = image_partition_by_fixed_size(((Xp, Yp, Zp), spacing), s0, e0, list(cfg.partition_size),
                                                     list(cfg.partition_stride), ms)
        box = tuple(ends[0][d] - starts[0][d] for d in range(3))
    else:
        raise ValueError('Unsupported partition type!')
    norm = model['crop_normalizer_dicts'][0] if model['crop_normalizer_dicts'] else None
    probs, _, batcher = sliding_window_inference(model['net'], vol, starts, box, num_classes, norm,
                                                 batch_size=min(batch_size, max(1, len(starts))))
    # (voxels no patch covered -- bounding-box runs -- have count 0: their probabilities are 0 and the arg-max there is
    # class 0, as with ITK's division in the reference, see finalize_argmax_kernel)
    # back to the image grid (identity when the image already is at the model spacing and a stride multiple)
    same_grid = (Xp, Yp, Zp) == (X, Y, Z) and all(abs(a - b) <= 1e-9 * max(abs(a), abs(b), 1.0)
                                                 for a, b in zip(image.GetSpacing(), spacing))
    if same_grid:
        out_probs = probs
    else:
        out_probs = torch.stack([image_tools.resample_device(probs[c], iso_frame, (X, Y, Z), img_frame, 'LINEAR',
                                                              1.0 if c == 0 else 0.0) for c in range(num_classes)])
    mask = out_probs.argmax(0).to(torch.int8)       # first maximum wins, like tensor.max(0) in the reference
    labels = list(range(1, num_classes))
    if getattr(cfg, 'pick_largest_cc', False) and labels:
        mask = image_tools.connected_component_filter_device(mask, labels, 'largest')
    if getattr(cfg, 'remove_small_cc', 0) and cfg.remove_small_cc > 0 and labels:
        mask = image_tools.connected_component_filter_device(mask, labels, 'min_size', int(cfg.remove_small_cc))
    mean_probs = [Image3d(out_probs[c].cpu().numpy(), *img_frame) for c in range(num_classes)]
    return mean_probs, Image3d(mask.cpu().numpy(), *img_frame)


_IMAGE_SUFFIXES = ('.mhd', '.nii', '.hdr', '.nii.gz', '.mha', '.image3d')     # core/seg_infer.py:80, 375-376
_READABLE_SUFFIXES = ('.mha', '.mhd', '.nii', '.nii.gz')


def read_test_txt(txt_file):
    """single-modality list file: first line = number of cases, then `<case name> <image path>` per line
    (reference: seg_infer.py:23-45)"""
    from segmentation3d.utils.file_io import readlines
    lines = readlines(txt_file)
    case_num = int(lines[0])
    if len(lines) - 1 != case_num:
        raise ValueError('case num do not equal path num!')
    names, paths = [], []
    for line in lines[1:1 + case_num]:
        parts = line.strip().split()
        if len(parts) < 2:
            raise ValueError('expected "<case name> <image path>", got: {}'.format(line))
        if not os.path.isfile(parts[1]):
            raise ValueError('image not exist: {}'.format(parts[1]))
        names.append(parts[0])
        paths.append(parts[1])
    return names, paths


def read_test_folder(folder_path):
    """every image file of a folder, case name = file name up to the image suffix (reference: seg_infer.py:68-96; DICOM
    series folders are out of scope here)"""
    import glob
    files = []
    for suf in _IMAGE_SUFFIXES:
        files += glob.glob(os.path.join(folder_path, '*' + suf))
    names, paths = [], []
    for path in sorted(set(files)):
        name = os.path.basename(path)
        for suf in _IMAGE_SUFFIXES:
            idx = name.find(suf)
            if idx != -1:
                name = name[:idx]
                break
        names.append(name)
        paths.append(path)
    return names, paths


def segmentation(input_path, model_folder, output_folder, seg_name, gpu_id, return_mask, save_mask, save_image,
                 save_prob):
    """volumetric image segmentation engine for image files (reference: seg_infer.py:353-493): single-scale
    ('coarse' / 'fine') or the coarse -> fine cascade ('DISABLE') through the coarse mask's bounding box.
    input_path: a list file (.txt), one image file, or a folder of image files; results go to
    `<output_folder>/<case name>/` with the reference's file names (seg_name, org.mha, mean_prob_<c>.mha)."""
    from segmentation3d.utils.image_io import read_image, write_image
    begin = time.time()
    models = load_models(model_folder, gpu_id)
    load_model_time = time.time() - begin
    if os.path.isfile(input_path):
        if input_path.endswith('.txt'):
            names, paths = read_test_txt(input_path)
        elif input_path.endswith(_IMAGE_SUFFIXES):
            names, paths = [os.path.basename(input_path)], [input_path]        # seg_infer.py:377-379: name = file name
        else:
            raise ValueError('Unsupported input path.')
    elif os.path.isdir(input_path):
        names, paths = read_test_folder(input_path)
        if len(names) == 0:
            raise ValueError('Empty test folder!')
    else:
        raise ValueError('The file {} does not exist.'.format(input_path))
    for path in paths:
        if not path.endswith(_READABLE_SUFFIXES):
            raise ValueError('Unsupported image format (MetaImage and NIfTI are read here): {}'.format(path))
    scale = models['infer_cfg'].general.single_scale
    if scale not in ('coarse', 'fine', 'DISABLE'):
        raise ValueError('Unsupported scale type!')
    masks, total = [], 0.0
    for i, path in enumerate(paths):
        print('{}: {}'.format(i, path))
        begin = time.time()
        image = read_image(path)
        if image.array.dtype != np.float32:                                     # sitk.ReadImage(path, sitk.sitkFloat32)
            image = image.like(image.array.astype(np.float32))
        read_image_time = time.time() - begin
        begin = time.time()
        if scale == 'coarse':
            mean_probs, mask = segmentation_volume(models['coarse_model'], models['infer_cfg'].coarse, image, None, None, True)
        elif scale == 'fine':
            mean_probs, mask = segmentation_volume(models['fine_model'], models['infer_cfg'].fine, image, None, None, True)
        else:
            # coarse -> fine cascade (seg_infer.py:428-444): the coarse mask's bounding box restricts the fine pass
            from segmentation3d.utils.image_tools import get_bounding_box
            print('Coarse segmentation: ')
            _, mask = segmentation_volume(models['coarse_model'], models['infer_cfg'].coarse, image, None, None, True)
            start_voxel, end_voxel = get_bounding_box(mask, None)
            if start_voxel is None:
                start_voxel, end_voxel = [0, 0, 0], list(mask.GetSize())
            bbox_ratio = 100
            for idx in range(3):
                bbox_ratio *= (end_voxel[idx] - start_voxel[idx]) / mask.GetSize()[idx]
            print('Fine segmentation (bbox ratio: {:.2f}%): '.format(bbox_ratio))
            mean_probs, mask = segmentation_volume(models['fine_model'], models['infer_cfg'].fine, image, start_voxel,
                                                   end_voxel, True)
        torch.cuda.synchronize()
        inference_time = time.time() - begin
        total += inference_time
        if return_mask:
            masks.append(mask)
        begin = time.time()
        case = names[i]
        if save_mask or save_image or save_prob:
            os.makedirs(os.path.join(output_folder, case), exist_ok=True)
        if save_mask:
            write_image(mask, os.path.join(output_folder, case, seg_name))
        if save_image:
            write_image(image, os.path.join(output_folder, case, 'org.mha'))
        if save_prob:
            for c, p in enumerate(mean_probs):
                write_image(p, os.path.join(output_folder, case, 'mean_prob_{}.mha'.format(c)))
        save_time = time.time() - begin
        print('total test time: {:.2f}, average inference time: {:.2f}'.format(
            load_model_time + read_image_time + inference_time + save_time, total / (i + 1)))
    return masks
