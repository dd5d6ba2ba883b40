"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI, overlapped with backward.

Replaces `nn.parallel.DataParallel` (core/seg_train.py:76-78 of the reference), which re-broadcasts the parameters,
scatters the batch, gathers the outputs and reduces the gradients to GPU 0 every step inside ONE process.  Here
every rank owns a full replica and its own 4-patch batch; the only exchange is one sum all-reduce of the flat fp32
gradient buffer (58.3 MB for the 14.56 M-parameter V-Net) per step:

  * gradients live in ONE contiguous buffer (FusedAdam's flat layout), so a bucket is just a slice -- no packing;
  * buckets are cut in reverse parameter order (the decoder/head gradients are produced first in backward) and each
    bucket's all-reduce is enqueued from a post-accumulate hook as soon as its last gradient has been written, so the
    transfer overlaps the rest of backward (RCCL runs on its own stream, ordered after the producing kernels);
  * xGMI is a point-to-point full mesh (7 links/GPU): a few large buckets (default 4 x ~16 MB) keep every link busy
    without paying per-collective latency 116 times; the bucket that completes last is split so that only ~1 MB of it
    is reduced after backward has ended; the 1/world scaling is folded into the Adam kernel.
GroupNorm(1, C) statistics are per sample, Dice is per sample then batch mean and Focal is a mean over voxels, so with
equal per-rank batches the averaged gradient equals the global-batch gradient: no other collective is needed.
"""
import torch
import torch.distributed as dist


class FlatGradients(object):
    """one contiguous gradient buffer for a list of parameters (used when the optimizer is not FusedAdam, e.g. the
    CPU/gloo tests); `p.grad` becomes a view into it"""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.offsets, total = [], 0
        for p in self.params:
            self.offsets.append(total)
            total += p.numel()
        ref = self.params[0]
        self.buffer = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        self.attach()

    def attach(self):
        for p, off in zip(self.params, self.offsets):
            p.grad = self.buffer[off:off + p.numel()].view(p.shape)

    def zero(self):
        self.buffer.zero_()
        self.attach()

    def layout(self):
        return [(p, 0, off, p.numel()) for p, off in zip(self.params, self.offsets)]


def bucket_cuts(entries, total, num_buckets=4, tail_elems=1 << 18):
    """[(lo, hi, [params])] in the order the buckets complete in backward, for one flat buffer whose `entries` are
    (param, offset, numel) in buffer order.  Byte-balanced cuts at parameter boundaries walking from the END of the buffer
    (those gradients arrive first), then the last-completing bucket is split:

    the bucket at the START of the buffer completes with the very last weight gradient of backward (the stem's), so its
    all-reduce is the one transfer nothing can hide.  The parameters whose gradients arrive last (about the first
    `tail_elems` elements of the buffer: stem and the first encoder stages, 0.6 MB for the V-Net) form a small bucket of their
    own, and the rest of the old bucket (the deeper encoder stages: 8-14 MB) is reduced while those last stages are still in
    backward.  tail_elems = 0, or a bucket that is not at least twice that size: no split."""
    target = max(1, total // max(1, num_buckets))
    cuts = []
    hi, acc, members = total, 0, []
    for p, off, n in reversed(entries):
        members.append(p)
        acc += n
        if acc >= target and len(cuts) < num_buckets - 1:
            cuts.append((off, hi, members))
            hi, acc, members = off, 0, []
    if members:
        cuts.append((0, hi, members))
    if tail_elems and cuts and cuts[-1][0] == 0 and cuts[-1][1] >= 2 * tail_elems:
        lo, hi_last, mem = cuts[-1]                 # mem is in reverse buffer order: mem[-1] is the first parameter
        offs = {id(p): off for p, off, n in entries}
        split = next((offs[id(p)] for p in mem if offs[id(p)] <= tail_elems), None)   # largest offset <= tail_elems
        if split is not None and 0 < split < hi_last:
            head = [p for p in mem if offs[id(p)] >= split]
            tail = [p for p in mem if offs[id(p)] < split]
            cuts[-1:] = [(split, hi_last, head), (0, split, tail)]
    return cuts


class GradientReducer(object):
    """bucketed, backward-overlapped sum all-reduce of flat gradient buffers.

    :param flat_buffers: list of flat gradient tensors (one per parameter group)
    :param layout: [(param, buffer index, offset, numel)] in buffer order
    :param num_buckets: buckets per buffer (cut by bytes, aligned to parameter boundaries)
    :param tail_elems: the bucket that completes last is split so that its final part holds at most about this many
                       elements (0 = no split); see the comment in __init__
    """

    def __init__(self, flat_buffers, layout, process_group=None, num_buckets=4, tail_elems=1 << 18):
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self.group = process_group
        self._join, self._side = None, None
        if flat_buffers and flat_buffers[0].is_cuda:
            from segmentation3d import _ops
            self._join = _ops.wgrad_stream_join
            if dist.get_backend(process_group) == 'nccl':     # (gloo stages through the host: it needs the join)
                self._side = _ops.wgrad_side_stream
            # choose the side stream now: its probe synchronises the device, which has no place inside the first gradient hook
            with torch.cuda.device(flat_buffers[0].device):
                _ops.prepare_side_stream(flat_buffers[0].device)
        self.world_size = dist.get_world_size(process_group)
        self.buffers = flat_buffers
        self._buckets = []          # dict(buffer, lo, hi, pending, nparams, work)
        self._param_bucket = {}
        for bi, buf in enumerate(flat_buffers):
            entries = [(p, off, n) for p, b, off, n in layout if b == bi]
            if not entries:
                continue
            cuts = bucket_cuts(entries, buf.numel(), num_buckets, tail_elems)
            for lo, hi_b, mem in cuts:
                self._add_bucket(bi, lo, hi_b, mem)
        # autograd runs a leaf's AccumulateGrad node (and this hook) once all of its uses have been processed, also
        # when the producing function returned None because its kernel wrote the gradient straight into the flat
        # buffer (gradient sinks, _grad_sink.py) -- so the hook is the "gradient ready" signal in both modes.  Should a
        # hook not fire, finish_step() reduces the bucket after backward (correct, just not overlapped).
        self._handles = []
        for p in self._param_bucket:
            self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad_ready))
        self._active = False
        self.launched_in_backward = 0   # buckets whose all-reduce was enqueued from a hook during the last backward

    def _buckets_for(self, bi):
        return [b for b in self._buckets if b['buffer'] == bi]

    def _add_bucket(self, bi, lo, hi, members):
        idx = len(self._buckets)
        self._buckets.append({'buffer': bi, 'lo': lo, 'hi': hi, 'nparams': len(members), 'pending': len(members),
                              'work': None})
        for p in members:
            self._param_bucket[p] = idx

    def bucket_sizes(self):
        return [b['hi'] - b['lo'] for b in self._buckets]

    # ---- per step ----------------------------------------------------------------------------------------------------
    def begin_step(self):
        """call before backward (after zero_grad)"""
        for b in self._buckets:
            b['pending'], b['work'] = b['nparams'], None
        self.launched_in_backward = 0
        self._active = True

    def _launch(self, b):
        view = self.buffers[b['buffer']][b['lo']:b['hi']]
        side = self._side() if self._side is not None else None
        if side is not None:
            # A bucket holds weight gradients (written on the weight-gradient side stream, segmentation3d._ops) and GroupNorm /
            # bias gradients (written on the main stream).  The collective is enqueued with the SIDE stream current, after
            # that stream has been told to wait for the main stream's present position: RCCL then orders itself behind both,
            # and the main stream -- the data-gradient chain, backward's critical path -- is not made to wait for the pending
            # weight gradients at every bucket launch (one-rank RCCL run, tools/ddp_overhead.py: 16.38 -> see DESIGN.md 5).
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                b['work'] = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            return
        if self._join is not None:
            self._join()    # weight gradients written on the side stream (segmentation3d._ops) must have landed
        b['work'] = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _on_grad_ready(self, param):
        if not self._active:
            return
        b = self._buckets[self._param_bucket[param]]
        b['pending'] -= 1
        if b['pending'] == 0 and b['work'] is None:
            self._launch(b)
            self.launched_in_backward += 1

    def finish_step(self):
        """call after backward: reduces buckets whose hooks did not all fire, then waits (stream-ordered on GPUs).
        Gradients hold the SUM over ranks afterwards; divide by `world_size` (FusedAdam.grad_scale does it in-kernel)."""
        self._active = False
        for b in self._buckets:
            if b['work'] is None:
                self._launch(b)
        for b in self._buckets:
            b['work'].wait()
            b['work'] = None

    def broadcast_parameters(self, flat_params_or_list, src=0):
        """make every replica start from rank `src`'s parameters"""
        tensors = flat_params_or_list if isinstance(flat_params_or_list, (list, tuple)) else [flat_params_or_list]
        for t in tensors:
            dist.broadcast(t, src=src, group=self.group)

    def remove_hooks(self):
        for h in self._handles:
            h.remove()
        self._handles = []
