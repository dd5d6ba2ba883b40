"""FusedAdam -- `torch.optim.Adam` semantics (core/seg_train.py:83,127 of the reference: Adam(lr, betas), eps 1e-8,
no weight decay, no amsgrad) executed as ONE HIP kernel over a flat parameter buffer.

MI355X-first layout: all parameters of a group live in one contiguous fp32 buffer (likewise gradients, exp_avg,
exp_avg_sq); `p.data` / `p.grad` are views into it.  One launch updates 14.56 M parameters (28 B/param of HBM traffic)
and the data-parallel gradient all-reduce works on contiguous slices of the same buffer (core/ddp.py).
With `direct_grads` (default) every parameter's slice of the gradient buffer is registered as a gradient sink
(`_grad_sink.py`): the backward kernels accumulate into it themselves instead of autograd adding a temporary.
`state_dict()` / `load_state_dict()` keep torch.optim.Adam's structure (`step`, `exp_avg`, `exp_avg_sq` per
parameter), so `optimizer.pth` files are interchangeable with the reference's.
"""
import torch

from segmentation3d import _engine as E
from segmentation3d import _grad_sink as G

_ALIGN = 64  # floats; keeps every parameter view 256-byte aligned (kernels read gamma/beta/weights with 16-byte loads)


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, direct_grads=True):
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError('invalid Adam hyper-parameters')
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        super(FusedAdam, self).__init__(params, defaults)
        self._flat = []  # per group: dict(params, grads, exp_avg, exp_avg_sq, offsets, step)
        self.grad_scale = 1.0  # set to 1/world_size by the data-parallel wrapper after a sum all-reduce
        self.direct_grads = bool(direct_grads)
        # device_step: the step count lives on the device and the kernel advances it itself (seg3d_adam_step_devstep), so
        # that a whole train step can be captured in a hipGraph and replayed; the host count follows in note_replayed_step()
        self.device_step = False
        for group in self.param_groups:
            self._flat.append(self._flatten_group(group))

    # ---- flat buffers ------------------------------------------------------------------------------------------
    def _flatten_group(self, group):
        ps = [p for p in group['params'] if p.requires_grad]
        if not ps:
            return None
        dev = ps[0].device
        for p in ps:
            if p.device != dev or p.dtype != torch.float32:
                raise ValueError('FusedAdam needs all parameters in float32 on one device')
        E.require_device(ps[0])
        offsets, total = [], 0
        for p in ps:
            offsets.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, off in zip(ps, offsets):
            n = p.numel()
            flat_p[off:off + n].copy_(p.data.reshape(-1))
            p.data = flat_p[off:off + n].view(p.shape)
            if p.grad is not None:
                flat_g[off:off + n].copy_(p.grad.reshape(-1))
            p.grad = flat_g[off:off + n].view(p.shape)
            if self.direct_grads:
                G.register(p, p.grad)
            self.state[p] = {'step': torch.tensor(0.0), 'exp_avg': flat_m[off:off + n].view(p.shape),
                             'exp_avg_sq': flat_v[off:off + n].view(p.shape)}
        return {'list': ps, 'offsets': offsets, 'params': flat_p, 'grads': flat_g, 'exp_avg': flat_m,
                'exp_avg_sq': flat_v, 'step': 0, 'total': total}

    def use_device_step(self):
        """switch to the device-resident step counter (before capturing a train step in a hipGraph); idempotent"""
        for f in self._flat:
            if f is None:
                continue
            dev = f['params'].device
            f['step_dev'] = torch.full((1,), int(f['step']), dtype=torch.int32, device=dev)
            f['bc_dev'] = torch.zeros(2, dtype=torch.float32, device=dev)
        self.device_step = True

    def note_replayed_step(self):
        """a captured step was replayed: the device advanced its counter, bring the host-side bookkeeping along"""
        for f in self._flat:
            if f is None:
                continue
            f['step'] += 1
            step_t = torch.tensor(float(f['step']))
            for p in f['list']:
                self.state[p]['step'] = step_t

    def flat_grads(self):
        """list of flat gradient buffers (one per parameter group) -- what the data-parallel reducer all-reduces"""
        return [f['grads'] for f in self._flat if f is not None]

    def flat_layout(self):
        """[(parameter, group index, offset, numel)] in buffer order"""
        out = []
        for gi, f in enumerate(self._flat):
            if f is None:
                continue
            for p, off in zip(f['list'], f['offsets']):
                out.append((p, gi, off, p.numel()))
        return out

    # ---- optimizer API -------------------------------------------------------------------------------------------
    def zero_grad(self, set_to_none=False):
        """zero the flat gradient buffer (one memset) and keep `p.grad` pointing into it"""
        for f in self._flat:
            if f is None:
                continue
            f['grads'].zero_()
            for p, off in zip(f['list'], f['offsets']):
                n = p.numel()
                if p.grad is None or p.grad.data_ptr() != f['grads'].data_ptr() + 4 * off:
                    p.grad = f['grads'][off:off + n].view(p.shape)
                    if self.direct_grads:
                        G.register(p, p.grad)

    def _gather_stray_grads(self, f):
        for p, off in zip(f['list'], f['offsets']):
            n = p.numel()
            view = f['grads'][off:off + n]
            if p.grad is None:
                if not self.direct_grads:
                    view.zero_()          # with sinks the kernels wrote here even though autograd never set .grad
                p.grad = view.view(p.shape)
            elif p.grad.data_ptr() != view.data_ptr():
                # a stray tensor (someone assigned p.grad): what autograd put there joins what the sinks wrote
                if self.direct_grads:
                    view.add_(p.grad.reshape(-1))
                else:
                    view.copy_(p.grad.reshape(-1))
                p.grad = view.view(p.shape)
            if p.data.data_ptr() != f['params'].data_ptr() + 4 * off:
                # someone re-assigned p.data (e.g. load_state_dict keeps storage, .to() does not): re-adopt it
                f['params'][off:off + n].copy_(p.data.reshape(-1))
                p.data = f['params'][off:off + n].view(p.shape)
                from segmentation3d import _ops
                _ops.PACK_CACHE.invalidate()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        touched = False
        for group, f in zip(self.param_groups, self._flat):
            if f is None:
                continue
            self._gather_stray_grads(f)
            f['step'] += 1
            beta1, beta2 = group['betas']
            if self.device_step:
                E.call('seg3d_adam_step_devstep', E.ptr(f['params']), E.ptr(f['grads']), E.ptr(f['exp_avg']),
                       E.ptr(f['exp_avg_sq']), f['total'], E.ptr(f['step_dev']), E.ptr(f['bc_dev']), float(group['lr']),
                       float(beta1), float(beta2), float(group['eps']), float(group['weight_decay']),
                       float(self.grad_scale), E.stream_ptr())
            else:
                E.call('seg3d_adam_step', E.ptr(f['params']), E.ptr(f['grads']), E.ptr(f['exp_avg']),
                       E.ptr(f['exp_avg_sq']), f['total'], f['step'], float(group['lr']), float(beta1), float(beta2),
                       float(group['eps']), float(group['weight_decay']), float(self.grad_scale), E.stream_ptr())
            touched = True
            step_t = torch.tensor(float(f['step']))     # one host tensor shared by the group's per-parameter states
            for p in f['list']:
                self.state[p]['step'] = step_t
        if touched:
            # the kernel rewrote the parameters without touching their version counters: refresh (one launch) or
            # invalidate the packed conv-weight images
            from segmentation3d import _ops
            if _ops.PACK_CACHE.enabled:
                _ops.PACK_CACHE.repack_all()
            else:
                _ops.PACK_CACHE.invalidate()
        return loss

    def release_grad_sinks(self):
        """stop routing gradients into the flat buffer (e.g. before using torch.autograd.grad on these parameters)"""
        self.direct_grads = False
        for f in self._flat:
            if f is not None:
                G.unregister(f['list'])

    def load_state_dict(self, state_dict):
        super(FusedAdam, self).load_state_dict(state_dict)
        # torch replaced the state tensors with loaded copies: move them back into the flat buffers
        for f in self._flat:
            if f is None:
                continue
            step = 0
            for p, off in zip(f['list'], f['offsets']):
                n = p.numel()
                st = self.state.get(p, None)
                if st is None or 'exp_avg' not in st:
                    continue
                f['exp_avg'][off:off + n].copy_(st['exp_avg'].reshape(-1).to(f['exp_avg'].device))
                f['exp_avg_sq'][off:off + n].copy_(st['exp_avg_sq'].reshape(-1).to(f['exp_avg'].device))
                st['exp_avg'] = f['exp_avg'][off:off + n].view(p.shape)
                st['exp_avg_sq'] = f['exp_avg_sq'][off:off + n].view(p.shape)
                step = max(step, int(float(st['step'])))
                st['step'] = torch.tensor(float(step))
            f['step'] = step
            if self.device_step and 'step_dev' in f:
                f['step_dev'].fill_(int(step))        # a captured step reads its count from the device
