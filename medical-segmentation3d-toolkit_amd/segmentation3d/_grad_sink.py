"""Gradient sinks: parameter gradients written by the backward kernels straight into the optimizer's flat buffer.

`torch.optim.Adam` in the reference (core/seg_train.py:83,125-127) receives its gradients through autograd's
AccumulateGrad nodes: every weight-gradient kernel writes a fresh tensor and a second elementwise kernel adds it into
`p.grad` -- 116 tiny launches per V-Net step.  FusedAdam owns one flat gradient buffer, so it registers a *sink* (the
parameter's slice of that buffer) for every parameter; the fused autograd functions in `_ops` look the sink up in
forward and their weight-gradient / GroupNorm-finalize kernels accumulate into it directly (`accumulate` flag of the
C ABI), returning None to autograd for that input.

Consequences, by construction:
  * `p.grad` IS the sink view, so after `loss.backward()` it holds exactly what AccumulateGrad would have produced
    (the buffer is zeroed by `optimizer.zero_grad()`; several backward passes accumulate);
  * autograd still runs the parameter's AccumulateGrad node with an undefined gradient (a no-op) and then its
    post-accumulate hooks, after the producing kernel was enqueued -- the data-parallel bucket reducer
    (core/ddp.py) keeps using those hooks as its "gradient ready" signal;
  * `torch.autograd.grad(loss, params)` would see None for a sunk parameter: construct FusedAdam with
    `direct_grads=False` (or call `unregister`) for that use.
This module has no device dependency (the CPU/gloo tests import it).
"""
import weakref

_SINKS = {}  # id(param) -> _Sink


class _Sink(object):
    __slots__ = ('ref', 'view')

    def __init__(self, param, view):
        self.ref = weakref.ref(param)
        self.view = view


def register(param, view):
    """route `param`'s gradient into `view` (same shape, contiguous, fp32, on the parameter's device)"""
    if tuple(view.shape) != tuple(param.shape) or not view.is_contiguous():
        raise ValueError('gradient sink must be a contiguous tensor of the parameter shape')
    sink = _Sink(param, view)
    _SINKS[id(param)] = sink
    return sink


def unregister(params):
    for p in params:
        s = _SINKS.get(id(p))
        if s is not None and s.ref() is p:
            del _SINKS[id(p)]


def lookup(t):
    """the sink registered for exactly this tensor object, or None"""
    if t is None or not _SINKS:
        return None
    s = _SINKS.get(id(t))
    if s is None:
        return None
    if s.ref() is not t:          # id reuse after the parameter died
        del _SINKS[id(t)]
        return None
    return s
