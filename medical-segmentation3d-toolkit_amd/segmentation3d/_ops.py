"""Autograd-visible operators of the HIP engine.

Every operator here enqueues hand-written gfx950 kernels from libseg3d_hip.so on torch's current stream; torch is
used for device memory (caching allocator), streams and the autograd graph only.  Activations travel between
operators as *logical* NCDHW tensors whose memory is NDHWC (a permuted view of a contiguous [N,D,H,W,C] buffer), so
module boundaries keep the reference's tensor shapes while the kernels see channels-last rows.

Fused unit (the reference's `ConvGnRelu3`, network/module/conv_gn_relu3.py:4-20, and the GN+ReLU tails of
InputBlock/DownBlock/UpBlock/OutputBlock):   out = act( GN_1(conv(x) + b) [+ residual] )
"""
import contextlib
import ctypes

import torch

from . import _engine as E
from . import _grad_sink as G

GN_EPS = 1e-5  # nn.GroupNorm default (network/module/conv_gn_relu3.py:11)

# conv kinds: (ksize, stride, ntaps, transposed)
_KINDS = {'k3': (3, 1, 27, False), 'k2s2': (2, 2, 8, False), 'k1': (1, 1, 1, False), 'convT': (2, 2, 8, True)}

# set to True to force the VALU kernels everywhere (used by tests to cross-check the MFMA path)
FORCE_DIRECT = False
# C -> C 3x3x3 forward / data-gradient at the big levels: Winograd F(2, 3) along x (False: the direct implicit GEMM everywhere)
WINOGRAD = True
WINOGRAD2D = True    # F(2x2, 3x3) over (y, x) where it is preferred, else F(2, 3) along x
STREAM_K = True      # F(2x2, 3x3) tile kernel: stream-K chunk ranges where whole items would fill the CUs' rounds badly (432 / 864 items)
_TWO_FORWARDS = 0    # > 0 inside two_forwards(): see there


@contextlib.contextmanager
def two_forwards():
    """two forwards are being enqueued on two streams (core/seg_infer._forward_two_streams): the CUs a launch leaves idle in its
    last round are taken by the other stream's kernels, so stream-K has nothing to win there and its finish pass is pure cost
    (measured: 512x512x400 job 0.843 s with whole items, 0.857 s with stream-K; the one-stream train step 14.64 -> 14.51 ms WITH
    it).  The choice is made when a launch is enqueued, so it is part of a captured hipGraph."""
    global _TWO_FORWARDS
    _TWO_FORWARDS += 1
    try:
        yield
    finally:
        _TWO_FORWARDS -= 1

# bf16 mode (BASELINE config 5): activations between fused units and the gradients handed to the conv kernels are bf16,
# packed k3 weights are bf16 (fp32 master weights), accumulation / conv outputs / GroupNorm statistics / losses fp32.
# The reference has no such switch (it computes in fp32 throughout); the default here is fp32 as well.
_ACT_BF16 = [False]
# bf16 mode, second switch: the raw conv output y (read by GroupNorm forward and backward, saved for backward) is bf16 as
# well -- rounded by the conv epilogue AFTER the fp32 (sum, sumsq) statistics are taken.  BF16_CONV_OUTPUT = False keeps y in fp32 (tests/test_gpu_bf16.py runs both).
BF16_CONV_OUTPUT = True


def set_activation_dtype(name):
    """'fp32' (default, the reference's arithmetic) or 'bf16'; returns the previous setting"""
    prev = 'bf16' if _ACT_BF16[0] else 'fp32'
    if name not in ('fp32', 'bf16'):
        raise ValueError("activation dtype must be 'fp32' or 'bf16', got {!r}".format(name))
    _ACT_BF16[0] = name == 'bf16'
    return prev


def activation_dtype_name():
    """'fp32' or 'bf16': the current activation storage mode"""
    return 'bf16' if _ACT_BF16[0] else 'fp32'


class activation_dtype(object):
    """context manager form of set_activation_dtype"""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.prev = set_activation_dtype(self.name)

    def __exit__(self, *exc):
        set_activation_dtype(self.prev)


def _out_bf16(C):
    """does a unit with C output channels emit bf16 activations in the current mode (thin heads stay fp32)"""
    return _ACT_BF16[0] and C >= 16 and C % 16 == 0 and not FORCE_DIRECT


def _is_bf16(t):
    return t is not None and t.dtype == torch.bfloat16


# ------------------------------------------------------------------------------------------------------------------
# layout helpers
# ------------------------------------------------------------------------------------------------------------------
def to_ndhwc(x, allow_slice=False):
    """logical NCDHW tensor -> contiguous [N,D,H,W,C] fp32 tensor (zero-copy when the memory is already NDHWC).
    allow_slice: a channel slice of a wider NDHWC buffer (rows of C floats at a constant stride > C) is returned as the strided
    view it is instead of being repacked.  Only callers whose kernels take a row stride may ask for it -- the stride-2 conv of
    an inference forward and up_cat's skip; every other consumer passes E.ptr() to a kernel that assumes C-contiguous rows."""
    E.require_device(x)
    if x.dim() != 5:
        raise ValueError('expected a 5-D [N,C,D,H,W] tensor, got shape {}'.format(tuple(x.shape)))
    if x.dtype == torch.bfloat16:      # bf16 mode: activations travel between units already in NDHWC memory
        xp = x.permute(0, 2, 3, 4, 1)
        return xp if xp.is_contiguous() else xp.contiguous()
    if x.dtype != torch.float32:
        raise TypeError('segmentation3d HIP engine computes in float32 (or bf16 activations in bf16 mode), got {}'
                        .format(x.dtype))
    xp = x.permute(0, 2, 3, 4, 1)
    if xp.is_contiguous():
        return xp
    if _is_channel_slice(xp):
        # a channel slice of a wider NDHWC buffer (inference: an encoder feature living in its half of a decoder's concatenated
        # skip buffer, VNetBase.forward; training: the skip gradient UpCatFunction.backward hands back without a link): the two
        # consumers that can meet one in place -- the stride-2 conv and up_cat -- ask for it, everyone else gets a packed copy
        return xp if allow_slice else xp.contiguous()
    N, C, D, H, W = x.shape
    xc = x.contiguous()
    out = torch.empty((N, D, H, W, C), dtype=torch.float32, device=x.device)
    E.call('seg3d_ncdhw_to_ndhwc', E.ptr(xc), E.ptr(out), N, C, D * H * W, E.stream_ptr())
    return out


def _is_channel_slice(t):
    """[N,D,H,W,C] view whose voxel rows are C contiguous floats at a constant stride > C inside one NDHWC buffer"""
    if t.dim() != 5 or t.is_contiguous():
        return False
    N, D, H, W_, C = t.shape
    ld = t.stride(3)
    return (t.stride(4) == 1 and ld > C and ld % 4 == 0 and t.stride(2) == W_ * ld and t.stride(1) == H * t.stride(2) and
            t.stride(0) == D * t.stride(1) and t.storage_offset() % 4 == 0)


def from_ndhwc(t):
    """contiguous [N,D,H,W,C] -> logical NCDHW view (no copy)"""
    return t.permute(0, 4, 1, 2, 3)


def to_ncdhw_contiguous(t_ndhwc):
    """[N,D,H,W,C] contiguous -> [N,C,D,H,W] contiguous via the HIP layout kernel"""
    N, D, H, W, C = t_ndhwc.shape
    out = torch.empty((N, C, D, H, W), dtype=torch.float32, device=t_ndhwc.device)
    E.call('seg3d_ndhwc_to_ncdhw', E.ptr(t_ndhwc), E.ptr(out), N, C, D * H * W, E.stream_ptr())
    return out


def _empty(shape, like, dtype=torch.float32):
    return torch.empty(shape, dtype=dtype, device=like.device)   # fp32 unless asked otherwise (also for bf16 `like`)


# ------------------------------------------------------------------------------------------------------------------
# convolution primitives on NDHWC buffers (no autograd)
# ------------------------------------------------------------------------------------------------------------------
def _pack_tapmajor(w, A, B, T, sa, sb, flip=0):
    BP = (B + 3) // 4 * 4
    wp = _empty((T, A, BP), w)
    E.call('seg3d_pack_weights_tapmajor', E.ptr(w), E.ptr(wp), A, B, BP, T, sa, sb, flip, E.stream_ptr())
    return wp


class _PackJob(ctypes.Structure):   # mirrors Seg3dPackJob (include/seg3d_hip.h)
    _fields_ = [('w', ctypes.c_void_p), ('wp', ctypes.c_void_p), ('sa', ctypes.c_longlong), ('sb', ctypes.c_longlong),
                ('first_block', ctypes.c_longlong), ('A', ctypes.c_int), ('B', ctypes.c_int), ('T', ctypes.c_int),
                ('flip', ctypes.c_int)]


class PackedWeightCache(object):
    """Packed (MFMA-layout) images of conv weights kept across calls.

    Every C->C conv packs its weight twice per train step (forward and data-gradient orientation): 52 tiny launches
    for the V-Net.  With the cache enabled (`weight_cache(True)`; core/seg_train.TrainStep, bench.py and the
    sliding-window inference path do it) an image is packed on first use and then re-used while it is current:
      * `repack_all()` -- called by FusedAdam.step() after its kernel -- refreshes ALL registered images with ONE
        launch (seg3d_pack_weights_mfma_multi) and marks them current;
      * `invalidate()` marks everything stale (parameters rewritten behind autograd's back: checkpoint load,
        broadcast, weight init); a stale image is re-packed in place on its next use, so pointers captured in a
        hipGraph stay valid;
      * an in-place torch update of a parameter (torch.optim.*) bumps `tensor._version`, which is part of the check.
    The cache is OFF by default: code that edits `param.data` in place (no version bump) must call `invalidate()`.
    """

    def __init__(self):
        self.enabled = False
        self.entries = {}     # key -> dict(w, wp, args, epoch, version)
        self.epoch = 0
        self._table = None    # (device uint8 tensor, njobs, total_blocks), rebuilt when entries change

    def invalidate(self):
        self.epoch += 1

    def clear(self):
        self.entries, self._table = {}, None
        self.epoch += 1

    def get(self, w, A, B, T, sa, sb, flip, bf16=False):
        key = (w.data_ptr(), w.device.index, A, B, T, sa, sb, flip, bf16)
        e = self.entries.get(key)
        if e is not None and e['epoch'] == self.epoch and e['version'] == w._version:
            return e['wp']
        if e is None:
            if bf16:
                wp = _empty((E.query('seg3d_packed_mfma_bf16_elems', A, B, T),), w, torch.bfloat16)
            else:
                wp = _empty((E.query('seg3d_packed_mfma_floats', A, B, T),), w)
            e = {'w': w, 'wp': wp, 'args': (A, B, T, sa, sb, flip), 'bf16': bf16}
            self.entries[key] = e
            self._table = None
        e['w'] = w
        E.call('seg3d_pack_weights_mfma_bf16' if bf16 else 'seg3d_pack_weights_mfma', E.ptr(w), E.ptr(e['wp']), A, B, T,
               sa, sb, flip, E.stream_ptr())
        e['epoch'], e['version'] = self.epoch, w._version
        return e['wp']

    def repack_all(self):
        """weights were just rewritten in place (optimizer step): refresh every registered image with one launch"""
        self.epoch += 1
        if not self.entries:
            return
        if self._table is None:
            tables = []
            for want_bf16 in (False, True):           # one job table (one launch) per packed format
                group = [e for e in self.entries.values() if e['bf16'] == want_bf16]
                if not group:
                    tables.append(None)
                    continue
                jobs, first = (_PackJob * len(group))(), 0
                for k, e in enumerate(group):
                    A, B, T, sa, sb, flip = e['args']
                    jobs[k] = _PackJob(e['w'].data_ptr(), e['wp'].data_ptr(), sa, sb, first, A, B, T, flip)
                    first += E.query('seg3d_pack_job_blocks_bf16' if want_bf16 else 'seg3d_pack_job_blocks', A, B, T)
                host = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8)
                tables.append((host.to(group[0]['wp'].device), len(group), first))
            self._table = tables
        for table, name in zip(self._table, ('seg3d_pack_weights_mfma_multi', 'seg3d_pack_weights_mfma_bf16_multi')):
            if table is not None:
                E.call(name, E.ptr(table[0]), table[1], table[2], E.stream_ptr())
        for e in self.entries.values():
            e['epoch'], e['version'] = self.epoch, e['w']._version


PACK_CACHE = PackedWeightCache()


def weight_cache(enabled):
    """turn the packed-weight cache on/off (see PackedWeightCache); returns the previous setting"""
    prev = PACK_CACHE.enabled
    PACK_CACHE.enabled = bool(enabled)
    if not enabled:
        PACK_CACHE.clear()
    return prev


def _pack_mfma(w, A, B, T, sa, sb, flip=0, bf16=False):
    if PACK_CACHE.enabled and not torch.cuda.is_current_stream_capturing():
        return PACK_CACHE.get(w, A, B, T, sa, sb, flip, bf16)
    if PACK_CACHE.enabled:   # inside a hipGraph capture: only a current image may be used (no allocation, no launch)
        key = (w.data_ptr(), w.device.index, A, B, T, sa, sb, flip, bf16)
        e = PACK_CACHE.entries.get(key)
        if e is not None and e['epoch'] == PACK_CACHE.epoch and e['version'] == w._version:
            return e['wp']
    if bf16:
        wp = _empty((E.query('seg3d_packed_mfma_bf16_elems', A, B, T),), w, torch.bfloat16)
        E.call('seg3d_pack_weights_mfma_bf16', E.ptr(w), E.ptr(wp), A, B, T, sa, sb, flip, E.stream_ptr())
        return wp
    n = E.query('seg3d_packed_mfma_floats', A, B, T)
    wp = _empty((n,), w)
    E.call('seg3d_pack_weights_mfma', E.ptr(w), E.ptr(wp), A, B, T, sa, sb, flip, E.stream_ptr())
    return wp


def _to_f32(t):
    """bf16 tensor -> fp32 copy (HIP converter kernel); fp32 passes through"""
    if t.dtype != torch.bfloat16:
        return t
    tc = t if t.is_contiguous() else t.contiguous()
    out = torch.empty(tc.shape, dtype=torch.float32, device=tc.device)
    E.call('seg3d_bf16_to_f32', E.ptr(tc), E.ptr(out), tc.numel(), E.stream_ptr())
    return out


def _use_mfma(cin, cout):
    """3x3x3 layers go to the matrix cores when both channel counts are MFMA-shaped (every C->C layer of vnet/vbnet)"""
    return (not FORCE_DIRECT) and cin % 4 == 0 and cin >= 8 and cout >= 8


def _conv_k3_generic(xn, w, bias, A, B, sa, sb, flip, want_stats, addend=None, out_bf16=False):
    """y[v][b] = bias[b] + sum_{t,a} x[v + t - 1][a] W(a,b,t) [+ addend];  returns (y, stats_partial or None)"""
    N, D, H, W_, Cin = xn.shape
    assert Cin == A
    thin_in16 = out_bf16 and not _is_bf16(xn) and A <= 8 and not FORCE_DIRECT and addend is None and not _use_mfma(A, B)
    if out_bf16 and not thin_in16 and not (_is_bf16(xn) and A % 16 == 0 and B % 4 == 0 and B >= 8 and not FORCE_DIRECT):
        out_bf16 = False      # only the bf16 MFMA kernel and the thin-input kernel write bf16 (the flag is a request)
    y = _empty((N, D, H, W_, B), xn, torch.bfloat16 if out_bf16 else torch.float32)
    if _is_bf16(xn):
        if A % 16 == 0 and B % 4 == 0 and B >= 8 and not FORCE_DIRECT:
            wp = _pack_mfma(w, A, B, 27, sa, sb, flip, bf16=True)
            stats = None
            if want_stats:
                stats = _empty((N, E.query('seg3d_conv3d_k3_bf16_stats_count', N, D, H, W_, A, B), 2), xn)
            nws = E.query('seg3d_conv3d_k3_bf16_fwd_workspace_floats', N, D, H, W_, A, B)
            ws = _empty((nws,), xn) if nws else None
            E.call('seg3d_conv3d_k3_bf16_fwd', E.ptr(xn), E.ptr(wp), E.ptr(bias), E.ptr(addend), E.ptr(y), E.ptr(stats),
                   E.ptr(ws), N, D, H, W_, A, B, int(out_bf16), E.stream_ptr())
            return y, stats
        if B <= 8 and A % 4 == 0 and addend is None and not FORCE_DIRECT:   # head forward on bf16 activations
            stats = None
            if want_stats:
                stats = _empty((N, E.query('seg3d_conv3d_k3_thin_out_stats_count', D, H, W_), 2), xn)
            if E.query('seg3d_conv3d_k3_thin_out_mfma_supported', A, B):
                wq = torch.empty(E.query('seg3d_thin_out_mfma_packed_elems', A), dtype=torch.bfloat16, device=w.device)
                E.call('seg3d_pack_weights_thin_out_mfma', E.ptr(w), E.ptr(wq), A, B, sa, sb, flip, E.stream_ptr())
                E.call('seg3d_conv3d_k3_thin_out_mfma_fwd', E.ptr(xn), E.ptr(wq), E.ptr(bias), E.ptr(y), E.ptr(stats), N, D,
                       H, W_, A, B, E.stream_ptr())
                return y, stats
            CO = 2 if B <= 2 else (4 if B <= 4 else 8)
            wq = _empty(((A + 7) // 8 * 27 * 8 * CO,), w)
            E.call('seg3d_pack_weights_thin_out', E.ptr(w), E.ptr(wq), A, B, CO, sa, sb, flip, E.stream_ptr())
            E.call('seg3d_conv3d_k3_thin_out_bf16_fwd', E.ptr(xn), E.ptr(wq), E.ptr(bias), E.ptr(y), E.ptr(stats), N, D, H,
                   W_, A, B, CO, E.stream_ptr())
            return y, stats
        xn = _to_f32(xn)       # shapes outside the bf16 kernels (not produced by vnet / vbnet): widen and continue
    if _use_mfma(A, B) and WINOGRAD and WINOGRAD2D and E.query('seg3d_conv3d_k3_wino2d_preferred', N, D, H, W_, A, B):
        # the big levels: Winograd F(2x2, 3x3) over (y, x), 4/9 of the fp32 MFMAs (csrc/conv_wino2d.hip); T = 48 image
        wp = _pack_mfma(w, A, B, 48, sa, sb, flip)
        stats = None
        if want_stats:
            stats = _empty((N, E.query('seg3d_conv3d_k3_wino2d_stats_count', N, D, H, W_, A, B), 2), xn)
        # stream-K partial slabs where the (tile, column block) items would fill the CUs' rounds badly (the 24^3 level); a fresh block
        # of the caching allocator per call: two forwards on two streams never share one
        nws = E.query('seg3d_conv3d_k3_wino2d_fwd_workspace_floats', N, D, H, W_, A, B) if (STREAM_K and not _TWO_FORWARDS) else 0
        ws = _empty((nws,), xn) if nws else None
        E.call('seg3d_conv3d_k3_wino2d_fwd_ws', E.ptr(xn), E.ptr(wp), E.ptr(bias), E.ptr(addend), E.ptr(y), E.ptr(stats), E.ptr(ws),
               N, D, H, W_, A, B, E.stream_ptr())
        return y, stats
    if _use_mfma(A, B) and WINOGRAD and E.query('seg3d_conv3d_k3_wino_preferred', N, D, H, W_, A, B):
        # the big levels: Winograd F(2, 3) along x, 2/3 of the fp32 MFMAs (csrc/conv_wino.hip); T = 36 = transformed image
        wp = _pack_mfma(w, A, B, 36, sa, sb, flip)
        stats = None
        if want_stats:
            stats = _empty((N, E.query('seg3d_conv3d_k3_wino_stats_count', N, D, H, W_, A, B), 2), xn)
        E.call('seg3d_conv3d_k3_wino_fwd', E.ptr(xn), E.ptr(wp), E.ptr(bias), E.ptr(addend), E.ptr(y), E.ptr(stats), N, D, H,
               W_, A, B, E.stream_ptr())
        return y, stats
    if _use_mfma(A, B):
        wp = _pack_mfma(w, A, B, 27, sa, sb, flip)
        stats = None
        if want_stats:
            cnt = E.query('seg3d_conv3d_k3_mfma_stats_count', N, D, H, W_, A, B)
            stats = _empty((N, cnt, 2), xn)
        nws = E.query('seg3d_conv3d_k3_mfma_fwd_workspace_floats', N, D, H, W_, A, B)
        ws = _empty((nws,), xn) if nws else None    # split-K partial slabs for the deep, spatially tiny levels
        E.call('seg3d_conv3d_k3_mfma_fwd', E.ptr(xn), E.ptr(wp), E.ptr(bias), E.ptr(addend), E.ptr(y), E.ptr(stats),
               E.ptr(ws), N, D, H, W_, A, B, E.stream_ptr())
        return y, stats
    if addend is not None:   # the special-case kernels below have no fused addend: add afterwards (device op)
        y2, st = _conv_k3_generic(xn, w, bias, A, B, sa, sb, flip, want_stats)
        return y2.add_(addend), st
    if 3 <= A <= 8 and not FORCE_DIRECT and WINOGRAD and WINOGRAD2D and not _is_bf16(xn) and not _is_bf16(y) and \
            E.query('seg3d_conv3d_k3_wino2d_preferred', N, D, H, W_, 8, B):
        # 3..8 thin channels onto a big level (the data-gradient of a 3..8-class head): zero-pad the thin side to 8 channels
        # and run the Winograd kernel -- the thin-input kernel folds all 27*A taps into the MFMA K dimension, which pays
        # for A <= 2 only (5 classes at 4 x 96^3: 0.75 ms there, 0.3 ms this way); the T = 48 pack zero-fills rows a >= A
        x8 = torch.zeros((N, D, H, W_, 8), dtype=torch.float32, device=xn.device)
        E.call('seg3d_copy_channels', E.ptr(xn), E.ptr(x8), N * D * H * W_, A, A, 0, 8, 0, E.stream_ptr())
        wp = _pack_mfma(w, A, B, 48, sa, sb, flip)
        stats = None
        if want_stats:
            stats = _empty((N, E.query('seg3d_conv3d_k3_wino2d_stats_count', N, D, H, W_, 8, B), 2), xn)
        E.call('seg3d_conv3d_k3_wino2d_fwd', E.ptr(x8), E.ptr(wp), E.ptr(bias), E.ptr(addend), E.ptr(y), E.ptr(stats), N, D, H,
               W_, 8, B, E.stream_ptr())
        return y, stats
    if A <= 8 and not FORCE_DIRECT:
        # thin input (stem forward, head data-gradient): all 27*A taps folded into one MFMA K dimension
        if _is_bf16(y) and E.query('seg3d_conv3d_k3_thin_in_mfma16_supported', A, B):
            stats = None
            if want_stats:
                stats = _empty((N, E.query('seg3d_conv3d_k3_thin_stats_count', D, H, W_, (B + 31) // 32), 2), xn)
            wq = torch.empty(E.query('seg3d_packed_thin_in16_elems', A, B), dtype=torch.bfloat16, device=w.device)
            E.call('seg3d_pack_weights_thin_in16', E.ptr(w), E.ptr(wq), A, B, sa, sb, flip, E.stream_ptr())
            E.call('seg3d_conv3d_k3_thin_in_mfma16_fwd', E.ptr(xn), E.ptr(wq), E.ptr(bias), E.ptr(y), E.ptr(stats), N, D, H,
                   W_, A, B, E.stream_ptr())
            return y, stats
        # fp32 arithmetic: the persistent kernel (tile-invariant index work hoisted, one statistics slot per wave)
        stats = None
        if want_stats:
            stats = _empty((N, E.query('seg3d_conv3d_k3_thin_in_persistent_stats_count', D, H, W_, (B + 31) // 32), 2), xn)
        wp = _empty((E.query('seg3d_packed_thin_in_floats', A, B),), w)
        E.call('seg3d_pack_weights_thin_in', E.ptr(w), E.ptr(wp), A, B, sa, sb, flip, E.stream_ptr())
        E.call('seg3d_conv3d_k3_thin_in_persistent_fwd', E.ptr(xn), E.ptr(wp), E.ptr(bias), E.ptr(y), E.ptr(stats), N, D, H,
               W_, A, B, int(_is_bf16(y)), E.stream_ptr())
        return y, stats
    if not FORCE_DIRECT and E.query('seg3d_conv3d_k3_thin_out_f32mfma_supported', A, B):
        # thin output (head forward, Cin 16 / 32 -> <= 5 classes): fp32 matrix cores, (kz, ky) taps in the MFMA rows
        wq = _empty((E.query('seg3d_thin_out_f32mfma_packed_floats', A, B),), w)
        E.call('seg3d_pack_weights_thin_out_f32mfma', E.ptr(w), E.ptr(wq), A, B, sa, sb, flip, E.stream_ptr())
        stats = None
        if want_stats:
            stats = _empty((N, E.query('seg3d_conv3d_k3_thin_out_f32mfma_stats_count', N, D, H, W_), 2), xn)
        E.call('seg3d_conv3d_k3_thin_out_f32mfma_fwd', E.ptr(xn), E.ptr(wq), E.ptr(bias), E.ptr(y), E.ptr(stats), N, D, H, W_,
               A, B, E.stream_ptr())
        return y, stats
    if B <= 8 and A % 4 == 0 and not FORCE_DIRECT:
        # thin output, other shapes: LDS-tiled VALU kernel, CO outputs per voxel
        CO = 2 if B <= 2 else (4 if B <= 4 else 8)
        wq = _empty(((A + 7) // 8 * 27 * 8 * CO,), w)
        E.call('seg3d_pack_weights_thin_out', E.ptr(w), E.ptr(wq), A, B, CO, sa, sb, flip, E.stream_ptr())
        stats = None
        if want_stats:
            stats = _empty((N, E.query('seg3d_conv3d_k3_thin_out_stats_count', D, H, W_), 2), xn)
        E.call('seg3d_conv3d_k3_thin_out_fwd', E.ptr(xn), E.ptr(wq), E.ptr(bias), E.ptr(y), E.ptr(stats), N, D, H, W_, A, B,
               CO, E.stream_ptr())
        return y, stats
    wp = _pack_tapmajor(w, A, B, 27, sa, sb, flip)
    E.call('seg3d_conv3d_fwd_direct', E.ptr(xn), E.ptr(wp), E.ptr(bias), E.ptr(y), N, D, H, W_, A, B, 3, 1, E.stream_ptr())
    return y, None


def _k2_gather(xn, w, bias, y, A, B, sa, sb, want_stats, ldx=0):
    """y[v][b] = bias[b] + sum_{t,a} x[2v + t][a] W(a,b,t) on the matrix cores; y preallocated [N,Do,Ho,Wo,B]"""
    N, Do, Ho, Wo, _ = y.shape
    w16 = _is_bf16(xn) and A % 16 == 0    # bf16 weight image -> the kernel's bf16-MFMA mode
    wp = _pack_mfma(w, A, B, 8, sa, sb, bf16=w16)
    stats = None
    if want_stats:
        stats = _empty((N, E.query('seg3d_conv3d_k2s2_mfma_stats_count', Do, Ho, Wo, B), 2), xn)
    if _is_bf16(xn):
        E.call('seg3d_conv3d_k2s2_bf16_fwd', E.ptr(xn), E.ptr(wp), E.ptr(bias), E.ptr(y), E.ptr(stats), N, Do, Ho, Wo, A, B,
               int(_is_bf16(y)), int(w16), E.stream_ptr())
        return y, stats
    if ldx:
        E.call('seg3d_conv3d_k2s2_mfma_fwd_ld', E.ptr(xn), int(ldx), E.ptr(wp), E.ptr(bias), E.ptr(y), E.ptr(stats), N, Do, Ho,
               Wo, A, B, E.stream_ptr())
        return y, stats
    E.call('seg3d_conv3d_k2s2_mfma_fwd', E.ptr(xn), E.ptr(wp), E.ptr(bias), E.ptr(y), E.ptr(stats), N, Do, Ho, Wo, A, B,
           E.stream_ptr())
    return y, stats


def _k2_scatter(xn, w, bias, y, A, B, sa, sb, want_stats):
    """y[2i + t][b] = bias[b] + sum_a x[i][a] W(a,b,t) on the matrix cores; y preallocated [N,2D,2H,2W,B]"""
    N, D, H, W_, _ = xn.shape
    w16 = _is_bf16(xn) and A % 16 == 0
    wp = _pack_mfma(w, A, B, 8, sa, sb, bf16=w16)
    stats = None
    if want_stats:
        stats = _empty((N, E.query('seg3d_convT3d_k2s2_mfma_stats_count', D, H, W_, B), 2), xn)
    if _is_bf16(xn):
        E.call('seg3d_convT3d_k2s2_bf16_fwd', E.ptr(xn), E.ptr(wp), E.ptr(bias), E.ptr(y), E.ptr(stats), N, D, H, W_, A, B,
               int(_is_bf16(y)), int(w16), E.stream_ptr())
        return y, stats
    E.call('seg3d_convT3d_k2s2_mfma_fwd', E.ptr(xn), E.ptr(wp), E.ptr(bias), E.ptr(y), E.ptr(stats), N, D, H, W_, A, B,
           E.stream_ptr())
    return y, stats


def _k2_wgrad(P, Q, CA, CB, out_shape, sa, sb, out=None):
    """dW(t,a,b) = sum_v P[2v + t][a] Q[v][b] on the matrix cores, written to dw[a*sa + b*sb + t] (added when `out`)"""
    N, Dq, Hq, Wq, _ = Q.shape
    if _is_bf16(P) != _is_bf16(Q):
        P, Q = _to_f32(P), _to_f32(Q)
    ws = _empty((E.query('seg3d_k2_mfma_wgrad_workspace_floats', N, Dq, Hq, Wq, CA, CB),), P)
    dw = _empty(out_shape, P) if out is None else out
    E.call('seg3d_k2_bf16_wgrad' if _is_bf16(P) else 'seg3d_k2_mfma_wgrad', E.ptr(P), E.ptr(Q), E.ptr(dw), E.ptr(ws), N,
           Dq, Hq, Wq, CA, CB, sa, sb, int(out is not None), E.stream_ptr())
    return dw


def _thin_wgrad(thin, fat, CT, CF, out_shape, s_ct, s_cf, flip, out=None):
    """dw[ct*s_ct + cf*s_cf + tap] = sum_u fat[u][cf] thin[u + off(tap)][ct] (taps reversed when flip) on the matrix cores"""
    N, D, H, W_, _ = thin.shape
    ws = _empty((E.query('seg3d_k3_thin_wgrad_workspace_floats', N, D, H, W_, CT, CF),), thin)
    dw = _empty(out_shape, thin) if out is None else out
    thin = _to_f32(thin)
    E.call('seg3d_k3_thin_wgrad_fatbf16' if _is_bf16(fat) else 'seg3d_k3_thin_wgrad', E.ptr(thin), E.ptr(fat), E.ptr(dw),
           E.ptr(ws), N, D, H, W_, CT, CF, s_ct, s_cf, flip, int(out is not None), E.stream_ptr())
    return dw


def conv_forward(xn, w, bias, kind, want_stats=False, out_bf16=False):
    """xn: [N,D,H,W,Cin] contiguous; w in the reference layout; returns (y NDHWC, stats_partial or None).
    out_bf16 (bf16 mode): REQUEST a bf16 y; honoured by the kernels that take bf16 input (check y.dtype)"""
    ks, stride, T, transposed = _KINDS[kind]
    N, D, H, W_, Cin = xn.shape
    ldx = 0
    if not xn.is_contiguous():
        if kind == 'k2s2' and xn.dtype == torch.float32 and _is_channel_slice(xn) and _use_mfma(Cin, w.shape[0]) and \
                w.shape[0] % 4 == 0:
            ldx = xn.stride(3)            # read in place by seg3d_conv3d_k2s2_mfma_fwd_ld
        else:
            xn = xn.contiguous()
    if kind == 'k3':
        Cout = w.shape[0]
        _check_w(w, (Cout, Cin, 3, 3, 3), kind)
        return _conv_k3_generic(xn, w, bias, Cin, Cout, 27, Cin * 27, 0, want_stats, out_bf16=out_bf16)
    if kind == 'k2s2':
        Cout = w.shape[0]
        _check_w(w, (Cout, Cin, 2, 2, 2), kind)
        if D % 2 or H % 2 or W_ % 2:
            raise ValueError('Conv3d k2 s2 needs even spatial dims, got {}'.format((D, H, W_)))
        if _use_mfma(Cin, Cout) and Cout % 4 == 0:
            ydt = torch.bfloat16 if (out_bf16 and _is_bf16(xn)) else torch.float32
            y = _empty((N, D // 2, H // 2, W_ // 2, Cout), xn, ydt)
            return _k2_gather(xn, w, bias, y, Cin, Cout, 8, Cin * 8, want_stats, ldx=ldx)
        y = _empty((N, D // 2, H // 2, W_ // 2, Cout), xn)
        xn = _to_f32(xn)
        wp = _pack_tapmajor(w, Cin, Cout, 8, 8, Cin * 8)
        E.call('seg3d_conv3d_fwd_direct', E.ptr(xn), E.ptr(wp), E.ptr(bias), E.ptr(y), N, D, H, W_, Cin, Cout, 2, 2,
               E.stream_ptr())
        return y, None
    if kind == 'k1':
        Cout = w.shape[0]
        _check_w(w, (Cout, Cin, 1, 1, 1), kind)
        xn = _to_f32(xn)
        wp = _pack_tapmajor(w, Cin, Cout, 1, 1, Cin)
        y = _empty((N, D, H, W_, Cout), xn)
        E.call('seg3d_conv3d_fwd_direct', E.ptr(xn), E.ptr(wp), E.ptr(bias), E.ptr(y), N, D, H, W_, Cin, Cout, 1, 1,
               E.stream_ptr())
        return y, None
    if kind == 'convT':
        Cout = w.shape[1]
        _check_w(w, (Cin, Cout, 2, 2, 2), kind)
        if _use_mfma(Cin, Cout) and Cout % 4 == 0:
            ydt = torch.bfloat16 if (out_bf16 and _is_bf16(xn)) else torch.float32
            y = _empty((N, 2 * D, 2 * H, 2 * W_, Cout), xn, ydt)
            return _k2_scatter(xn, w, bias, y, Cin, Cout, Cout * 8, 8, want_stats)
        y = _empty((N, 2 * D, 2 * H, 2 * W_, Cout), xn)
        xn = _to_f32(xn)
        wp = _pack_tapmajor(w, Cin, Cout, 8, Cout * 8, 8)
        E.call('seg3d_convT3d_k2s2_fwd_direct', E.ptr(xn), E.ptr(wp), E.ptr(bias), E.ptr(y), N, D, H, W_, Cin, Cout,
               E.stream_ptr())
        return y, None
    raise ValueError('unknown conv kind {}'.format(kind))


def _check_w(w, shape, kind):
    if tuple(w.shape) != tuple(shape):
        raise ValueError('weight shape {} does not match {} for conv kind {}'.format(tuple(w.shape), shape, kind))
    if not w.is_contiguous():
        raise ValueError('conv weights must be contiguous')


# stride-2 conv data-gradient: the skip connection's gradient is added in the kernel epilogue


def _addend_row_stride(t, C):
    """row stride (elements) of an NDHWC tensor or channel slice whose voxels are evenly spaced rows; 0 if it is not one"""
    if t.dim() != 5 or t.shape[4] != C or t.stride(4) != 1:
        return 0
    ld = t.stride(3)
    if ld < C or ld % 4 or t.stride(2) != ld * t.shape[3] or t.stride(1) != ld * t.shape[3] * t.shape[2] or \
            t.stride(0) != ld * t.shape[3] * t.shape[2] * t.shape[1]:
        return 0
    return ld


def conv_dgrad(dyn, w, kind, addend=None, want_bf16=None):
    """gradient w.r.t. the conv input (+ addend, an extra gradient for the same tensor that is folded into the kernel
    epilogue); dyn: [N,Do,Ho,Wo,Cout] contiguous.  want_bf16: the input is a bf16 activation, so its gradient should be
    written as bf16 by the kernel where it can (default: follows dyn's dtype)"""
    if want_bf16 is None:
        want_bf16 = _is_bf16(dyn)
    N, D, H, W_, _ = dyn.shape
    if kind == 'k3':
        Cout, Cin = w.shape[0], w.shape[1]
        # dx[v][ci] = sum_{t',co} dy[v + t' - 1][co] w[co][ci][26 - t']
        # (a bf16 dy means the unit's input is a bf16 activation: its gradient is written as bf16 by the kernel itself)
        dx, _ = _conv_k3_generic(dyn, w, None, Cout, Cin, Cin * 27, 27, 1, False, addend=addend, out_bf16=want_bf16)
        return dx
    if addend is not None:
        if kind == 'k2s2':
            Cout, Cin = w.shape[0], w.shape[1]
            dy16 = _is_bf16(dyn) and _use_mfma(Cout, Cin) and Cin % 4 == 0
            ld = _addend_row_stride(addend, Cin)
            if _use_mfma(Cout, Cin) and Cin % 8 == 0 and ld and _is_bf16(addend) == dy16 and \
                    tuple(addend.shape) == (N, 2 * D, 2 * H, 2 * W_, Cin):
                # the skip connection's gradient (a channel slice of the concatenated gradient) joins in the epilogue
                if _is_bf16(dyn) and not dy16:
                    dyn = _to_f32(dyn)
                w16 = dy16 and Cout % 16 == 0
                wp = _pack_mfma(w, Cout, Cin, 8, Cin * 8, 8, bf16=w16)
                dx = _empty((N, 2 * D, 2 * H, 2 * W_, Cin), dyn, dyn.dtype)
                E.call('seg3d_convT3d_k2s2_scatter_addend', E.ptr(dyn), (2 if w16 else 1) if dy16 else 0, E.ptr(wp),
                       E.ptr(addend), ld, E.ptr(dx), N, D, H, W_, Cout, Cin, int(dy16), E.stream_ptr())
                return dx
        return conv_dgrad(dyn, w, kind).add_(addend)
    if _is_bf16(dyn):
        cout, cin = (w.shape[0], w.shape[1]) if kind == 'k2s2' else (w.shape[1], w.shape[0])
        if not (kind in ('k2s2', 'convT') and _use_mfma(cout, cin) and cin % 4 == 0):
            dyn = _to_f32(dyn)         # only the stride-2 MFMA kernels take bf16 gradients here
    if kind == 'k2s2':
        Cout, Cin = w.shape[0], w.shape[1]
        # dx[2v + t][ci] = sum_co dy[v][co] w[co][ci][t]  == transposed conv of dy
        if _use_mfma(Cout, Cin) and Cin % 4 == 0:
            dx = _empty((N, 2 * D, 2 * H, 2 * W_, Cin), dyn, dyn.dtype)     # bf16 dy -> bf16 dx, written by the kernel
            return _k2_scatter(dyn, w, None, dx, Cout, Cin, Cin * 8, 8, False)[0]
        dx = _empty((N, 2 * D, 2 * H, 2 * W_, Cin), dyn)
        wp = _pack_tapmajor(w, Cout, Cin, 8, Cin * 8, 8)
        E.call('seg3d_convT3d_k2s2_fwd_direct', E.ptr(dyn), E.ptr(wp), None, E.ptr(dx), N, D, H, W_, Cout, Cin,
               E.stream_ptr())
        return dx
    if kind == 'k1':
        Cout, Cin = w.shape[0], w.shape[1]
        wp = _pack_tapmajor(w, Cout, Cin, 1, Cin, 1)
        dx = _empty((N, D, H, W_, Cin), dyn)
        E.call('seg3d_conv3d_fwd_direct', E.ptr(dyn), E.ptr(wp), None, E.ptr(dx), N, D, H, W_, Cout, Cin, 1, 1,
               E.stream_ptr())
        return dx
    if kind == 'convT':
        Cin, Cout = w.shape[0], w.shape[1]
        # dx[i][ci] = sum_{t,co} dy[2i + t][co] w[ci][co][t]  == k2 s2 conv of dy
        if _use_mfma(Cout, Cin) and Cin % 4 == 0:
            dx = _empty((N, D // 2, H // 2, W_ // 2, Cin), dyn, dyn.dtype)
            return _k2_gather(dyn, w, None, dx, Cout, Cin, 8, Cout * 8, False)[0]
        dx = _empty((N, D // 2, H // 2, W_ // 2, Cin), dyn)
        wp = _pack_tapmajor(w, Cout, Cin, 8, 8, Cout * 8)
        E.call('seg3d_conv3d_fwd_direct', E.ptr(dyn), E.ptr(wp), None, E.ptr(dx), N, D, H, W_, Cout, Cin, 2, 2,
               E.stream_ptr())
        return dx
    raise ValueError('unknown conv kind {}'.format(kind))


def _wgrad_direct(P, Q, CA, CB, ks, stride, T, out_shape, sa, sb, out=None):
    N, Dp, Hp, Wp, _ = P.shape
    _, Dq, Hq, Wq, _ = Q.shape
    nfl = E.query('seg3d_wgrad_direct_workspace_floats', N, Dq, Hq, Wq, CA, CB, T)
    part = _empty((nfl,), P)
    chunks = ctypes.c_int(0)
    E.call('seg3d_wgrad_direct', E.ptr(P), E.ptr(Q), E.ptr(part), N, Dp, Hp, Wp, CA, CB, ks, stride,
           ctypes.byref(chunks), E.stream_ptr())
    dw = _empty(out_shape, P) if out is None else out
    E.call('seg3d_wgrad_reduce', E.ptr(part), E.ptr(dw), chunks.value, T, CA, CB, sa, sb, int(out is not None),
           E.stream_ptr())
    return dw


def conv_wgrad(xn, dyn, w_shape, kind, out=None):
    """gradient w.r.t. the weight in the reference layout `w_shape`: returned as a new tensor, or ADDED into `out`
    (a contiguous tensor of that shape, e.g. the optimizer's flat-buffer slice) by the reduce kernel itself"""
    N, D, H, W_, Cin_x = xn.shape
    if out is not None and (tuple(out.shape) != tuple(w_shape) or not out.is_contiguous()):
        raise ValueError('weight-gradient destination must be contiguous with shape {}'.format(tuple(w_shape)))
    if kind == 'k3':
        Cout, Cin = w_shape[0], w_shape[1]
        if _is_bf16(xn) and _is_bf16(dyn) and _use_mfma(Cin, Cout) and Cout % 4 == 0:
            ws = _empty((E.query('seg3d_conv3d_k3_bf16_wgrad_workspace_floats', N, D, H, W_, Cin, Cout),), xn)
            dw = _empty(w_shape, xn) if out is None else out
            E.call('seg3d_conv3d_k3_bf16_wgrad', E.ptr(xn), E.ptr(dyn), E.ptr(dw), E.ptr(ws), N, D, H, W_, Cin, Cout,
                   int(out is not None), E.stream_ptr())
            return dw
        if _is_bf16(xn) and not _is_bf16(dyn) and not FORCE_DIRECT and Cout <= 8 and Cin % 4 == 0 and \
                not (_use_mfma(Cin, Cout) and Cout % 4 == 0):
            # bf16-mode head: thin = fp32 dy, fat = the unit's bf16 input, widened inside the kernel
            return _thin_wgrad(dyn, xn, Cout, Cin, w_shape, Cin * 27, 27, 1, out)
        if not _is_bf16(xn) and _is_bf16(dyn) and not FORCE_DIRECT and Cin <= 8 and Cout % 4 == 0:
            # bf16-mode stem: thin = the fp32 image, fat = the bf16 gradient of the unit's conv output
            return _thin_wgrad(xn, dyn, Cin, Cout, w_shape, 27, Cin * 27, 0, out)
        xn, dyn = _to_f32(xn), _to_f32(dyn)     # remaining mixed cases: widen, then the fp32 kernels
        if _use_mfma(Cin, Cout) and Cout % 4 == 0 and WINOGRAD and \
                E.query('seg3d_conv3d_k3_wino_wgrad_supported', N, D, H, W_, Cin, Cout):
            # Winograd F(3x3, 2x2) over (y, x) (4/9 of the MFMAs of the 27-tap kernel) or F(3, 2) along x (2/3)
            form = 'wino2d' if WINOGRAD2D and E.query('seg3d_conv3d_k3_wino2d_wgrad_preferred', N, D, H, W_, Cin, Cout) else 'wino'
            nfl = E.query('seg3d_conv3d_k3_{}_wgrad_workspace_floats'.format(form), N, D, H, W_, Cin, Cout)
            ws = _empty((nfl,), xn)
            dw = _empty(w_shape, xn) if out is None else out
            E.call('seg3d_conv3d_k3_{}_wgrad'.format(form), E.ptr(xn), E.ptr(dyn), E.ptr(dw), E.ptr(ws), N, D, H, W_, Cin, Cout,
                   int(out is not None), E.stream_ptr())
            return dw
        if _use_mfma(Cin, Cout) and Cout % 4 == 0:
            nfl = E.query('seg3d_conv3d_k3_mfma_wgrad_workspace_floats', N, D, H, W_, Cin, Cout)
            ws = _empty((nfl,), xn)
            dw = _empty(w_shape, xn) if out is None else out
            E.call('seg3d_conv3d_k3_mfma_wgrad', E.ptr(xn), E.ptr(dyn), E.ptr(dw), E.ptr(ws), N, D, H, W_, Cin, Cout,
                   int(out is not None), E.stream_ptr())
            return dw
        if not FORCE_DIRECT and Cin <= 8 and Cout % 4 == 0:      # stem: thin = x, fat = dy
            return _thin_wgrad(xn, dyn, Cin, Cout, w_shape, 27, Cin * 27, 0, out)
        if not FORCE_DIRECT and Cout <= 8 and Cin % 4 == 0:      # head: thin = dy, fat = x, taps reversed
            return _thin_wgrad(dyn, xn, Cout, Cin, w_shape, Cin * 27, 27, 1, out)
        return _wgrad_direct(xn, dyn, Cin, Cout, 3, 1, 27, w_shape, 27, Cin * 27, out)
    if kind == 'k2s2':
        Cout, Cin = w_shape[0], w_shape[1]
        if _use_mfma(Cin, Cout) and Cout % 4 == 0:
            return _k2_wgrad(xn, dyn, Cin, Cout, w_shape, 8, Cin * 8, out)
        return _wgrad_direct(_to_f32(xn), _to_f32(dyn), Cin, Cout, 2, 2, 8, w_shape, 8, Cin * 8, out)
    if kind == 'k1':
        Cout, Cin = w_shape[0], w_shape[1]
        return _wgrad_direct(_to_f32(xn), _to_f32(dyn), Cin, Cout, 1, 1, 1, w_shape, 1, Cin, out)
    if kind == 'convT':
        Cin, Cout = w_shape[0], w_shape[1]
        # dW(t, a=co, b=ci) = sum_i dy[2i + t][co] x[i][ci]  -> w[ci][co][t]
        if _use_mfma(Cin, Cout) and Cout % 4 == 0:
            return _k2_wgrad(dyn, xn, Cout, Cin, w_shape, 8, Cout * 8, out)
        return _wgrad_direct(_to_f32(dyn), _to_f32(xn), Cout, Cin, 2, 2, 8, w_shape, 8, Cout * 8, out)
    raise ValueError('unknown conv kind {}'.format(kind))


# ------------------------------------------------------------------------------------------------------------------
# GroupNorm(1, C) primitives on NDHWC buffers (no autograd)
# ------------------------------------------------------------------------------------------------------------------
def gn_stats(yn, stats_partial=None, eps=GN_EPS):
    """(mean, rstd) per sample over all C*D*H*W elements -> [N,2]"""
    N = yn.shape[0]
    M = yn[0].numel()
    if stats_partial is None:
        yn = _to_f32(yn)
        cnt = E.query('seg3d_gn_stats_count', M)
        stats_partial = _empty((N, cnt, 2), yn)
        E.call('seg3d_gn_stats_partial', E.ptr(yn), E.ptr(stats_partial), N, M, E.stream_ptr())
    cnt = stats_partial.shape[1]
    mean_rstd = _empty((N, 2), yn)
    E.call('seg3d_gn_stats_finalize', E.ptr(stats_partial), E.ptr(mean_rstd), N, cnt, M, float(eps), E.stream_ptr())
    return mean_rstd


def gn_apply(yn, mean_rstd, gamma, beta, resn, relu, out=None, out_bf16=False):
    """out: optional destination -- a channel slice [..., c0:c0+C] of a wider contiguous NDHWC buffer (the up-branch
    half of a skip concatenation is normalised straight into the concatenated tensor).
    bf16 mode: `resn` may be bf16; the output is bf16 when out_bf16 (or when `out` is a bf16 buffer)."""
    N, D, H, W_, C = yn.shape
    ld = 0
    if _is_bf16(yn) and C % 4:
        yn = _to_f32(yn)
    if out is None:
        out = torch.empty(yn.shape, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=yn.device)
    else:
        if tuple(out.shape) != tuple(yn.shape) or out.stride(4) != 1 or out.stride(3) < C or out.stride(3) % 4 or \
                out.stride(2) != W_ * out.stride(3) or out.stride(1) != H * out.stride(2) or out.stride(0) != D * out.stride(1):
            raise ValueError('gn_apply destination must be a channel slice of a contiguous NDHWC buffer')
        ld = out.stride(3)
    if _is_bf16(out) or _is_bf16(resn) or _is_bf16(yn):
        E.call('seg3d_gn_apply_mixed', E.ptr(yn), E.ptr(mean_rstd), E.ptr(gamma), E.ptr(beta), E.ptr(resn), E.ptr(out), N,
               D * H * W_, C, int(relu), ld, int(_is_bf16(resn)), int(_is_bf16(out)), int(_is_bf16(yn)), E.stream_ptr())
        return out
    E.call('seg3d_gn_apply', E.ptr(yn), E.ptr(mean_rstd), E.ptr(gamma), E.ptr(beta), E.ptr(resn), E.ptr(out), N,
           D * H * W_, C, int(relu), ld, E.stream_ptr())
    return out


def _row_stride(t, C):
    """0 for a contiguous [N,D,H,W,C] tensor, else the voxel stride of a channel slice of a wider contiguous buffer"""
    if t.is_contiguous():
        return 0
    N, D, H, W_, _ = t.shape
    ld = t.stride(3)
    if t.stride(4) != 1 or ld < C or ld % 4 or t.stride(2) != W_ * ld or t.stride(1) != H * t.stride(2) or \
            t.stride(0) != D * t.stride(1) or (t.storage_offset() % 4):
        raise ValueError('expected a contiguous NDHWC tensor or a channel slice of one')
    return ld


# GroupNorm backward finalize: two tiny launches (per-sample sums, then the parameter gradients).  GN_FUSED_FINALIZE = True
# runs both stages in one launch with a last-ticket workgroup (round 1's default; the fused kernel takes a ticket counter
# that is zero between calls, one per device -- all GroupNorm backward calls are issued on one stream).  Measured on one box
# inside the captured step (tools/ab_step.py, 60 steps, three pairs): the two-launch form is 0.02-0.08 ms per step faster in
# fp32 and 0.06 ms in bf16 -- in a graph replay a launch boundary costs less than the fence + atomic ticket it replaces.
GN_FUSED_FINALIZE = False
_GN_TICKETS = {}


def _gn_ticket(device):
    t = _GN_TICKETS.get(device)
    if t is None:
        t = _GN_TICKETS[device] = torch.zeros(1, dtype=torch.int32, device=device)
    return t


def gn_backward(doutn, outn, yn, mean_rstd, gamma, beta, relu, want_dres, want_dbias=True, sinks=(None, None, None),
                dy_bf16=False):
    """returns (dy, dres or None, dgamma, dbeta, dbias or None).  `outn` (the unit's forward output) is only read when
    it cannot be recomputed from y, i.e. when a residual was added; pass None otherwise.
    sinks = (dgamma, dbeta, dbias) destinations ([C] tensors) the finalize kernel ADDS into; the matching return
    value is then None."""
    N, D, H, W_, C = yn.shape
    S = D * H * W_
    nblk = E.query('seg3d_gn_bwd_blocks', S)
    part = _empty((N, nblk, C, 3), yn)
    mask_src = outn if relu else None
    ldd = _row_stride(doutn, C)   # dout may be a channel slice of the concatenated gradient (UpCatFunction)
    act_bf16 = _is_bf16(doutn)    # bf16 mode: the incoming gradient has the dtype of the unit's (bf16) output
    if act_bf16 != (_is_bf16(mask_src) if mask_src is not None else act_bf16):
        raise TypeError('GroupNorm backward: gradient and saved output differ in dtype')
    if dy_bf16 and not act_bf16:
        raise TypeError('GroupNorm backward: a bf16 conv gradient needs a bf16 unit output')
    if _is_bf16(yn) and not act_bf16:
        yn = _to_f32(yn)              # (not produced by the fused units: a bf16 y always comes with a bf16 output)
    y16 = int(_is_bf16(yn))
    if act_bf16:
        E.call('seg3d_gn_bwd_reduce_bf16', E.ptr(doutn), E.ptr(mask_src), E.ptr(yn), E.ptr(mean_rstd), E.ptr(gamma),
               E.ptr(beta), E.ptr(part), N, S, C, int(relu), ldd, y16, E.stream_ptr())
    else:
        E.call('seg3d_gn_bwd_reduce', E.ptr(doutn), E.ptr(mask_src), E.ptr(yn), E.ptr(mean_rstd), E.ptr(gamma),
               E.ptr(beta), E.ptr(part), N, S, C, int(relu), ldd, E.stream_ptr())
    abx = _empty((N, C, 3), yn)
    s12 = _empty((N, 2), yn)
    sg, sb_, sc = sinks
    for t in sinks:
        if t is not None and (t.numel() != C or not t.is_contiguous()):
            raise ValueError('GroupNorm gradient destination must be a contiguous [C] tensor')
    dgamma = _empty((C,), yn) if sg is None else sg
    dbeta = _empty((C,), yn) if sb_ is None else sb_
    dbias = (_empty((C,), yn) if sc is None else sc) if want_dbias else None
    acc_mask = (1 if sg is not None else 0) | (2 if sb_ is not None else 0) | (4 if (want_dbias and sc is not None) else 0)
    if GN_FUSED_FINALIZE:
        E.call('seg3d_gn_bwd_finalize_fused', E.ptr(part), E.ptr(gamma), E.ptr(mean_rstd), E.ptr(abx), E.ptr(s12),
               E.ptr(dgamma), E.ptr(dbeta), E.ptr(dbias), E.ptr(_gn_ticket(yn.device)), N, S, C, acc_mask, E.stream_ptr())
    else:
        E.call('seg3d_gn_bwd_finalize', E.ptr(part), E.ptr(gamma), E.ptr(mean_rstd), E.ptr(abx), E.ptr(s12), E.ptr(dgamma),
               E.ptr(dbeta), E.ptr(dbias), N, S, C, acc_mask, E.stream_ptr())
    dy = torch.empty(yn.shape, dtype=torch.bfloat16 if dy_bf16 else torch.float32, device=yn.device)
    dres = torch.empty(yn.shape, dtype=torch.float32, device=yn.device) if want_dres else None
    if act_bf16:
        E.call('seg3d_gn_bwd_apply_bf16', E.ptr(doutn), E.ptr(mask_src), E.ptr(yn), E.ptr(mean_rstd), E.ptr(s12),
               E.ptr(gamma), E.ptr(beta), E.ptr(dy), E.ptr(dres), N, S, C, int(relu), ldd, int(dy_bf16), y16,
               E.stream_ptr())
    else:
        E.call('seg3d_gn_bwd_apply', E.ptr(doutn), E.ptr(mask_src), E.ptr(yn), E.ptr(mean_rstd), E.ptr(s12), E.ptr(gamma),
               E.ptr(beta), E.ptr(dy), E.ptr(dres), N, S, C, int(relu), ldd, E.stream_ptr())
    return (dy, dres, dgamma if sg is None else None, dbeta if sb_ is None else None,
            dbias if (want_dbias and sc is None) else None)


# ------------------------------------------------------------------------------------------------------------------
# weight gradients on a side stream
# ------------------------------------------------------------------------------------------------------------------
# In backward the data-gradient chain (GroupNorm backward -> dgrad -> GroupNorm backward -> ...) is the critical path;
# a weight gradient only feeds the optimizer.  When its destination is a gradient sink (no tensor goes back to
# autograd) it is enqueued on a second HIP stream: the MFMA-bound weight-gradient kernel then shares the chip with the
# HBM-bound GroupNorm / stride-2 / thin kernels of the main stream instead of running in series with them.
# The main stream joins the side stream when backward finishes (engine callback); consumers that read gradients DURING
# backward (the bucketed all-reduce, core/ddp.py) call wgrad_stream_join() themselves.
WGRAD_SIDE_STREAM = True      # False (bench.py --no-wgrad-overlap, profiling): weight gradients on the main stream
_SIDE_STREAMS = {}
_SIDE_STREAMS_UNPROBED = {}   # device index -> provisional stream handed out during a capture (never memoised as the choice)
SIDE_KEEPALIVE = True     # inputs of side-stream weight gradients are held until the join (False: Tensor.record_stream)
_SIDE_KEEP = {}           # device index -> [(x, dy, launching stream)] of the weight gradients issued since the last join
_JOIN_QUEUED_FOR = [-1]   # id of the backward pass (graph task) whose end-of-backward join is already queued


def _runs_beside(busy, other, spin_cycles=1500000):
    """does a kernel launched on stream `other` run while an earlier kernel on stream `busy` is still executing?  HIP maps
    streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order, and two streams that land on one queue are
    serialised however independent their work is.  Probe: a busy-wait kernel on `busy`, then a small kernel on `other` -- on
    its own queue it completes long before the spin ends.  (The order matters with the null stream: a pool-stream kernel
    launched AFTER a null-stream kernel always waits for it on this stack, the reverse order overlaps -- and that is the order
    of the train step: a long weight gradient on the side stream first, the main stream's short kernels behind it.)"""
    probe = torch.zeros(1024, device=busy.device)
    votes = 0
    for _ in range(3):             # majority of three: a single reading can be spoilt by a late launch of the probe kernel
        torch.cuda.synchronize()
        s0, e_busy, e_other = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        s0.record(busy)
        other.wait_event(s0)
        with torch.cuda.stream(busy):
            torch.cuda._sleep(spin_cycles)
        e_busy.record(busy)
        with torch.cuda.stream(other):
            probe.add_(1.0)        # a real dispatch, not just a marker packet
        e_other.record(other)
        torch.cuda.synchronize()
        votes += int(s0.elapsed_time(e_other) < 0.5 * s0.elapsed_time(e_busy))
    return votes >= 2


def _side_stream(device):
    """the weight-gradient side stream of a device, created at first use: the first of a few fresh streams that demonstrably
    lets the current (main) stream's kernels run BESIDE its own.  Without the probe the choice is an accident of creation order
    -- with nothing but an RCCL communicator initialised in the process the next pool stream shares the main stream's hardware
    queue (tools/stream_probe.py), the overlap of weight gradients with the data-gradient chain is gone and the fp32 step reads
    17.05 instead of 15.97 ms
    (tools/ddp_overhead.py; GPU_MAX_HW_QUEUES = 3 / 5 / 8 also avoid that collision but 5+ queues slow hipGraph replays with
    side branches down by a third, so the default stays and the stream is chosen by measurement)."""
    st = _SIDE_STREAMS.get(device.index)
    if st is None:
        with torch.cuda.device(device):
            if torch.cuda.is_current_stream_capturing():
                # no probe inside a capture (it synchronises): hand out a provisional stream and do NOT remember it as the
                # choice -- the next eager use probes (TrainStep / GradientReducer call prepare_side_stream() in their
                # constructors, so a capture normally finds the probed stream already)
                st = _SIDE_STREAMS_UNPROBED.get(device.index)
                if st is None:
                    st = _SIDE_STREAMS_UNPROBED[device.index] = torch.cuda.Stream(device=device)
                return st
            main = torch.cuda.current_stream()
            first = _SIDE_STREAMS_UNPROBED.pop(device.index, None)
            for i in range(6):                     # candidates are created one at a time, only as long as the probe fails
                cand = first if (i == 0 and first is not None) else torch.cuda.Stream(device=device)
                if st is None:
                    st = cand                      # fallback when no candidate passes: the first one, as before the probe existed
                if _runs_beside(cand, main):
                    st = cand
                    break
        _SIDE_STREAMS[device.index] = st
    return st


def prepare_side_stream(device=None):
    """choose (probe) the weight-gradient side stream of `device` NOW, outside any backward pass or stream capture: the probe
    synchronises the device three times and launches spin kernels, which has no place inside the first backward or a
    gradient hook.  Called by TrainStep and GradientReducer when they are built; a no-op once the stream is chosen."""
    if not WGRAD_SIDE_STREAM:
        return None
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device())
    device = torch.device(device)
    if device.type != 'cuda':
        return None
    if device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    return _side_stream(device)


def wgrad_stream_join():
    """make the current stream wait for every weight gradient issued on the side stream so far"""
    cur = torch.cuda.current_stream()
    st = _SIDE_STREAMS.get(cur.device.index)
    if st is not None:
        cur.wait_stream(st)
        # everything the side stream read is safe to recycle behind this wait -- on the stream that waits.  A tensor whose
        # allocator pool belongs to ANOTHER stream (forward under torch.cuda.stream(s), backward() joined on a different one)
        # would become reusable on s before the side stream is done with it: those fall back to record_stream
        for xn, dyn, owner in _SIDE_KEEP.pop(cur.device.index, ()):
            if owner != cur.cuda_stream:
                xn.record_stream(st)
                dyn.record_stream(st)


def _join_after_backward():
    wgrad_stream_join()


def wgrad_side_stream(device=None):
    """the weight-gradient side stream of `device` (default: the current one), or None when weight gradients run on the main
    stream.  The gradient reducer (core/ddp.py) enqueues its collectives behind THIS stream instead of joining it into the
    main stream: the data-gradient chain is then never held up by a bucket launch."""
    if not WGRAD_SIDE_STREAM:
        return None
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device())
    return _side_stream(device)


def _wgrad_to_sink(xn, dyn, w_shape, kind, sink_view):
    """conv_wgrad accumulated into `sink_view`, on the side stream when enabled"""
    if not WGRAD_SIDE_STREAM:
        conv_wgrad(xn, dyn, w_shape, kind, out=sink_view)
        return
    # (inside a hipGraph capture the same fork / join becomes graph edges: the side stream joins the capture through
    # wait_stream and is joined back by the end-of-backward callback before the capture ends)
    side = _side_stream(dyn.device)
    side.wait_stream(torch.cuda.current_stream())       # x and dy are complete on the main stream
    with torch.cuda.stream(side):
        conv_wgrad(xn, dyn, w_shape, kind, out=sink_view)
    # the allocator must not recycle x / dy before the side stream is done with them: they are kept alive until the main
    # stream has joined the side stream (wgrad_stream_join drops the references).  Tensor.record_stream would do the same
    # through per-block events in the caching allocator: ~30 x 2 calls and event queries per backward on the launching thread
    # (tools/ab_step.py SIDE_KEEPALIVE=True / False, three pairs, eager: 15.69 / 15.79 / 15.74 against 15.88 / 15.87 / 15.87 ms;
    # a reusable event + bare set_stream calls instead of wait_stream + the stream context manager: no change)
    if SIDE_KEEPALIVE:
        _SIDE_KEEP.setdefault(dyn.device.index, []).append((xn, dyn, torch.cuda.current_stream().cuda_stream))
    else:
        xn.record_stream(side)
        dyn.record_stream(side)
    task = torch._C._current_graph_task_id()
    if task != _JOIN_QUEUED_FOR[0]:                      # once per backward pass (also after one that raised)
        _JOIN_QUEUED_FOR[0] = task
        torch.autograd.Variable._execution_engine.queue_callback(_join_after_backward)


# ------------------------------------------------------------------------------------------------------------------
# autograd functions
# ------------------------------------------------------------------------------------------------------------------
def _views(*sinks):
    return tuple(None if s is None else s.view for s in sinks)


class ResidualLink(object):
    """Side channel between the LAST unit of a residual block (which receives the block input as `residual`) and its
    FIRST unit (whose conv input is that same tensor).  In backward the last unit runs first and parks the gradient of
    the identity path here; the first unit's data-gradient kernel adds it in its epilogue, so the block input's
    gradient is written once instead of being summed by a separate elementwise pass over two full tensors
    (residual_block3.py:24: act(input + ops(input)))."""

    def __init__(self):
        self.grad = None


class ConvGnActFunction(torch.autograd.Function):
    """out = act(GroupNorm_1(conv(x) + bias) [+ residual]); conv kind in {'k3','k2s2','k1','convT'}.
    link_in: ResidualLink whose parked gradient this unit adds to its input gradient (first unit of a block);
    link_out: ResidualLink where this unit parks the residual gradient instead of returning it (last unit).
    When x is the residual itself (single-conv block) the two gradients are fused without a link."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, residual, kind, relu, eps, link_in=None, link_out=None, out_slot=None,
                no_backward=False):
        E.require_device(x, weight, bias, gamma, beta, residual)
        # (the inference-only guard of out_slot and `no_backward` come from conv_gn_act(): grad mode is always off inside
        # Function.forward, and needs_input_grad is set for parameters under torch.no_grad() as well)
        # a channel-slice input is read in place only by the stride-2 conv, and only when no backward will follow, whose
        # weight-gradient kernel would read the saved slice as packed rows
        xn = to_ndhwc(x, allow_slice=(kind == 'k2s2' and no_backward))
        w = weight.detach()
        cout = w.shape[1] if kind == 'convT' else w.shape[0]
        want16 = BF16_CONV_OUTPUT and _out_bf16(cout) and (_is_bf16(xn) or (kind == 'k3' and xn.shape[4] <= 8))
        yn, partial = conv_forward(xn, w, None if bias is None else bias.detach(), kind, want_stats=True, out_bf16=want16)
        mean_rstd = gn_stats(yn, partial, eps)
        resn = None
        if residual is not None:
            resn = to_ndhwc(residual)
            if resn.shape != yn.shape:
                raise ValueError('residual shape {} does not match conv output {}'.format(tuple(residual.shape),
                                                                                          tuple(from_ndhwc(yn).shape)))
        C = yn.shape[4]
        outn = gn_apply(yn, mean_rstd, gamma.detach(), beta.detach(), resn, relu, out=out_slot, out_bf16=_out_bf16(C))
        # the gradient w.r.t. the conv output goes to the dgrad / wgrad kernels as bf16 when they take bf16 (same
        # condition as in forward: bf16 input activations and MFMA-shaped channel counts on both sides)
        ctx.dy_bf16 = _is_bf16(xn) and _is_bf16(outn) and xn.shape[4] % 16 == 0
        if kind == 'k3' and not _is_bf16(xn) and _is_bf16(outn) and xn.shape[4] <= 8 and not ctx.needs_input_grad[0] \
                and not FORCE_DIRECT:
            ctx.dy_bf16 = True     # stem (fp32 image in, bf16 out, no input gradient): its weight gradient takes a bf16 dy
        ctx.kind, ctx.relu = kind, bool(relu)
        ctx.has_bias, ctx.has_res = bias is not None, residual is not None
        ctx.w_shape = tuple(weight.shape)
        ctx.res_is_x = residual is not None and residual is x
        ctx.link_in, ctx.link_out = link_in, (link_out if residual is not None else None)
        ctx.sinks = (G.lookup(weight), G.lookup(bias), G.lookup(gamma), G.lookup(beta))
        # the forward output is kept for backward only when a residual was added (otherwise the ReLU mask is
        # recomputed from y, saving one full-tensor read in each of the two GroupNorm backward passes)
        ctx.save_for_backward(xn, w, gamma.detach(), beta.detach(), yn, outn if residual is not None else None, mean_rstd)
        return from_ndhwc(outn)

    @staticmethod
    def backward(ctx, dout):
        xn, w, gamma, beta, yn, outn, mean_rstd = ctx.saved_tensors
        dn = to_ndhwc(dout)
        sw, sb_, sg, sbt = ctx.sinks
        dy, dres, dgamma, dbeta, dbias = gn_backward(dn, outn, yn, mean_rstd, gamma, beta, ctx.relu,
                                                     want_dres=ctx.has_res and ctx.needs_input_grad[5],
                                                     want_dbias=ctx.has_bias, sinks=_views(sg, sbt, sb_),
                                                     dy_bf16=ctx.dy_bf16)
        dx = None
        addend = None
        if ctx.res_is_x and dres is not None and ctx.needs_input_grad[0]:
            addend, dres = dres, None                      # single-conv block: identity-path gradient joins dx here
        elif ctx.link_out is not None and dres is not None:
            ctx.link_out.grad, dres = dres, None           # parked for the block's first unit (runs later in backward)
        if ctx.link_in is not None and ctx.link_in.grad is not None:
            if addend is not None or not ctx.needs_input_grad[0]:
                raise RuntimeError('ResidualLink misuse: the linked unit must produce exactly one input gradient')
            addend, ctx.link_in.grad = ctx.link_in.grad, None
        if ctx.needs_input_grad[0]:
            dx = from_ndhwc(conv_dgrad(dy, w, ctx.kind, addend=addend, want_bf16=_is_bf16(xn)))
        dw = None
        if ctx.needs_input_grad[1]:
            if sw is not None:
                _wgrad_to_sink(xn, dy, ctx.w_shape, ctx.kind, sw.view)
            else:
                dw = conv_wgrad(xn, dy, ctx.w_shape, ctx.kind)
        return (dx, dw, dbias if ctx.has_bias else None, dgamma, dbeta,
                from_ndhwc(dres) if dres is not None else None, None, None, None, None, None, None, None)


def conv_gn_act(x, weight, bias, gamma, beta, residual=None, kind='k3', relu=True, eps=GN_EPS, link_in=None,
                link_out=None, out_slot=None):
    """out_slot (inference only): [N,D,H,W,C] channel slice of a wider NDHWC buffer the unit's output is written into"""
    no_backward = not (torch.is_grad_enabled() and any(
        t is not None and t.requires_grad for t in (x, weight, bias, gamma, beta, residual)))
    if out_slot is not None and not no_backward:
        raise RuntimeError('out_slot (output written into a slice of another buffer) is an inference-only path: '
                           'call it under torch.no_grad()')
    return ConvGnActFunction.apply(x, weight, bias, gamma, beta, residual, kind, relu, eps, link_in, link_out, out_slot,
                                   no_backward)


class UpCatFunction(torch.autograd.Function):
    """cat((act(GroupNorm_1(convT(x) + bias)), skip), 1) -- the head of an UpBlock (network/module/vnet_upblock.py:19-21)
    without the two copies of a separate concatenation: the normalised up-branch is written straight into its channel
    slice of the concatenated buffer (gn_apply with a row stride) and only `skip` is copied; in backward GroupNorm
    reads its gradient in place from the slice of the incoming gradient (row stride), and the skip gradient goes back
    to autograd as a strided view of it (no copy either)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, skip, relu, eps, link_out=None, cat_buf=None):
        E.require_device(x, weight, bias, gamma, beta, skip)
        ctx.link_out = link_out
        xn = to_ndhwc(x)
        sn = to_ndhwc(skip, allow_slice=True)   # (copied into the concatenated buffer, or checked against cat_buf, below)
        w = weight.detach()
        want16 = BF16_CONV_OUTPUT and _is_bf16(xn) and _is_bf16(sn) and _out_bf16(w.shape[1])
        yn, partial = conv_forward(xn, w, None if bias is None else bias.detach(), 'convT', want_stats=True, out_bf16=want16)
        if sn.shape[:4] != yn.shape[:4]:
            raise ValueError('cat: spatial shapes differ: {} vs {}'.format(tuple(from_ndhwc(yn).shape), tuple(skip.shape)))
        mean_rstd = gn_stats(yn, partial, eps)
        N, D, H, W_, Ca = yn.shape
        Cb = sn.shape[4]
        if Ca % 4 or Cb % 4:
            raise ValueError('fused up + cat needs channel counts that are multiples of 4')
        bf = _is_bf16(sn)          # bf16 mode: the skip arrives as bf16, the concatenated buffer is bf16 as a whole
        if bf and (Ca % 16 or Cb % 16):
            raise ValueError('fused up + cat in bf16 mode needs channel counts that are multiples of 16')
        if cat_buf is not None:
            # inference: the skip tensor already lives in the second half of the concatenated buffer (the encoder stage that
            # produced it wrote it there, VNetBase.forward), so nothing is copied
            if tuple(cat_buf.shape) != (N, D, H, W_, Ca + Cb) or not cat_buf.is_contiguous() or cat_buf.dtype != sn.dtype or \
                    sn.data_ptr() != cat_buf[..., Ca:].data_ptr() or sn.stride(3) != Ca + Cb:
                raise ValueError('up_cat: cat_buf must be the contiguous [N,D,H,W,Ca+Cb] buffer whose slice [..., Ca:] is `skip`')
            cat = cat_buf
            gn_apply(yn, mean_rstd, gamma.detach(), beta.detach(), None, relu, out=cat[..., :Ca])
        else:
            if not sn.is_contiguous():
                sn = sn.contiguous()
            cat = _empty((N, D, H, W_, Ca + Cb), yn, torch.bfloat16 if bf else torch.float32)
            gn_apply(yn, mean_rstd, gamma.detach(), beta.detach(), None, relu, out=cat[..., :Ca])
            k = 2 if bf else 1         # the copy kernel moves 32-bit words: a bf16 row of C channels is C / 2 of them
            E.call('seg3d_copy_channels', E.ptr(sn), E.ptr(cat), N * D * H * W_, Cb // k, Cb // k, 0, (Ca + Cb) // k, Ca // k,
                   E.stream_ptr())
        ctx.dy_bf16 = bf and _is_bf16(xn) and xn.shape[4] % 16 == 0
        ctx.relu, ctx.has_bias, ctx.w_shape, ctx.ca = bool(relu), bias is not None, tuple(weight.shape), Ca
        ctx.sinks = (G.lookup(weight), G.lookup(bias), G.lookup(gamma), G.lookup(beta))
        ctx.save_for_backward(xn, w, gamma.detach(), beta.detach(), yn, mean_rstd)
        return from_ndhwc(cat)

    @staticmethod
    def backward(ctx, dout):
        xn, w, gamma, beta, yn, mean_rstd = ctx.saved_tensors
        dn = to_ndhwc(dout)                                  # [N, D, H, W, Ca + Cb], contiguous
        sw, sb_, sg, sbt = ctx.sinks
        dy, _, dgamma, dbeta, dbias = gn_backward(dn[..., :ctx.ca], None, yn, mean_rstd, gamma, beta, ctx.relu,
                                                  want_dres=False, want_dbias=ctx.has_bias, sinks=_views(sg, sbt, sb_),
                                                  dy_bf16=ctx.dy_bf16)
        dx = from_ndhwc(conv_dgrad(dy, w, 'convT')) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            if sw is not None:
                _wgrad_to_sink(xn, dy, ctx.w_shape, 'convT', sw.view)
            else:
                dw = conv_wgrad(xn, dy, ctx.w_shape, 'convT')
        dskip = None
        if ctx.needs_input_grad[5]:
            if ctx.link_out is not None:
                # the skip tensor's other consumer (the next DownBlock's stride-2 conv) runs later in backward and adds
                # this slice in its data-gradient epilogue: no elementwise pass over two full tensors
                ctx.link_out.grad = dn[..., ctx.ca:]
            else:
                dskip = from_ndhwc(dn[..., ctx.ca:])                                  # strided view, no copy
        return dx, dw, dbias if ctx.has_bias else None, dgamma, dbeta, dskip, None, None, None, None


def up_cat(x, weight, bias, gamma, beta, skip, relu=True, eps=GN_EPS, link_out=None, cat_buf=None):
    return UpCatFunction.apply(x, weight, bias, gamma, beta, skip, relu, eps, link_out, cat_buf)


class ConvFunction(torch.autograd.Function):
    """plain convolution + bias (nn.Conv3d / nn.ConvTranspose3d called on their own)"""

    @staticmethod
    def forward(ctx, x, weight, bias, kind):
        E.require_device(x, weight, bias)
        xn = to_ndhwc(x)
        w = weight.detach()
        yn, _ = conv_forward(xn, w, None if bias is None else bias.detach(), kind)
        ctx.kind, ctx.has_bias, ctx.w_shape = kind, bias is not None, tuple(weight.shape)
        ctx.sinks = (G.lookup(weight), G.lookup(bias))
        ctx.save_for_backward(xn, w)
        return from_ndhwc(yn)

    @staticmethod
    def backward(ctx, dout):
        xn, w = ctx.saved_tensors
        dn = to_ndhwc(dout)
        dx = from_ndhwc(conv_dgrad(dn, w, ctx.kind)) if ctx.needs_input_grad[0] else None
        sw, sb_ = ctx.sinks
        dw = None
        if ctx.needs_input_grad[1]:
            dw = conv_wgrad(xn, dn, ctx.w_shape, ctx.kind, out=None if sw is None else sw.view)
            if sw is not None:
                dw = None
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dn.reshape(-1, dn.shape[-1]).sum(0)  # plumbing-only path (bias of a bare conv); fused path uses gn_bwd
            if sb_ is not None:
                sb_.view.add_(db.view_as(sb_.view))
                db = None
        return dx, dw, db, None


def conv(x, weight, bias, kind):
    return ConvFunction.apply(x, weight, bias, kind)


class GroupNormFunction(torch.autograd.Function):
    """GroupNorm(1, C) [+ ReLU] on its own (nn.GroupNorm called outside the fused unit)"""

    @staticmethod
    def forward(ctx, x, gamma, beta, relu, eps):
        E.require_device(x, gamma, beta)
        yn = to_ndhwc(x)
        mean_rstd = gn_stats(yn, None, eps)
        outn = gn_apply(yn, mean_rstd, gamma.detach(), beta.detach(), None, relu)
        ctx.relu = bool(relu)
        ctx.sinks = (G.lookup(gamma), G.lookup(beta))
        ctx.save_for_backward(yn, mean_rstd, gamma.detach(), beta.detach())
        return from_ndhwc(outn)

    @staticmethod
    def backward(ctx, dout):
        yn, mean_rstd, gamma, beta = ctx.saved_tensors
        dn = to_ndhwc(dout)
        sg, sbt = ctx.sinks
        dy, _, dgamma, dbeta, _ = gn_backward(dn, None, yn, mean_rstd, gamma, beta, ctx.relu, want_dres=False,
                                              want_dbias=False, sinks=_views(sg, sbt, None))
        return from_ndhwc(dy), dgamma, dbeta, None, None


def group_norm(x, gamma, beta, relu=False, eps=GN_EPS):
    return GroupNormFunction.apply(x, gamma, beta, relu, eps)


class CatChannelsFunction(torch.autograd.Function):
    """torch.cat((a, b), 1) written straight into one NDHWC buffer (network/module/vnet_upblock.py:21)"""

    @staticmethod
    def forward(ctx, a, b):
        E.require_device(a, b)
        an, bn = to_ndhwc(a), to_ndhwc(b)
        if an.shape[:4] != bn.shape[:4]:
            raise ValueError('cat: spatial shapes differ: {} vs {}'.format(tuple(a.shape), tuple(b.shape)))
        N, D, H, W_, Ca = an.shape
        Cb = bn.shape[4]
        out = _empty((N, D, H, W_, Ca + Cb), an)
        nvox = N * D * H * W_
        E.call('seg3d_copy_channels', E.ptr(an), E.ptr(out), nvox, Ca, Ca, 0, Ca + Cb, 0, E.stream_ptr())
        E.call('seg3d_copy_channels', E.ptr(bn), E.ptr(out), nvox, Cb, Cb, 0, Ca + Cb, Ca, E.stream_ptr())
        ctx.ca, ctx.cb = Ca, Cb
        return from_ndhwc(out)

    @staticmethod
    def backward(ctx, dout):
        dn = to_ndhwc(dout)
        N, D, H, W_, C = dn.shape
        nvox = N * D * H * W_
        da = db = None
        if ctx.needs_input_grad[0]:
            dan = _empty((N, D, H, W_, ctx.ca), dn)
            E.call('seg3d_copy_channels', E.ptr(dn), E.ptr(dan), nvox, ctx.ca, C, 0, ctx.ca, 0, E.stream_ptr())
            da = from_ndhwc(dan)
        if ctx.needs_input_grad[1]:
            dbn = _empty((N, D, H, W_, ctx.cb), dn)
            E.call('seg3d_copy_channels', E.ptr(dn), E.ptr(dbn), nvox, ctx.cb, C, ctx.ca, ctx.cb, 0, E.stream_ptr())
            db = from_ndhwc(dbn)
        return da, db


def cat_channels(a, b):
    return CatChannelsFunction.apply(a, b)


class SoftmaxFunction(torch.autograd.Function):
    """nn.Softmax(dim=1); emits a contiguous NCDHW tensor (the plugin API's output layout)"""

    @staticmethod
    def forward(ctx, x):
        E.require_device(x)
        xn = to_ndhwc(x)
        N, D, H, W_, C = xn.shape
        probs = _empty((N, C, D, H, W_), xn)
        E.call('seg3d_softmax_fwd', E.ptr(xn), E.ptr(probs), N, C, D * H * W_, E.stream_ptr())
        ctx.save_for_backward(probs)
        return probs

    @staticmethod
    def backward(ctx, dprobs):
        (probs,) = ctx.saved_tensors
        N, C, D, H, W_ = probs.shape
        dp = dprobs.contiguous()
        din = _empty((N, D, H, W_, C), probs)
        E.call('seg3d_softmax_bwd', E.ptr(probs), E.ptr(dp), E.ptr(din), N, C, D * H * W_, E.stream_ptr())
        return from_ndhwc(din)


def softmax_channels(x):
    return SoftmaxFunction.apply(x)


class DiceLossFunction(torch.autograd.Function):
    """MultiDiceLoss.forward (loss/multi_dice_loss.py:24-43) as one fused reduction + one elementwise backward"""

    @staticmethod
    def forward(ctx, probs, target, weights):
        E.require_device(probs, target, weights)
        p = probs.contiguous()
        t = target.contiguous().float()
        N, C = p.shape[0], p.shape[1]
        S = p[0, 0].numel()
        if t.numel() != N * S:
            raise ValueError('target shape {} does not match input {}'.format(tuple(target.shape), tuple(probs.shape)))
        nblk = E.query('seg3d_dice_blocks', S)
        part = _empty((N, nblk, C, 3), p)
        sums = _empty((N, C, 2), p)
        loss = _empty((1,), p)
        E.call('seg3d_dice_fwd', E.ptr(p), E.ptr(t), E.ptr(weights), E.ptr(part), E.ptr(sums), E.ptr(loss), N, C, S,
               E.stream_ptr())
        ctx.save_for_backward(p, t, sums, weights)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        p, t, sums, weights = ctx.saved_tensors
        N, C = p.shape[0], p.shape[1]
        S = p[0, 0].numel()
        g = gout.contiguous().reshape(1).float()
        dp = torch.empty_like(p)
        E.call('seg3d_dice_bwd', E.ptr(p), E.ptr(t), E.ptr(sums), E.ptr(weights), E.ptr(g), E.ptr(dp), N, C, S,
               E.stream_ptr())
        return dp, None, None


class BinaryDiceLossFunction(torch.autograd.Function):
    """BinaryDiceLoss.forward called on its own (loss/binary_dice_loss.py:9-36): one fused reduction + one elementwise
    backward; probs [N, 2, ...], float target with N * S elements"""

    @staticmethod
    def forward(ctx, probs, target):
        E.require_device(probs, target)
        p = probs.contiguous()
        t = target.contiguous().float()
        N = p.shape[0]
        S = p[0, 0].numel()
        if p.shape[1] != 2 or t.numel() != N * S:
            raise ValueError('BinaryDiceLoss needs a [N, 2, ...] input and a target of N x spatial elements, got {} and {}'
                             .format(tuple(probs.shape), tuple(target.shape)))
        nblk = E.query('seg3d_dice_blocks', S)
        part = _empty((N, nblk, 3), p)
        sums = _empty((N, 2), p)
        loss = _empty((1,), p)
        one = torch.ones(1, dtype=torch.float32, device=p.device)
        E.call('seg3d_binary_dice_fwd', E.ptr(p), E.ptr(t), E.ptr(one), E.ptr(part), E.ptr(sums), E.ptr(loss), N, S,
               E.stream_ptr())
        ctx.save_for_backward(p, t, sums)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        p, t, sums = ctx.saved_tensors
        N = p.shape[0]
        S = p[0, 0].numel()
        g = gout.contiguous().reshape(1).float()
        dp = torch.empty_like(p)
        E.call('seg3d_binary_dice_bwd', E.ptr(p), E.ptr(t), E.ptr(sums), E.ptr(g), E.ptr(dp), N, S, E.stream_ptr())
        return dp, None


class FocalLossFunction(torch.autograd.Function):
    """FocalLoss.forward (loss/focal_loss.py:27-61); probs viewed as [N][C][S] through strides (sn, sc, ss)"""

    @staticmethod
    def forward(ctx, probs, target, alpha, gamma, size_average, N, C, S, sn, sc, ss):
        E.require_device(probs, target, alpha)
        p = probs.contiguous()
        t = target.contiguous().float()
        total = N * S
        if t.numel() != total:
            raise ValueError('target has {} elements, expected {}'.format(t.numel(), total))
        nblk = E.query('seg3d_focal_blocks', total)
        part = _empty((nblk,), p)
        loss = _empty((1,), p)
        E.call('seg3d_focal_fwd', E.ptr(p), E.ptr(t), E.ptr(alpha), E.ptr(part), E.ptr(loss), N, C, S, sn, sc, ss,
               float(gamma), int(bool(size_average)), E.stream_ptr())
        ctx.save_for_backward(p, t, alpha)
        ctx.cfg = (float(gamma), int(bool(size_average)), N, C, S, sn, sc, ss)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        p, t, alpha = ctx.saved_tensors
        gamma, size_average, N, C, S, sn, sc, ss = ctx.cfg
        g = gout.contiguous().reshape(1).float()
        dp = torch.empty_like(p)
        E.call('seg3d_focal_bwd', E.ptr(p), E.ptr(t), E.ptr(alpha), E.ptr(g), E.ptr(dp), N, C, S, sn, sc, ss, gamma,
               size_average, E.stream_ptr())
        return (dp,) + (None,) * 10
