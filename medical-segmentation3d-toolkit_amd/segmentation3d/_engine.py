"""ctypes binding of libseg3d_hip.so -- the only compute backend of this package.

There is deliberately no CPU or eager-PyTorch fallback: if the shared library is missing or a tensor is not on a
HIP device, the ops raise.  (The CPU restatement used to check results lives in /oracle and is test-only.)
The C ABI is declared in include/seg3d_hip.h.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get('SEG3D_HIP_LIB', os.path.join(_PKG_ROOT, 'lib', 'libseg3d_hip.so'))

_c_int, _c_ll, _c_f, _c_p = ctypes.c_int, ctypes.c_longlong, ctypes.c_float, ctypes.c_void_p

# name -> (restype, argtypes); keep in sync with include/seg3d_hip.h (tests/test_abi.py checks every symbol)
_SIGNATURES = {
    'seg3d_last_error': (ctypes.c_char_p, []),
    'seg3d_abi_version': (_c_int, []),
    'seg3d_target_arch': (ctypes.c_char_p, []),
    'seg3d_device_count': (_c_int, []),
    'seg3d_ncdhw_to_ndhwc': (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_ll, _c_p]),
    'seg3d_ndhwc_to_ncdhw': (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_ll, _c_p]),
    'seg3d_copy_channels': (_c_int, [_c_p, _c_p, _c_ll, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p]),
    'seg3d_pack_weights_tapmajor': (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_ll, _c_ll, _c_int, _c_p]),
    'seg3d_pack_weights_mfma': (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_ll, _c_ll, _c_int, _c_p]),
    'seg3d_packed_mfma_floats': (_c_ll, [_c_int, _c_int, _c_int]),
    'seg3d_pack_job_blocks': (_c_ll, [_c_int, _c_int, _c_int]),
    'seg3d_label_overlap_counts': (_c_int, [_c_p, _c_p, _c_int, _c_ll, _c_p, _c_int, _c_p, _c_p]),
    'seg3d_pack_weights_mfma_multi': (_c_int, [_c_p, _c_int, _c_ll, _c_p]),
    'seg3d_resample_affine': (_c_int, [_c_p, _c_p] + [_c_int] * 6 + [_c_p, _c_int, _c_f, _c_p]),
    'seg3d_mask_bounding_box': (_c_int, [_c_p, _c_int, _c_int, _c_int, _c_p, _c_int, _c_p, _c_p]),
    'seg3d_ccl_workspace_ints': (_c_ll, [_c_ll]),
    'seg3d_ccl26_select': (_c_int, [_c_p] + [_c_int] * 8 + [_c_p, _c_p, _c_p]),
    'seg3d_conv3d_fwd_direct': (_c_int, [_c_p, _c_p, _c_p, _c_p] + [_c_int] * 8 + [_c_p]),
    'seg3d_convT3d_k2s2_fwd_direct': (_c_int, [_c_p, _c_p, _c_p, _c_p] + [_c_int] * 6 + [_c_p]),
    'seg3d_wgrad_direct_workspace_floats': (_c_ll, [_c_int] * 7),
    'seg3d_wgrad_direct': (_c_int, [_c_p, _c_p, _c_p] + [_c_int] * 8 + [ctypes.POINTER(_c_int), _c_p]),
    'seg3d_wgrad_reduce': (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_ll, _c_ll, _c_int, _c_p]),
    'seg3d_conv3d_k3_mfma_stats_count': (_c_ll, [_c_int] * 6),
    'seg3d_conv3d_k3_mfma_fwd_workspace_floats': (_c_ll, [_c_int] * 6),
    'seg3d_conv3d_k3_mfma_variant': (_c_int, [_c_int] * 6),
    'seg3d_conv3d_k3_mfma_fwd': (_c_int, [_c_p] * 7 + [_c_int] * 6 + [_c_p]),
    'seg3d_packed_mfma_bf16_elems': (_c_ll, [_c_int, _c_int, _c_int]),
    'seg3d_pack_weights_mfma_bf16': (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_ll, _c_ll, _c_int, _c_p]),
    'seg3d_pack_job_blocks_bf16': (_c_ll, [_c_int, _c_int, _c_int]),
    'seg3d_pack_weights_mfma_bf16_multi': (_c_int, [_c_p, _c_int, _c_ll, _c_p]),
    'seg3d_f32_to_bf16': (_c_int, [_c_p, _c_p, _c_ll, _c_p]),
    'seg3d_bf16_to_f32': (_c_int, [_c_p, _c_p, _c_ll, _c_p]),
    'seg3d_conv3d_k3_bf16_stats_count': (_c_ll, [_c_int] * 6),
    'seg3d_conv3d_k3_bf16_fwd_workspace_floats': (_c_ll, [_c_int] * 6),
    'seg3d_conv3d_k3_bf16_variant': (_c_int, [_c_int] * 6),
    'seg3d_conv3d_k3_bf16_fwd': (_c_int, [_c_p] * 7 + [_c_int] * 7 + [_c_p]),
    'seg3d_conv3d_k3_bf16_wgrad_workspace_floats': (_c_ll, [_c_int] * 6),
    'seg3d_conv3d_k3_bf16_wgrad': (_c_int, [_c_p] * 4 + [_c_int] * 7 + [_c_p]),
    'seg3d_conv3d_k2s2_bf16_fwd': (_c_int, [_c_p] * 5 + [_c_int] * 8 + [_c_p]),
    'seg3d_convT3d_k2s2_bf16_fwd': (_c_int, [_c_p] * 5 + [_c_int] * 8 + [_c_p]),
    'seg3d_convT3d_k2s2_scatter_addend': (_c_int, [_c_p, _c_int, _c_p, _c_p, _c_int, _c_p] + [_c_int] * 7 + [_c_p]),
    'seg3d_k2_bf16_wgrad': (_c_int, [_c_p] * 4 + [_c_int] * 6 + [_c_ll, _c_ll, _c_int, _c_p]),
    'seg3d_conv3d_k3_thin_out_bf16_fwd': (_c_int, [_c_p] * 5 + [_c_int] * 7 + [_c_p]),
    'seg3d_conv3d_k3_thin_in_mfma16_supported': (_c_int, [_c_int] * 2),
    'seg3d_packed_thin_in16_elems': (_c_ll, [_c_int] * 2),
    'seg3d_pack_weights_thin_in16': (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_ll, _c_ll, _c_int, _c_p]),
    'seg3d_conv3d_k3_thin_in_mfma16_fwd': (_c_int, [_c_p] * 5 + [_c_int] * 6 + [_c_p]),
    'seg3d_conv3d_k3_thin_out_mfma_supported': (_c_int, [_c_int] * 2),
    'seg3d_thin_out_mfma_packed_elems': (_c_ll, [_c_int]),
    'seg3d_pack_weights_thin_out_mfma': (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_ll, _c_ll, _c_int, _c_p]),
    'seg3d_conv3d_k3_thin_out_mfma_fwd': (_c_int, [_c_p] * 5 + [_c_int] * 6 + [_c_p]),
    'seg3d_gn_apply_mixed': (_c_int, [_c_p] * 6 + [_c_int, _c_ll, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p]),
    'seg3d_gn_bwd_reduce_bf16': (_c_int, [_c_p] * 7 + [_c_int, _c_ll, _c_int, _c_int, _c_int, _c_int, _c_p]),
    'seg3d_gn_bwd_apply_bf16': (_c_int, [_c_p] * 9 + [_c_int, _c_ll, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p]),
    'seg3d_conv3d_k3_mfma_wgrad_workspace_floats': (_c_ll, [_c_int] * 6),
    'seg3d_conv3d_k3_mfma_wgrad': (_c_int, [_c_p] * 4 + [_c_int] * 7 + [_c_p]),
    'seg3d_conv3d_k2s2_mfma_stats_count': (_c_ll, [_c_int] * 4),
    'seg3d_conv3d_k2s2_mfma_fwd': (_c_int, [_c_p] * 5 + [_c_int] * 6 + [_c_p]),
    'seg3d_conv3d_k2s2_mfma_fwd_ld': (_c_int, [_c_p, _c_int] + [_c_p] * 4 + [_c_int] * 6 + [_c_p]),
    'seg3d_convT3d_k2s2_mfma_stats_count': (_c_ll, [_c_int] * 4),
    'seg3d_convT3d_k2s2_mfma_fwd': (_c_int, [_c_p] * 5 + [_c_int] * 6 + [_c_p]),
    'seg3d_k2_mfma_wgrad_workspace_floats': (_c_ll, [_c_int] * 6),
    'seg3d_k2_mfma_wgrad': (_c_int, [_c_p] * 4 + [_c_int] * 6 + [_c_ll, _c_ll, _c_int, _c_p]),
    'seg3d_packed_thin_in_floats': (_c_ll, [_c_int, _c_int]),
    'seg3d_pack_weights_thin_in': (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_ll, _c_ll, _c_int, _c_p]),
    'seg3d_conv3d_k3_thin_stats_count': (_c_ll, [_c_int] * 4),
    'seg3d_conv3d_k3_thin_in_fwd': (_c_int, [_c_p] * 5 + [_c_int] * 6 + [_c_p]),
    'seg3d_conv3d_k3_thin_in_bf16out_fwd': (_c_int, [_c_p] * 5 + [_c_int] * 6 + [_c_p]),
    'seg3d_conv3d_k3_thin_in_persistent_stats_count': (_c_ll, [_c_int] * 4),
    'seg3d_conv3d_k3_thin_in_persistent_fwd': (_c_int, [_c_p] * 5 + [_c_int] * 7 + [_c_p]),
    'seg3d_conv3d_k3_wino_supported': (_c_int, [_c_int] * 6),
    'seg3d_conv3d_k3_wino_preferred': (_c_int, [_c_int] * 6),
    'seg3d_conv3d_k3_wino_stats_count': (_c_ll, [_c_int] * 6),
    'seg3d_conv3d_k3_wino_fwd': (_c_int, [_c_p] * 6 + [_c_int] * 6 + [_c_p]),
    'seg3d_conv3d_k3_wino2d_supported': (_c_int, [_c_int] * 6),
    'seg3d_conv3d_k3_wino2d_preferred': (_c_int, [_c_int] * 6),
    'seg3d_conv3d_k3_wino2d_stats_count': (_c_ll, [_c_int] * 6),
    'seg3d_conv3d_k3_wino2d_fwd': (_c_int, [_c_p] * 6 + [_c_int] * 6 + [_c_p]),
    'seg3d_conv3d_k3_wino2d_fwd_workspace_floats': (_c_ll, [_c_int] * 6),
    'seg3d_conv3d_k3_wino2d_fwd_ws': (_c_int, [_c_p] * 7 + [_c_int] * 6 + [_c_p]),
    'seg3d_conv3d_k3_wino2d_wgrad_supported': (_c_int, [_c_int] * 6),
    'seg3d_conv3d_k3_wino2d_wgrad_preferred': (_c_int, [_c_int] * 6),
    'seg3d_conv3d_k3_wino2d_wgrad_workspace_floats': (_c_ll, [_c_int] * 6),
    'seg3d_conv3d_k3_wino2d_wgrad': (_c_int, [_c_p] * 4 + [_c_int] * 7 + [_c_p]),
    'seg3d_conv3d_k3_wino_wgrad_supported': (_c_int, [_c_int] * 6),
    'seg3d_conv3d_k3_wino_wgrad_workspace_floats': (_c_ll, [_c_int] * 6),
    'seg3d_conv3d_k3_wino_wgrad': (_c_int, [_c_p] * 4 + [_c_int] * 7 + [_c_p]),
    'seg3d_conv3d_k3_thin_out_f32mfma_supported': (_c_int, [_c_int] * 2),
    'seg3d_thin_out_f32mfma_packed_floats': (_c_ll, [_c_int] * 2),
    'seg3d_pack_weights_thin_out_f32mfma': (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_ll, _c_ll, _c_int, _c_p]),
    'seg3d_conv3d_k3_thin_out_f32mfma_stats_count': (_c_ll, [_c_int] * 4),
    'seg3d_conv3d_k3_thin_out_f32mfma_fwd': (_c_int, [_c_p] * 5 + [_c_int] * 6 + [_c_p]),
    'seg3d_pack_weights_thin_out': (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_ll, _c_ll, _c_int, _c_p]),
    'seg3d_conv3d_k3_thin_out_stats_count': (_c_ll, [_c_int] * 3),
    'seg3d_conv3d_k3_thin_out_fwd': (_c_int, [_c_p] * 5 + [_c_int] * 7 + [_c_p]),
    'seg3d_k3_thin_wgrad_workspace_floats': (_c_ll, [_c_int] * 6),
    'seg3d_k3_thin_wgrad': (_c_int, [_c_p] * 4 + [_c_int] * 6 + [_c_ll, _c_ll, _c_int, _c_int, _c_p]),
    'seg3d_k3_thin_wgrad_fatbf16': (_c_int, [_c_p] * 4 + [_c_int] * 6 + [_c_ll, _c_ll, _c_int, _c_int, _c_p]),
    'seg3d_gn_stats_count': (_c_ll, [_c_ll]),
    'seg3d_gn_stats_partial': (_c_int, [_c_p, _c_p, _c_int, _c_ll, _c_p]),
    'seg3d_gn_stats_finalize': (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_ll, _c_f, _c_p]),
    'seg3d_gn_apply': (_c_int, [_c_p] * 6 + [_c_int, _c_ll, _c_int, _c_int, _c_int, _c_p]),
    'seg3d_gn_bwd_blocks': (_c_ll, [_c_ll]),
    'seg3d_gn_bwd_reduce': (_c_int, [_c_p] * 7 + [_c_int, _c_ll, _c_int, _c_int, _c_int, _c_p]),
    'seg3d_gn_bwd_finalize': (_c_int, [_c_p] * 8 + [_c_int, _c_ll, _c_int, _c_int, _c_p]),
    'seg3d_gn_bwd_finalize_fused': (_c_int, [_c_p] * 9 + [_c_int, _c_ll, _c_int, _c_int, _c_p]),
    'seg3d_gn_bwd_apply': (_c_int, [_c_p] * 9 + [_c_int, _c_ll, _c_int, _c_int, _c_int, _c_p]),
    'seg3d_softmax_fwd': (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_ll, _c_p]),
    'seg3d_softmax_bwd': (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_ll, _c_p]),
    'seg3d_dice_blocks': (_c_ll, [_c_ll]),
    'seg3d_dice_fwd': (_c_int, [_c_p] * 6 + [_c_int, _c_int, _c_ll, _c_p]),
    'seg3d_dice_bwd': (_c_int, [_c_p] * 6 + [_c_int, _c_int, _c_ll, _c_p]),
    'seg3d_binary_dice_fwd': (_c_int, [_c_p] * 6 + [_c_int, _c_ll, _c_p]),
    'seg3d_binary_dice_bwd': (_c_int, [_c_p] * 5 + [_c_int, _c_ll, _c_p]),
    'seg3d_focal_blocks': (_c_ll, [_c_ll]),
    'seg3d_focal_fwd': (_c_int, [_c_p] * 5 + [_c_int, _c_int, _c_ll, _c_ll, _c_ll, _c_ll, _c_f, _c_int, _c_p]),
    'seg3d_focal_bwd': (_c_int, [_c_p] * 5 + [_c_int, _c_int, _c_ll, _c_ll, _c_ll, _c_ll, _c_f, _c_int, _c_p]),
    'seg3d_adam_step': (_c_int, [_c_p] * 4 + [_c_ll, _c_int] + [_c_f] * 6 + [_c_p]),
    'seg3d_adam_step_devstep': (_c_int, [_c_p] * 4 + [_c_ll, _c_p, _c_p] + [_c_f] * 6 + [_c_p]),
    'seg3d_patch_stats_blocks': (_c_ll, [_c_int] * 3),
    'seg3d_patch_gather_normalize': (_c_int, [_c_p] * 5 + [_c_int] * 8 + [_c_f, _c_f, _c_int, _c_f, _c_p]),
    'seg3d_patch_scatter_accumulate': (_c_int, [_c_p] * 5 + [_c_int] * 7 + [_c_ll, _c_p]),
    'seg3d_finalize_argmax': (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_ll, _c_ll, _c_p]),
}

_lib = None


class Seg3dEngineError(RuntimeError):
    pass


def symbols():
    """names of all bound C-ABI entry points"""
    return sorted(_SIGNATURES)


def lib():
    """load (once) and return the ctypes handle; raises if the HIP library has not been built"""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise Seg3dEngineError(
                'libseg3d_hip.so not found at {} -- build it with `python __graft_entry__.py` '
                '(segmentation3d has no CPU fallback)'.format(LIB_PATH))
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.seg3d_abi_version() != 1:
            raise Seg3dEngineError('libseg3d_hip.so ABI version mismatch')
        _lib = handle
    return _lib


def last_error():
    return lib().seg3d_last_error().decode('utf-8', 'replace')


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream_ptr():
    """hipStream_t of torch's current stream (so launches are captured by torch.cuda.graph and ordered with torch ops).
    Called once per kernel launch: the raw-handle query is ~10x cheaper than building a torch.cuda.Stream object."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """device pointer of a tensor (None -> NULL)"""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def require_device(*tensors):
    """every operator entry checks its tensors here: on a ROCm device, and on the CURRENT one -- stream_ptr() hands the
    kernels the current device's stream, so a tensor of another GPU would be launched on the wrong device's queue
    (callers pin the device: load_single_model / TrainStep call torch.cuda.set_device, the sliding window runs under
    torch.cuda.device(volume.device))"""
    current = None
    for t in tensors:
        if t is None:
            continue
        if not isinstance(t, torch.Tensor):
            raise TypeError('expected a torch.Tensor, got {}'.format(type(t)))
        if not t.is_cuda:
            raise Seg3dEngineError(
                'segmentation3d HIP engine needs tensors on a ROCm device (got device={}); '
                'there is no CPU path in this package'.format(t.device))
        if current is None:
            current = torch.cuda.current_device()
        if t.device.index is not None and t.device.index != current:
            raise Seg3dEngineError(
                'tensor on {} but the current device is cuda:{} -- segmentation3d launches on the current device\'s stream; '
                'select the device first (torch.cuda.set_device / `with torch.cuda.device(...)`)'.format(t.device, current))
        if t.dtype not in (torch.float32, torch.int32, torch.int8, torch.float64, torch.uint8, torch.int16, torch.int64,
                           torch.bfloat16):
            raise TypeError('unsupported dtype {}'.format(t.dtype))


def call(name, *args):
    """invoke an int-returning entry point; raise with the library's message on failure"""
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        msg = last_error()
        if rc == -3:
            raise NotImplementedError(msg)
        if rc == -1:
            raise ValueError(msg)
        raise Seg3dEngineError('{} failed (code {}): {}'.format(name, rc, msg))
    return rc


_QUERY_CACHE = {}


def query(name, *args):
    """invoke a size/count helper (returns a number).  The helpers are pure functions of their integer arguments
    (tile plans, workspace sizes); results are memoised -- a train step asks ~50 of them, always the same ones."""
    key = (name,) + args
    v = _QUERY_CACHE.get(key)
    if v is None:
        v = int(getattr(lib(), name)(*args))
        _QUERY_CACHE[key] = v
    return v
