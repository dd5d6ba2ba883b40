"""ORACLE (test infrastructure, not product code) -- ctypes wrapper of oracle/c/liboracle.so (plain C, double
accumulation).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'c', 'liboracle.so')
        if not os.path.isfile(path):
            subprocess.run(['make', '-s', '-C', _HERE], check=True)
        _LIB = ctypes.CDLL(path)
        _LIB.oracle_multi_dice.restype = ctypes.c_float
        _LIB.oracle_focal.restype = ctypes.c_float
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def conv3d(x, w, b, k, s, p):
    x, w, b = _f(x), _f(w), _f(b)
    N, Cin, D, H, W = x.shape
    Cout = w.shape[0]
    out = np.empty((N, Cout, (D + 2 * p - k) // s + 1, (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1), np.float32)
    lib().oracle_conv3d(_p(x), _p(w), _p(b), _p(out), N, Cin, D, H, W, Cout, k, s, p)
    return out


def conv_transpose3d_k2s2(x, w, b):
    x, w, b = _f(x), _f(w), _f(b)
    N, Cin, D, H, W = x.shape
    Cout = w.shape[1]
    out = np.empty((N, Cout, 2 * D, 2 * H, 2 * W), np.float32)
    lib().oracle_conv_transpose3d_k2s2(_p(x), _p(w), _p(b), _p(out), N, Cin, D, H, W, Cout)
    return out


def group_norm1(x, gamma, beta, res=None, relu=False, eps=1e-5):
    x, gamma, beta, res = _f(x), _f(gamma), _f(beta), _f(res)
    N, C = x.shape[:2]
    S = int(np.prod(x.shape[2:]))
    out = np.empty_like(x)
    lib().oracle_group_norm1(_p(x), _p(gamma), _p(beta), _p(res), _p(out), N, C, ctypes.c_long(S), ctypes.c_float(eps), int(relu))
    return out


def softmax_channels(x):
    x = _f(x)
    out = np.empty_like(x)
    lib().oracle_softmax_channels(_p(x), _p(out), x.shape[0], x.shape[1], ctypes.c_long(int(np.prod(x.shape[2:]))))
    return out


def multi_dice(probs, target, weights):
    probs, target, weights = _f(probs), _f(target), _f(weights)
    return float(lib().oracle_multi_dice(_p(probs), _p(target), _p(weights), probs.shape[0], probs.shape[1],
                                         ctypes.c_long(int(np.prod(probs.shape[2:])))))


def focal(probs, target, alpha, gamma, size_average=True):
    probs, target, alpha = _f(probs), _f(target), _f(alpha)
    return float(lib().oracle_focal(_p(probs), _p(target), _p(alpha), probs.shape[0], probs.shape[1],
                                    ctypes.c_long(int(np.prod(probs.shape[2:]))), ctypes.c_float(gamma), int(size_average)))


def partition(image_size, spacing, bbox_start, bbox_end, partition_size, partition_stride, max_stride):
    isz = (ctypes.c_int * 3)(*[int(v) for v in image_size])
    sp = (ctypes.c_double * 3)(*[float(v) for v in spacing])
    bs = (ctypes.c_int * 3)(*[int(v) for v in bbox_start])
    be = (ctypes.c_int * 3)(*[int(v) for v in bbox_end])
    ps = (ctypes.c_double * 3)(*[float(v) for v in partition_size])
    pst = (ctypes.c_double * 3)(*[float(v) for v in partition_stride])
    box = (ctypes.c_int * 3)()
    cap = 4096
    starts = (ctypes.c_int * (3 * cap))()
    n = lib().oracle_partition(isz, sp, bs, be, ps, pst, int(max_stride), starts, cap, box)
    assert n <= cap
    st = [[starts[3 * i + d] for d in range(3)] for i in range(n)]
    return st, [[s[d] + box[d] for d in range(3)] for s in st]
