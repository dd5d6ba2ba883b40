"""ORACLE helper -- deterministic, RNG-library-independent tensor generator.

Golden fixtures hold only outputs; inputs and weights are regenerated from (seed, name) on both sides (the build
container that imports the reference, and the GPU box that does not have it) with this counter-based generator, so
58 MB of weights never need to be committed.  splitmix64 over the element index -> two uniforms -> Box-Muller normal.
"""
import hashlib

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    with np.errstate(over='ignore'):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def _key(seed, name):
    h = hashlib.sha256('{}:{}'.format(seed, name).encode()).digest()
    return np.uint64(int.from_bytes(h[:8], 'little'))


def uniform(seed, name, shape):
    """float64 uniforms in (0, 1)"""
    n = int(np.prod(shape))
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over='ignore'):
        bits = _splitmix64(idx * np.uint64(2) + _key(seed, name))
    u = ((bits >> np.uint64(11)).astype(np.float64) + 0.5) / float(1 << 53)
    return u.reshape(shape)


def normal(seed, name, shape, std=1.0, mean=0.0):
    """float32 normal(mean, std)"""
    u1 = uniform(seed, name + '/u1', shape)
    u2 = uniform(seed, name + '/u2', shape)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return (mean + std * z).astype(np.float32)


def labels(seed, name, shape, num_classes):
    """float32 class ids in {0..C-1} with blocky structure (4x4x4 cells) so classes form blobs, ~uniform classes"""
    shape = tuple(shape)
    cells = tuple((s + 3) // 4 for s in shape[-3:])
    u = uniform(seed, name, shape[:-3] + cells)
    ids = np.minimum((u * num_classes).astype(np.int64), num_classes - 1)
    for ax in (-3, -2, -1):
        ids = np.repeat(ids, 4, axis=ax)
    sl = tuple(slice(None) for _ in shape[:-3]) + tuple(slice(0, s) for s in shape[-3:])
    return ids[sl].astype(np.float32)


def state_dict_like(shapes, seed, gain_sqrt2=True):
    """deterministic weights for a dict name -> shape: conv weights ~ kaiming-normal(fan_in), biases ~ N(0, 0.1),
    GroupNorm weight ~ N(1, 0.1), GroupNorm bias ~ N(0, 0.1) (non-trivial affine so parity tests exercise it)"""
    out = {}
    for name, shape in shapes.items():
        shape = tuple(shape)
        if name.endswith('.weight') and len(shape) == 5:
            fan_in = shape[1] * int(np.prod(shape[2:]))
            std = (2.0 ** 0.5 if gain_sqrt2 else 1.0) / np.sqrt(fan_in)
            out[name] = normal(seed, name, shape, std=std)
        elif name.endswith('.weight'):
            out[name] = normal(seed, name, shape, std=0.1, mean=1.0)
        else:
            out[name] = normal(seed, name, shape, std=0.1)
    return out


def metric_label_cases():
    """(name, gt, seg, labels, threshold): label volumes from the deterministic generator (shared with the tests)"""
    cases = []
    g = labels(301, 'metric/gt', (40, 48, 56), 4).astype(np.int8)
    s_ = labels(302, 'metric/seg', (40, 48, 56), 4).astype(np.int8)
    mix = np.where(uniform(303, 'metric/mix', (40, 48, 56)) < 0.7, g, s_).astype(np.int8)   # 70 % agreement
    cases.append(('agree70_int8', g, mix, [0, 1, 2, 3, 7], 1000))
    cases.append(('threshold_edges', g, mix, [1, 2], int((g == 1).sum())))          # area_gt == threshold -> TP side
    cases.append(('threshold_above', g, mix, [1, 2], int((g == 1).sum()) + 1))      # area_gt < threshold
    only_seg = np.where(g == 3, 0, g).astype(np.int8)
    cases.append(('label_missing_in_gt', only_seg, mix, [3], 50))                    # FP
    cases.append(('label_missing_in_seg', mix, only_seg, [3], 50))                   # FN
    cases.append(('float_labels', g.astype(np.float32), mix.astype(np.float32), [0, 2], 10))
    cases.append(('int16_odd_size', g[:37, :41, :53].astype(np.int16), mix[:37, :41, :53].astype(np.int16), [0, 1, 2, 3], 10))
    return cases


def sampling_cases():
    """(name, size xyz, spacing, origin, direction, crop_size, crop_spacing, seed)"""
    eye = [1.0, 0, 0, 0, 1.0, 0, 0, 0, 1.0]
    flip = [-1.0, 0, 0, 0, 1.0, 0, 0, 0, 1.0]
    return [
        ('big_iso', (200, 180, 160), (1.0, 1.0, 1.0), (0.0, 0.0, 0.0), eye, (96, 96, 96), (1.0, 1.0, 1.0), 11),
        ('aniso_offset', (120, 90, 60), (0.7, 0.9, 2.5), (-31.5, 12.0, 100.0), eye, (64, 48, 32), (1.2, 1.2, 1.2), 12),
        ('smaller_than_crop', (60, 200, 40), (1.0, 1.0, 1.0), (5.0, 5.0, 5.0), eye, (96, 96, 96), (1.0, 1.0, 1.0), 13),
        ('flipped_x', (100, 100, 100), (0.8, 0.8, 0.8), (40.0, -40.0, 0.0), flip, (48, 48, 48), (1.0, 1.0, 1.0), 14),
    ]


def normalizer_cases():
    """(name, roi float32 [z, y, x], 'fixed' | 'adaptive', params) shared by oracle/gen_golden.py and the tests"""
    ct = normal(401, 'norm/ct', (12, 20, 28), std=350.0, mean=-200.0)          # CT-like intensities
    mr = (uniform(402, 'norm/mr', (16, 16, 24)) ** 3 * 1800.0).astype(np.float32)  # skewed MR-like intensities
    flat = np.full((8, 8, 8), 7.25, dtype=np.float32)                           # zero variance -> std floor 1e-6
    return [
        ('fixed_ct_clip', ct, 'fixed', {'mean': -150.0, 'stddev': 350.0, 'clip': True}),
        ('fixed_ct_noclip', ct, 'fixed', {'mean': 40.0, 'stddev': 120.5, 'clip': False}),
        ('fixed_mr_clip', mr, 'fixed', {'mean': 300.0, 'stddev': 250.0, 'clip': True}),
        ('adaptive_ct_3', ct, 'adaptive', {'clip_sigma': 3}),
        ('adaptive_mr_2', mr, 'adaptive', {'clip_sigma': 2}),
        ('adaptive_mr_0p5', mr, 'adaptive', {'clip_sigma': 0.5}),
        ('adaptive_flat', flat, 'adaptive', {'clip_sigma': 3}),
    ]


def accumulate_case():
    """(volume shape [z, y, x], [(start xyz, end xyz), ...]): 12 x 10 x 8 boxes at stride 6 x 5 x 4 over a 24 x 20 x 16 volume
    (overlap counts 1..8), in the reference's x-outer / z-inner order"""
    vol = (16, 20, 24)
    box, stride = (12, 10, 8), (6, 5, 4)
    patches = []
    for x in range(0, 24 - box[0] + 1, stride[0]):
        for y in range(0, 20 - box[1] + 1, stride[1]):
            for z in range(0, 16 - box[2] + 1, stride[2]):
                patches.append(([x, y, z], [x + box[0], y + box[1], z + box[2]]))
    return vol, patches
