"""Crop intensity normalisers with the reference's names, parameters and checkpoint dictionaries
(utils/normalizer.py:6-81).  At inference they are applied on the device by the patch batcher
(seg3d_patch_gather_normalize); the numpy `__call__` here is host glue for the training data path and for tests.
"""
import numpy as np

from segmentation3d.utils.image3d import Image3d


def _apply(image, fn):
    if isinstance(image, Image3d):
        return image.like(fn(image.array))
    if isinstance(image, np.ndarray):
        return fn(image)
    if isinstance(image, (list, tuple)):
        for idx, im in enumerate(image):
            image[idx] = _apply(im, fn)
        return image
    raise ValueError('Unknown type of input. Normalizer only supports Image3d or Image3d list/tuple')


class FixedNormalizer(object):
    """intensity = (intensity - mean) / stddev, clipped to [-1, 1] when clip is enabled (type 0)"""

    def __init__(self, mean, stddev, clip=True):
        assert stddev > 0, 'stddev must be positive'
        assert isinstance(clip, bool), 'clip must be a boolean'
        self.mean, self.stddev, self.clip = mean, stddev, clip

    def __call__(self, image):
        def fn(a):
            out = ((a - self.mean) / self.stddev).astype(a.dtype)
            return np.clip(out, -1.0, 1.0) if self.clip else out
        return _apply(image, fn)

    def to_dict(self):
        return {'type': 0, 'mean': self.mean, 'stddev': self.stddev, 'clip': self.clip}


class AdaptiveNormalizer(object):
    """z-score with the crop's own mean / population std (floored at 1e-6), clipped to +-clip_sigma (type 1)"""

    def __init__(self, clip_sigma=3):
        assert clip_sigma > 0
        self.clip_sigma = clip_sigma

    def __call__(self, image):
        def fn(a):
            mean, std = np.mean(a), max(np.std(a), 1e-6)
            return np.clip(((a - mean) / std).astype(a.dtype), -self.clip_sigma, self.clip_sigma)
        return _apply(image, fn)

    def to_dict(self):
        return {'type': 1, 'clip_sigma': self.clip_sigma}


def normalizer_from_dict(d):
    """inverse of to_dict(); raises like core/seg_infer.py:151-162"""
    if d['type'] == 0:
        return FixedNormalizer(d['mean'], d['stddev'], d['clip'])
    if d['type'] == 1:
        return AdaptiveNormalizer(d['clip_sigma'])
    raise ValueError('Unsupported normalization type.')
