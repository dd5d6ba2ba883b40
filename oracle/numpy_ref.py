"""ORACLE (test infrastructure, not product code) -- numpy restatement of the reference's sliding-window host path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Restated from (all relative to /root/reference/segmentation3d):
  utils/image_tools.py:163-218  image_partition_by_fixed_size   -> partition_by_fixed_size
  utils/normalizer.py:6-81 + utils/image_tools.py:221-238,472-478 -> fixed_normalize / adaptive_normalize
  utils/image_tools.py:435-469  add_image_region / add_image_value -> accumulate_patch
  core/seg_infer.py:313-327,336-339  accumulate loop, 1/count, argmax -> sliding_window_inference
These reference modules import SimpleITK (absent here), so they cannot be imported; their function bodies were executed
in place instead (extracted with `ast`, see oracle/gen_golden.py) on duck-typed images: the partition function ->
tests/golden/partitions.json, the normalisers (both classes + normalize_image + get_mean_std_from_image) and the
accumulate helpers (add_image_region / add_image_value) -> tests/golden/normalizers.npz.  tests/test_oracle_golden.py
checks this restatement against those fixtures bit for bit.

Array convention: a volume with sitk size (X, Y, Z) is a numpy array [Z, Y, X] (image_tools.py:448,465).
"""
import math

import numpy as np


def partition_by_fixed_size(image_size, image_spacing, bbox_start_voxel, bbox_end_voxel, partition_size,
                            partition_stride, max_stride):
    """image_tools.py:163-218.  All triples are (x, y, z).  Returns (start_voxels, end_voxels) lists of [x, y, z]."""
    image_size = [int(s) for s in image_size]
    for d in range(3):
        assert image_size[d] >= max_stride and image_size[d] % max_stride == 0            # :176-177
    start = [int(v) for v in bbox_start_voxel]
    end = [int(v) for v in bbox_end_voxel]
    bbox_size = [min(image_size[d], end[d] - start[d]) for d in range(3)]                # :179
    for d in range(3):
        if bbox_size[d] % max_stride != 0:                                                 # :181-182
            bbox_size[d] = max_stride * (bbox_size[d] // max_stride + 1)
        bbox_size[d] = min(bbox_size[d], image_size[d])                                    # :183
        end[d] = start[d] + bbox_size[d]                                                   # :184
        if end[d] > image_size[d]:                                                         # :185-187
            end[d] = image_size[d]
            start[d] = end[d] - bbox_size[d]
        assert start[d] >= 0
    box = [int(partition_size[d] / image_spacing[d] + 0.5) for d in range(3)]             # :190
    for d in range(3):
        if box[d] % max_stride:                                                            # :192-193
            box[d] = max_stride * (box[d] // max_stride + 1)
        box[d] = min(bbox_size[d], box[d])                                                 # :194
    stride = [int(partition_stride[d] / image_spacing[d] + 0.5) for d in range(3)]        # :196
    for d in range(3):
        stride[d] = min(bbox_size[d], stride[d])                                           # :198
    count = [int(math.ceil((bbox_size[d] - box[d]) / stride[d]) + 1) for d in range(3)]   # :200
    starts, ends = [], []
    for ix in range(count[0]):                                                             # x outer, z inner :202-204
        for iy in range(count[1]):
            for iz in range(count[2]):
                s = [start[0] + ix * stride[0], start[1] + iy * stride[1], start[2] + iz * stride[2]]
                e = [s[d] + box[d] for d in range(3)]
                for d in range(3):
                    if e[d] > end[d]:                                                      # clamp the tail :209-213
                        e[d] = end[d]
                        s[d] = e[d] - box[d]
                        assert s[d] >= 0
                starts.append(s)
                ends.append(e)
    return starts, ends


def fixed_normalize(roi, mean, stddev, clip=True):
    """FixedNormalizer.__call__ -> normalize_image(image, mean, std, clip) -- normalizer.py:22-25, image_tools.py:221-238"""
    out = (roi - mean) / stddev
    if clip:
        out[out < -1.0] = -1.0
        out[out > 1.0] = 1.0
    return out.astype(roi.dtype)


def adaptive_normalize(roi, clip_sigma=3):
    """AdaptiveNormalizer.normalize -- normalizer.py:55-62 (population std, floor 1e-6, clip +-clip_sigma)"""
    mean, std = np.mean(roi), np.std(roi)                                                  # image_tools.py:472-478
    std = max(std, 1e-6)
    out = (roi - mean) / std
    out[out < -clip_sigma] = -clip_sigma
    out[out > clip_sigma] = clip_sigma
    return out.astype(roi.dtype)


def apply_normalizer(roi, normalizer):
    """normalizer: dict as stored in checkpoints (normalizer.py:36-39,78-81): {'type': 0, mean, stddev, clip} | {'type': 1, clip_sigma}"""
    if normalizer is None:
        return roi
    if normalizer['type'] == 0:
        return fixed_normalize(roi, normalizer['mean'], normalizer['stddev'], normalizer['clip'])
    if normalizer['type'] == 1:
        return adaptive_normalize(roi, normalizer['clip_sigma'])
    raise ValueError('Unsupported normalization type.')                                    # seg_infer.py:162


def accumulate_patch(acc, count, start, end, patch_probs):
    """add_image_region per class + add_image_value(+1) -- image_tools.py:435-469, seg_infer.py:319-322.
    acc: [C, Z, Y, X], count: [Z, Y, X], patch_probs: [C, bz, by, bx], start/end: (x, y, z)"""
    zs, ys, xs = slice(start[2], end[2]), slice(start[1], end[1]), slice(start[0], end[0])
    for c in range(acc.shape[0]):
        acc[c, zs, ys, xs] += patch_probs[c]
    count[zs, ys, xs] += 1.0


def finalize(acc, count):
    """probs *= 1/count; mask = argmax over classes (first maximum), int8 -- seg_infer.py:325-327,336-339.
    Voxels no patch covered (count 0; bounding-box runs of the coarse -> fine cascade): the reference computes
    `1.0 / count` on SimpleITK images, i.e. ITK's Div functor, which returns NumericTraits::max() for a zero denominator
    instead of inf, so the product with the zero accumulator is 0 (NOT NaN) and the arg-max there is class 0.
    Restated as inv = 0 where count == 0 (same products; ITK is absent here, so this rule is unpinned)."""
    with np.errstate(divide='ignore', invalid='ignore'):
        inv = np.where(count > 0, 1.0 / count, 0.0).astype(np.float32)
        probs = acc * inv[None]
    mask = np.argmax(probs, axis=0).astype(np.int8)
    return probs, mask


def sliding_window_inference(volume, net_fn, num_classes, spacing, partition_size, partition_stride, max_stride,
                             normalizer, double_forward=True, bbox=None, partition_type='SIZE'):
    """segmentation_volume's patch loop at model spacing (no resample): seg_infer.py:276-339.
    volume: float32 [Z, Y, X]; net_fn: array [1,1,bz,by,bx] -> array [1,C,bz,by,bx].
    double_forward reproduces the reference's two identical forwards + mean (seg_infer.py:230-234).
    bbox: (start xyz, end xyz) on this grid restricting the partition (seg_infer.py:292-303), None = whole volume;
    partition_type 'DISABLE' = one box covering the volume (seg_infer.py:277-279)."""
    Z, Y, X = volume.shape
    if partition_type == 'DISABLE':
        starts, ends = [[0, 0, 0]], [[X, Y, Z]]
    elif partition_type == 'SIZE':
        s0, e0 = ([0, 0, 0], [X, Y, Z]) if bbox is None else (list(bbox[0]), list(bbox[1]))
        starts, ends = partition_by_fixed_size((X, Y, Z), spacing, s0, e0, partition_size, partition_stride, max_stride)
    else:
        raise ValueError('Unsupported partition type!')                                    # seg_infer.py:311
    acc = np.zeros((num_classes, Z, Y, X), dtype=np.float32)
    count = np.zeros((Z, Y, X), dtype=np.float32)
    for s, e in zip(starts, ends):
        roi = volume[s[2]:e[2], s[1]:e[1], s[0]:e[0]].copy()
        roi = apply_normalizer(roi, normalizer)
        p = net_fn(roi[None, None])
        if double_forward:
            p = np.mean(np.stack([p, net_fn(roi[None, None])]), axis=0)
        accumulate_patch(acc, count, s, e, p[0])
    probs, mask = finalize(acc, count)
    return probs, mask, (starts, ends)


# ---------------------------------------------------------------------------------------------------------------------
# evaluation metric
# ---------------------------------------------------------------------------------------------------------------------
def cal_dsc(gt, seg, label, threshold):
    """restatement of utils/metrics.py:5-37: (dice, type) with the TN / FP / FN / TP typing by voxel-count threshold"""
    g, s = (np.asarray(gt) == label), (np.asarray(seg) == label)
    area_gt, area_seg = int(g.sum()), int(s.sum())
    if area_gt < threshold and area_seg < threshold:
        return 1.0, 'TN'
    if area_gt < threshold and area_seg >= threshold:
        return 0.0, 'FP'
    if area_gt >= threshold and area_seg < threshold:
        return 0.0, 'FN'
    return 2 * int((g & s).sum()) / (area_gt + area_seg), 'TP'


def label_overlap_counts(gt, seg, labels):
    g, s = np.asarray(gt), np.asarray(seg)
    return [(int((g == l).sum()), int((s == l).sum()), int(((g == l) & (s == l)).sum())) for l in labels]


# ---------------------------------------------------------------------------------------------------------------------
# geometry around the patch path -- ITK semantics restated, no SimpleITK here: parity unpinned
# ---------------------------------------------------------------------------------------------------------------------
def index_affine(src_frame, dst_frame):
    """3 x 4 map destination index -> continuous source index (identity transform between the physical spaces)"""
    s_sp, s_or, s_dir = (np.asarray(v, dtype=np.float64) for v in src_frame)
    d_sp, d_or, d_dir = (np.asarray(v, dtype=np.float64) for v in dst_frame)
    to_src = np.diag(1.0 / s_sp) @ np.linalg.inv(s_dir.reshape(3, 3))
    M = np.zeros((3, 4))
    M[:, :3] = to_src @ d_dir.reshape(3, 3) @ np.diag(d_sp)
    M[:, 3] = to_src @ (d_or - s_or)
    return M


def resample_affine(src, M, out_size, linear=True, pad=0.0):
    """src [Z, Y, X] sampled on the (Xo, Yo, Zo) grid; restates sitk.Resample(identity, LINEAR / NN, pad)
    (utils/image_tools.py:329-377): inside iff -0.5 <= c < size - 0.5, clamped neighbourhood, double arithmetic"""
    src = np.asarray(src, dtype=np.float64)
    Zi, Yi, Xi = src.shape
    Xo, Yo, Zo = out_size
    z, y, x = np.meshgrid(np.arange(Zo), np.arange(Yo), np.arange(Xo), indexing='ij')
    c = [M[r, 0] * x + M[r, 1] * y + M[r, 2] * z + M[r, 3] for r in range(3)]
    size = (Xi, Yi, Zi)
    inside = np.ones(x.shape, dtype=bool)
    for r in range(3):
        inside &= (c[r] >= -0.5) & (c[r] < size[r] - 0.5)
    if linear:
        f = [np.clip(c[r], 0.0, size[r] - 1) for r in range(3)]
        i0 = [np.floor(v).astype(np.int64) for v in f]
        i1 = [np.minimum(i0[r] + 1, size[r] - 1) for r in range(3)]
        d = [f[r] - i0[r] for r in range(3)]
        g = lambda zz, yy, xx: src[zz, yy, xx]
        a00 = g(i0[2], i0[1], i0[0]) + (g(i0[2], i0[1], i1[0]) - g(i0[2], i0[1], i0[0])) * d[0]
        a01 = g(i0[2], i1[1], i0[0]) + (g(i0[2], i1[1], i1[0]) - g(i0[2], i1[1], i0[0])) * d[0]
        a10 = g(i1[2], i0[1], i0[0]) + (g(i1[2], i0[1], i1[0]) - g(i1[2], i0[1], i0[0])) * d[0]
        a11 = g(i1[2], i1[1], i0[0]) + (g(i1[2], i1[1], i1[0]) - g(i1[2], i1[1], i0[0])) * d[0]
        b0 = a00 + (a01 - a00) * d[1]
        b1 = a10 + (a11 - a10) * d[1]
        val = b0 + (b1 - b0) * d[2]
    else:
        n = [np.clip(np.floor(c[r] + 0.5).astype(np.int64), 0, size[r] - 1) for r in range(3)]
        val = src[n[2], n[1], n[0]]
    return np.where(inside, val, pad).astype(np.float32)


def resampled_size(in_size, in_spacing, out_spacing, max_stride):
    out = [int(in_size[d] * in_spacing[d] / out_spacing[d] + 0.5) for d in range(3)]
    return [max_stride * (v // max_stride + 1) if v % max_stride else v for v in out]


def connected_component_filter(mask, labels, mode, threshold=0):
    """restates pick_largest_connected_component / remove_small_connected_component (utils/image_tools.py:380-432):
    26-connectivity; largest = most voxels, ties to the component met first in raster order; composition: first label
    -> 1, the others keep their value"""
    from scipy import ndimage
    mask = np.asarray(mask)
    out = np.zeros(mask.shape, dtype=np.int64)
    for k, label in enumerate(labels):
        cc, n = ndimage.label(mask == label, structure=np.ones((3, 3, 3), dtype=bool))   # labels in raster order
        keep = np.zeros(mask.shape, dtype=bool)
        if n > 0:
            sizes = np.bincount(cc.ravel(), minlength=n + 1)[1:]
            if mode == 'largest':
                keep = cc == (int(np.argmax(sizes)) + 1)          # argmax returns the first maximum
            else:
                keep = np.isin(cc, np.nonzero(sizes >= threshold)[0] + 1)
        out += (1 if k == 0 else int(label)) * keep
    return out.astype(mask.dtype)


def get_bounding_box(mask, selected_labels):
    """utils/image_tools.py:481-510: (start, end) in (x, y, z), end exclusive; (None, None) when nothing is selected"""
    mask = np.asarray(mask)
    sel = (mask > 0) if selected_labels is None else np.isin(mask, list(selected_labels))
    if not sel.any():
        return None, None
    zz, yy, xx = np.nonzero(sel)
    return [int(xx.min()), int(yy.min()), int(zz.min())], [int(xx.max()) + 1, int(yy.max()) + 1, int(zz.max()) + 1]


def physical_to_index(frame, point):
    """sitk TransformPhysicalPointToIndex: nearest index (round half up per axis)"""
    spacing, origin, direction = (np.asarray(v, dtype=np.float64) for v in frame)
    c = np.diag(1.0 / spacing) @ np.linalg.inv(direction.reshape(3, 3)) @ (np.asarray(point, dtype=np.float64) - origin)
    return [int(np.floor(v + 0.5)) for v in c]


def index_to_physical(frame, index):
    """sitk TransformContinuousIndexToPhysicalPoint"""
    spacing, origin, direction = (np.asarray(v, dtype=np.float64) for v in frame)
    return origin + direction.reshape(3, 3) @ (spacing * np.asarray(index, dtype=np.float64))


def segmentation_volume(image, frame, net_fn, num_classes, model_spacing, partition_size, partition_stride, max_stride,
                        normalizer, interpolation='LINEAR', pick_largest_cc=False, remove_small_cc=0, bbox=None,
                        partition_type='SIZE', double_forward=False):
    """the whole of core/seg_infer.py:249-350 on the host: resample to the model spacing -> patch loop -> resample the
    class probabilities back (padding 1.0 for class 0, 0.0 otherwise) -> arg-max -> component post-processing.
    image [Z, Y, X], frame = (spacing, origin, direction) of the image; bbox = (start xyz, end xyz) in IMAGE voxels
    (converted to the model grid like seg_infer.py:292-303); returns (probs [C, Z, Y, X], mask int8)"""
    Z, Y, X = image.shape
    iso_frame = (list(model_spacing), frame[1], frame[2])
    size = resampled_size((X, Y, Z), frame[0], model_spacing, max_stride)
    iso = resample_affine(image, index_affine(frame, iso_frame), size, interpolation == 'LINEAR', 0.0)
    iso_bbox = None
    if bbox is not None and bbox[0] is not None and bbox[1] is not None:
        s0 = physical_to_index(iso_frame, index_to_physical(frame, [float(v) for v in bbox[0]]))
        e0 = physical_to_index(iso_frame, index_to_physical(frame, [float(v) for v in bbox[1]]))
        iso_bbox = ([max(0, v) for v in s0], [min(v, lim) for v, lim in zip(e0, size)])
    probs, _, _ = sliding_window_inference(iso, net_fn, num_classes, model_spacing, partition_size, partition_stride,
                                           max_stride, normalizer, double_forward=double_forward, bbox=iso_bbox,
                                           partition_type=partition_type)
    back = index_affine(iso_frame, frame)
    out = np.stack([resample_affine(probs[c], back, (X, Y, Z), True, 1.0 if c == 0 else 0.0) for c in range(num_classes)])
    mask = np.argmax(out, axis=0).astype(np.int8)
    labels = list(range(1, num_classes))
    if pick_largest_cc and labels:
        mask = connected_component_filter(mask, labels, 'largest')
    if remove_small_cc > 0 and labels:
        mask = connected_component_filter(mask, labels, 'min_size', remove_small_cc)
    return out, mask


def segmentation_cascade(image, frame, coarse, fine):
    """core/seg_infer.py:428-444 (single_scale == 'DISABLE'): the coarse model on the whole image, the bounding box of its
    mask (all labels), then the fine model restricted to that box.  coarse / fine: dicts of segmentation_volume keyword
    arguments (net_fn, num_classes, model_spacing, partition_*, normalizer, ...).
    Returns (fine probs, fine mask, (bbox start, bbox end))"""
    _, cmask = segmentation_volume(image, frame, **coarse)
    start, end = get_bounding_box(cmask, None)
    if start is None:                 # (an empty coarse mask: the whole image; the reference would fail on None here)
        start, end = [0, 0, 0], list(image.shape[::-1])
    probs, mask = segmentation_volume(image, frame, bbox=(start, end), **fine)
    return probs, mask, (start, end)


# ---------------------------------------------------------------------------------------------------------------------
# training crops: dataloader/dataset.py:110-200, utils/image_tools.py:103-141, 246-271
# ---------------------------------------------------------------------------------------------------------------------
def crop_origin(center, size, spacing):
    return [float(center[d]) - int(size[d]) * float(spacing[d]) / 2.0 + float(spacing[d]) / 2.0 for d in range(3)]


def crop_image(volume, frame, center, size, spacing, linear):
    """restates crop_image: sitk.Resample(image, size, identity, interp, origin, spacing, image direction), pad 0"""
    dst_frame = ([float(v) for v in spacing], crop_origin(center, size, spacing), frame[2])
    return resample_affine(volume, index_affine(frame, dst_frame), [int(v) for v in size], linear, 0.0)


def global_sample(size, spacing, origin, crop_size, crop_spacing, rng):
    """dataset.py:110-127; rng: np.random-like (uniform)"""
    im_mm = [size[d] * spacing[d] for d in range(3)]
    crop_mm = np.asarray(crop_size, dtype=np.double) * np.asarray(crop_spacing, dtype=np.double)
    sp = np.array(origin, dtype=np.double)
    for d in range(3):
        if im_mm[d] > crop_mm[d]:
            sp[d] = origin[d] + rng.uniform(0, im_mm[d] - crop_mm[d])
    return sp + crop_mm / 2


def center_sample(size, spacing, origin, direction):
    """dataset.py:129-142"""
    end_world = np.asarray(origin, dtype=np.double) + np.asarray(direction, dtype=np.double).reshape(3, 3) @ (
        np.asarray(spacing, dtype=np.double) * (np.asarray(size, dtype=np.double) - 1))
    return (np.asarray(origin, dtype=np.double) + end_world) / 2.0


def select_random_voxel(mask, label, rng):
    """image_tools.py:246-271 with num_selected = 1: (x, y, z) or None; consumes one randint only when the label exists"""
    valid = np.argwhere(np.asarray(mask) == label)
    if len(valid) == 0:
        return None
    return [int(v) for v in valid[rng.randint(0, len(valid))][::-1]]
