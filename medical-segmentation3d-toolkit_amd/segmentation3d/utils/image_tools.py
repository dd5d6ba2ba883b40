"""Hot-path helpers of the reference's utils/image_tools.py, restated without SimpleITK.

Only the five functions that define the sliding-window semantics are provided:
  image_partition_by_fixed_size  (image_tools.py:163-218)   pure index arithmetic, host side
  add_image_region / add_image_value (image_tools.py:435-469) -> device kernels, see core/seg_infer.py
  convert_image_to_tensor / convert_tensor_to_image (image_tools.py:274-326)
plus the geometry functions around the patch path, on the device:
  resample / resample_spacing                         (image_tools.py:329-377)  trilinear / NN, ITK inside test and padding
  pick_largest_connected_component / remove_small_connected_component (image_tools.py:380-432)  26-connectivity
  get_bounding_box                                    (image_tools.py:481-510)
Images are `Image3d` (numpy [z, y, x] + spacing / origin / direction); the `*_device` variants take and return device
tensors so that core/seg_infer.segmentation_volume keeps the whole chain on the GPU.
"""
import ctypes

import numpy as np
import torch

from segmentation3d import _engine as E
from segmentation3d.utils.image3d import Image3d, image_size_spacing


def _ceil_to(value, multiple):
    return value if value % multiple == 0 else multiple * (value // multiple + 1)


def image_partition_by_fixed_size(image, bbox_start_voxel, bbox_end_voxel, partition_size, partition_stride,
                                  max_stride):
    """Split the bounding box of an image into fixed-size, overlapping boxes (reference: image_tools.py:163-218).

    :param image: Image3d / sitk-like object (GetSize, GetSpacing) or a (size_xyz, spacing_xyz) pair
    :param bbox_start_voxel, bbox_end_voxel: partition region [start, end) in voxels (x, y, z); updated in place like
           the reference does (rounded up to a multiple of max_stride and clamped into the image)
    :param partition_size, partition_stride: physical size / stride of the boxes (mm)
    :return: (start_voxels, end_voxels), lists of [x, y, z]; order: x outermost, z innermost
    """
    size, spacing = image_size_spacing(image)
    extent, box, step, count = [0] * 3, [0] * 3, [0] * 3, [0] * 3
    for d in range(3):
        assert size[d] >= max_stride and size[d] % max_stride == 0
        # region: rounded up to the stride multiple, never larger than the image, shifted back inside if needed
        extent[d] = min(_ceil_to(min(size[d], bbox_end_voxel[d] - bbox_start_voxel[d]), max_stride), size[d])
        bbox_end_voxel[d] = bbox_start_voxel[d] + extent[d]
        if bbox_end_voxel[d] > size[d]:
            bbox_end_voxel[d] = size[d]
            bbox_start_voxel[d] = size[d] - extent[d]
        assert bbox_start_voxel[d] >= 0
        # box and step in voxels (round half up), box a stride multiple, both at most the region
        box[d] = min(extent[d], _ceil_to(int(partition_size[d] / spacing[d] + 0.5), max_stride))
        step[d] = min(extent[d], int(partition_stride[d] / spacing[d] + 0.5))
        count[d] = int(np.ceil((extent[d] - box[d]) / step[d]) + 1)
    start_voxels, end_voxels = [], []
    for ix in range(count[0]):
        for iy in range(count[1]):
            for iz in range(count[2]):
                lo = [bbox_start_voxel[d] + i * step[d] for d, i in enumerate((ix, iy, iz))]
                for d in range(3):  # the last box of a row is pulled back so that it ends at the region border
                    lo[d] = min(lo[d], bbox_end_voxel[d] - box[d])
                    assert lo[d] >= 0
                start_voxels.append([int(v) for v in lo])
                end_voxels.append([int(lo[d] + box[d]) for d in range(3)])
    return start_voxels, end_voxels


def convert_image_to_tensor(image):
    """Image3d (or list of them) -> float tensor [C, z, y, x] (reference: image_tools.py:274-294)"""
    if isinstance(image, Image3d):
        return torch.from_numpy(np.ascontiguousarray(image.array)).unsqueeze(0).float()
    if isinstance(image, (list, tuple)):
        return torch.cat([convert_image_to_tensor(im) for im in image], 0)
    raise ValueError('unknown input type')


def convert_tensor_to_image(tensor, dtype=None):
    """3-D tensor -> Image3d, 4-D tensor -> list of Image3d (reference: image_tools.py:297-326)"""
    assert isinstance(tensor, torch.Tensor), 'input must be a tensor'
    data = tensor.detach().cpu().numpy()
    if dtype is not None:
        data = data.astype(dtype)
    if tensor.dim() == 3:
        return Image3d(data)
    if tensor.dim() == 4:
        return [Image3d(data[i]) for i in range(data.shape[0])]
    raise ValueError('Only supports 3-dimsional or 4-dimensional image volume')


# ---------------------------------------------------------------------------------------------------------------------
# geometry on the device
# ---------------------------------------------------------------------------------------------------------------------
def _device():
    return torch.device('cuda', torch.cuda.current_device())


def index_affine(src_frame, dst_frame):
    """3 x 4 matrix M with  c_src = M @ (x, y, z, 1)  for an index (x, y, z) of the destination grid: destination
    index -> physical point -> continuous source index, as sitk.Resample does with an identity transform.
    frame = (spacing, origin, direction), direction row-major 3 x 3"""
    s_sp, s_or, s_dir = (np.asarray(v, dtype=np.float64) for v in src_frame)
    d_sp, d_or, d_dir = (np.asarray(v, dtype=np.float64) for v in dst_frame)
    Ds, Dd = s_dir.reshape(3, 3), d_dir.reshape(3, 3)
    to_src = np.diag(1.0 / s_sp) @ np.linalg.inv(Ds)
    M = np.zeros((3, 4))
    M[:, :3] = to_src @ Dd @ np.diag(d_sp)
    M[:, 3] = to_src @ (d_or - s_or)
    return M


def resample_device(src, src_frame, out_size, dst_frame, interp_method, padding_value=0.0):
    """src: float32 device tensor [Z, Y, X]; returns the float32 device tensor [Zo, Yo, Xo] of the destination grid"""
    if interp_method not in ('LINEAR', 'NN'):
        raise ValueError('Unsupported interpolation type.')
    E.require_device(src)
    src = src.contiguous()
    Zi, Yi, Xi = src.shape
    Xo, Yo, Zo = (int(v) for v in out_size)
    dst = torch.empty((Zo, Yo, Xo), dtype=torch.float32, device=src.device)
    M = np.ascontiguousarray(index_affine(src_frame, dst_frame), dtype=np.float64)
    E.call('seg3d_resample_affine', E.ptr(src), E.ptr(dst), Xi, Yi, Zi, Xo, Yo, Zo,
           M.ctypes.data_as(ctypes.c_void_p), int(interp_method == 'LINEAR'), float(padding_value), E.stream_ptr())
    return dst


def _frame(image):
    return (image.GetSpacing(), image.GetOrigin(), image.GetDirection())


def resample(image, reference, interp_method, padding_value=0.0):
    """Resample `image` onto the grid of `reference` (reference: image_tools.py:329-345)"""
    assert isinstance(image, Image3d) and isinstance(reference, Image3d)
    src = torch.from_numpy(np.array(image.array, dtype=np.float32, order='C')).to(_device())
    out = resample_device(src, _frame(image), reference.GetSize(), _frame(reference), interp_method, padding_value)
    return Image3d(out.cpu().numpy(), *_frame(reference))


def resampled_size(in_size, in_spacing, out_spacing, max_stride):
    """output size of resample_spacing (image_tools.py:361-365): round half up, then up to a multiple of max_stride"""
    out = [int(in_size[d] * in_spacing[d] / out_spacing[d] + 0.5) for d in range(3)]
    return [max_stride * (v // max_stride + 1) if v % max_stride else v for v in out]


def resample_spacing(image, resampled_spacing, max_stride, interp_method):
    """Resample to a new spacing, same origin / direction, size a multiple of max_stride (image_tools.py:348-377);
    voxels beyond the input extent take ITK's default pixel value 0"""
    assert isinstance(image, Image3d)
    out_spacing = [float(v) for v in resampled_spacing]
    out_size = resampled_size(image.GetSize(), image.GetSpacing(), out_spacing, max_stride)
    dst_frame = (out_spacing, image.GetOrigin(), image.GetDirection())
    src = torch.from_numpy(np.array(image.array, dtype=np.float32, order='C')).to(_device())
    out = resample_device(src, _frame(image), out_size, dst_frame, interp_method, 0.0)
    return Image3d(out.cpu().numpy(), *dst_frame)


def connected_component_filter_device(mask, labels, mode, threshold=0):
    """mask: int8 device tensor [Z, Y, X]; per label keeps the largest 26-connected component (mode 'largest') or the
    components with at least `threshold` voxels (mode 'min_size'); composition as in the reference: the FIRST label's
    voxels become 1, every other label keeps its own value (image_tools.py:399-403, 429-431)"""
    assert isinstance(labels, list)
    E.require_device(mask)
    if mask.dtype != torch.int8:
        raise TypeError('mask must be int8')
    mask = mask.contiguous()
    Z, Y, X = mask.shape
    out = torch.zeros_like(mask)
    if not labels:
        return out
    ws = torch.empty(E.query('seg3d_ccl_workspace_ints', mask.numel()), dtype=torch.int32, device=mask.device)
    for k, label in enumerate(labels):
        E.call('seg3d_ccl26_select', E.ptr(mask), int(label), X, Y, Z, 0 if mode == 'largest' else 1, int(threshold),
               1 if k == 0 else int(label), int(k > 0), E.ptr(out), E.ptr(ws), E.stream_ptr())
    return out


def _mask_to_device(mask):
    assert isinstance(mask, Image3d)
    return torch.from_numpy(np.array(mask.array, dtype=np.int8, order='C')).to(_device())


def pick_largest_connected_component(mask, labels):
    """keep, for every label, only its largest 26-connected component (image_tools.py:380-405)"""
    out = connected_component_filter_device(_mask_to_device(mask), labels, 'largest')
    return Image3d(out.cpu().numpy().astype(mask.array.dtype), *_frame(mask))


def remove_small_connected_component(mask, labels, threshold):
    """drop, for every label, the 26-connected components smaller than `threshold` voxels (image_tools.py:408-432)"""
    out = connected_component_filter_device(_mask_to_device(mask), labels, 'min_size', threshold)
    return Image3d(out.cpu().numpy().astype(mask.array.dtype), *_frame(mask))


def get_bounding_box_device(mask, selected_labels):
    """mask: int8 device tensor [Z, Y, X] -> (start_voxel, end_voxel) in (x, y, z), end exclusive, or (None, None)"""
    E.require_device(mask)
    mask = mask.contiguous()
    Z, Y, X = mask.shape
    box = torch.tensor([2 ** 31 - 1] * 3 + [-1] * 3, dtype=torch.int32, device=mask.device)
    labels = [] if selected_labels is None else [int(v) for v in selected_labels]
    if selected_labels is not None and not labels:
        return None, None
    arr = (ctypes.c_int * max(1, len(labels)))(*labels) if labels else None
    E.call('seg3d_mask_bounding_box', E.ptr(mask), X, Y, Z, arr, len(labels), E.ptr(box), E.stream_ptr())
    b = box.cpu().tolist()
    if b[3] < 0:
        print('Fail to get the bounding box.')
        return None, None
    return [b[0], b[1], b[2]], [b[3] + 1, b[4] + 1, b[5] + 1]


def get_bounding_box(mask, selected_labels):
    """bounding box of the selected labels (None: every non-zero voxel), end exclusive (image_tools.py:481-510)"""
    return get_bounding_box_device(_mask_to_device(mask), selected_labels)


# ---------------------------------------------------------------------------------------------------------------------
# training crops on the device: crop_image (image_tools.py:103-141) + the crop normalisers
# ---------------------------------------------------------------------------------------------------------------------
def crop_origin(cropping_center, cropping_size, cropping_spacing):
    """world position of the crop's first voxel (image_tools.py:121-126): centre - size/2, moved in by half a voxel;
    like the reference this ignores the image direction"""
    out = []
    for idx in range(3):
        physical = int(cropping_size[idx]) * float(cropping_spacing[idx])
        out.append(float(cropping_center[idx]) - physical / 2.0 + float(cropping_spacing[idx]) / 2.0)
    return out


def crop_image_device(volume, frame, cropping_center, cropping_size, cropping_spacing, interp_method):
    """volume: float32 device tensor [Z, Y, X] with frame (spacing, origin, direction) -> crop [z, y, x] of
    `cropping_size` voxels at `cropping_spacing`, centred at the world point `cropping_center`, zero outside"""
    size = [int(cropping_size[idx]) for idx in range(3)]
    spacing = [float(cropping_spacing[idx]) for idx in range(3)]
    dst_frame = (spacing, crop_origin(cropping_center, size, spacing), frame[2])
    return resample_device(volume, frame, size, dst_frame, interp_method, 0.0)


def crop_image(image, cropping_center, cropping_size, cropping_spacing, interp_method):
    """Image3d in, Image3d out (reference: image_tools.py:103-141)"""
    assert isinstance(image, Image3d)
    src = torch.from_numpy(np.array(image.array, dtype=np.float32, order='C')).to(_device())
    out = crop_image_device(src, _frame(image), cropping_center, cropping_size, cropping_spacing, interp_method)
    spacing = [float(cropping_spacing[idx]) for idx in range(3)]
    return Image3d(out.cpu().numpy(), spacing, crop_origin(cropping_center, cropping_size, spacing), image.GetDirection())


def normalize_crop_device(crop, normalizer):
    """apply a FixedNormalizer / AdaptiveNormalizer (utils/normalizer.py) to a float32 device crop [z, y, x] with the
    patch kernel of the inference path (csrc/patch.hip: fp64 statistics, population std floored at 1e-6)"""
    d = normalizer.to_dict() if hasattr(normalizer, 'to_dict') else dict(normalizer)
    crop = crop.contiguous()
    bz, by, bx = crop.shape
    if d['type'] == 0:
        ntype, mean, std, clip, sigma = 0, float(d['mean']), float(d['stddev']), int(bool(d['clip'])), 1.0
    elif d['type'] == 1:
        ntype, mean, std, clip, sigma = 1, 0.0, 1.0, 1, float(d['clip_sigma'])
    else:
        raise ValueError('Unsupported normalization type.')
    dev = crop.device
    out = torch.empty((1, 1, bz, by, bx), dtype=torch.float32, device=dev)
    starts = torch.zeros((1, 3), dtype=torch.int32, device=dev)
    ws = torch.empty((E.query('seg3d_patch_stats_blocks', bx, by, bz) * 2,), dtype=torch.float64, device=dev)
    mean_std = torch.empty((1, 2), dtype=torch.float32, device=dev)
    E.call('seg3d_patch_gather_normalize', E.ptr(crop), E.ptr(starts), E.ptr(out), E.ptr(ws), E.ptr(mean_std), bz, by, bx,
           bx, by, bz, 1, ntype, mean, std, clip, sigma, E.stream_ptr())
    return out[0, 0]


def get_image_frame(image):
    """spacing, origin, direction packed into 15 float32 (image_tools.py:26-41)"""
    return np.array(list(image.GetSpacing()) + list(image.GetOrigin()) + list(image.GetDirection()), dtype=np.float32)


def set_image_frame(image, frame):
    """inverse of get_image_frame (image_tools.py:44-61)"""
    frame = np.asarray(frame)
    image.spacing = tuple(float(v) for v in frame[:3])
    image.origin = tuple(float(v) for v in frame[3:6])
    image.direction = tuple(float(v) for v in frame[6:15])


def select_random_voxels_in_multi_class_mask(mask, num_selected, selected_label):
    """`num_selected` random voxels (x, y, z) carrying `selected_label` (image_tools.py:246-271; numpy global RNG)"""
    mask_npy = mask.array if isinstance(mask, Image3d) else np.asarray(mask)
    valid_voxels = np.argwhere(mask_npy == selected_label)
    selected_voxels = []
    while len(valid_voxels) > 0 and len(selected_voxels) < num_selected:
        selected_voxels.append(valid_voxels[np.random.randint(0, len(valid_voxels))][::-1])
    return selected_voxels
