"""Hot-path helpers of the reference's utils/image_tools.py, restated without SimpleITK.

Only the five functions that define the sliding-window semantics are provided:
  image_partition_by_fixed_size  (image_tools.py:163-218)   pure index arithmetic, host side
  add_image_region / add_image_value (image_tools.py:435-469) -> device kernels, see core/seg_infer.py
  convert_image_to_tensor / convert_tensor_to_image (image_tools.py:274-326)
The ITK geometry functions (resample, crop, connected components, bounding box) are outside this round's scope.
"""
import numpy as np
import torch

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
