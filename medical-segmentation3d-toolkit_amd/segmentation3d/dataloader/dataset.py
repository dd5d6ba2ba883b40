"""Training data path: `SegmentationDataset` with the reference's constructor, sampling
methods and RNG call order (dataloader/dataset.py:55-209), but the crop itself runs on the GPU.

The reference reads both volumes from disk, resamples a crop with SimpleITK and normalises it in a DataLoader worker
for every sample; at >150 patches/s per GPU four such workers cannot feed eight GPUs.  Here every case is read ONCE
(MetaImage, utils/mha_io.py), kept resident in HBM, and a sample costs two launches of the resampling kernel
(csrc/postproc.hip: image LINEAR / NN, mask NN) plus the on-device normaliser (csrc/patch.hip) -- no worker
processes, no host copies.  The random decisions (crop centre, translation, scale) stay on the host with numpy's
global RNG in exactly the reference's order, so a seeded run draws the same crops.
`DeviceCropLoader` batches the samples of a sampler into device tensors for core/seg_train.train().
"""
import os

import numpy as np
import pandas as pd
import torch
from torch.utils.data import Dataset

from segmentation3d import _engine as E
from segmentation3d.utils.file_io import readlines
from segmentation3d.utils.image3d import Image3d
from segmentation3d.utils import image_tools
from segmentation3d.utils.image_io import read_image


def read_train_txt(imlist_file):
    """single-modality txt list: first line = number of cases, then image path / mask path pairs (dataset.py:12-32)"""
    lines = readlines(imlist_file)
    num_cases = int(lines[0])
    if len(lines) - 1 < num_cases * 2:
        raise ValueError('too few lines in imlist file')
    im_list, seg_list = [], []
    for i in range(num_cases):
        im_path, seg_path = lines[1 + i * 2], lines[2 + i * 2]
        assert os.path.isfile(im_path), 'image not exist: {}'.format(im_path)
        assert os.path.isfile(seg_path), 'mask not exist: {}'.format(seg_path)
        im_list.append(im_path)
        seg_list.append(seg_path)
    return im_list, seg_list


def read_train_csv(imlist_file, mode='train'):
    """csv list with columns image_name, image_path (, mask_path) (dataset.py:35-52)"""
    images_df = pd.read_csv(imlist_file)
    if mode == 'test':
        return images_df['image_name'].tolist(), images_df['image_path'].tolist()
    if mode in ('train', 'validation'):
        return images_df['image_path'].tolist(), images_df['mask_path'].tolist()
    raise ValueError('Unsupported mode type.')


class _Case(object):
    """one image / mask pair resident on the device, plus the host-side per-slice label histogram that lets MASK
    sampling pick the k-th voxel of a label (in np.argwhere order) without scanning the volume"""

    def __init__(self, image, seg, device):
        self.frame = (image.GetSpacing(), image.GetOrigin(), image.GetDirection())
        self.seg_frame = (seg.GetSpacing(), seg.GetOrigin(), seg.GetDirection())
        self.size = image.GetSize()
        self.seg_size = seg.GetSize()
        self.image = torch.from_numpy(np.array(image.array, dtype=np.float32, order='C')).to(device)
        self.seg_host = np.array(seg.array, order='C')                       # [z, y, x], label values as stored
        self.seg = torch.from_numpy(self.seg_host.astype(np.float32)).to(device)
        self._slice_counts = {}

    def label_voxel(self, label, draw):
        """`draw(n)` -> index in [0, n) of the voxel to take among the n voxels equal to `label`, enumerated like
        np.argwhere (z, then y, then x); returns (x, y, z) or None when the label is absent (draw is not called)"""
        cum = self._slice_counts.get(label)
        if cum is None:
            per_slice = (self.seg_host == label).reshape(self.seg_host.shape[0], -1).sum(1)
            cum = np.concatenate([[0], np.cumsum(per_slice)])
            self._slice_counts[label] = cum
        total = int(cum[-1])
        if total == 0:
            return None
        k = int(draw(total))
        z = int(np.searchsorted(cum, k, side='right') - 1)
        yx = np.argwhere(self.seg_host[z] == label)[k - int(cum[z])]
        return [int(yx[1]), int(yx[0]), z]


class SegmentationDataset(Dataset):
    """training data set for volumetric segmentation (constructor as dataloader/dataset.py:58-100)"""

    def __init__(self, imlist_file, num_classes, spacing, crop_size, sampling_method, random_translation, random_scale,
                 interpolation, crop_normalizers, device=None):
        if imlist_file.endswith('txt'):
            self.im_list, self.seg_list = read_train_txt(imlist_file)
        elif imlist_file.endswith('csv'):
            self.im_list, self.seg_list = read_train_csv(imlist_file)
        else:
            raise ValueError('imseg_list must be a txt file')
        self.num_classes = num_classes
        self.spacing = np.array(spacing, dtype=np.double)
        assert self.spacing.size == 3, 'only 3-element of spacing is supported'
        self.crop_size = np.array(crop_size, dtype=np.int32)
        assert self.crop_size.size == 3, 'only 3-element of crop size is supported'
        self.sampling_method = sampling_method
        assert self.sampling_method in ('CENTER', 'GLOBAL', 'MASK', 'HYBRID'), \
            'sampling_method must be CENTER, GLOBAL, MASK or HYBRID'
        self.random_translation = np.array(random_translation, dtype=np.double)
        assert self.random_translation.size == 3, 'Only 3-element of random translation is supported'
        self.random_scale = np.array(random_scale, dtype=np.double)
        assert self.random_scale.size == 2, 'Only 2-element of random scale is supported'
        self.interpolation = interpolation
        assert self.interpolation in ('LINEAR', 'NN'), 'interpolation must either be a LINEAR or NN'
        self.crop_normalizers = crop_normalizers
        assert isinstance(self.crop_normalizers, list), 'crop normalizers must be a list'
        self.device = device if device is not None else torch.device('cuda', torch.cuda.current_device())
        self._cases = {}

    def __len__(self):
        return len(self.im_list)

    def num_modality(self):
        return 1

    # ---- resident volumes --------------------------------------------------------------------------------------------
    def case(self, index):
        c = self._cases.get(index)
        if c is None:
            c = _Case(read_image(self.im_list[index]), read_image(self.seg_list[index], dtype=None), self.device)
            self._cases[index] = c
        return c

    # ---- crop centres (host, numpy global RNG, reference call order) ------------------------------------------------------
    def global_sample(self, case):
        """uniform position such that the crop lies inside the image where it fits (dataset.py:110-127)"""
        origin = case.seg_frame[1]
        im_size_mm = [case.seg_size[idx] * case.seg_frame[0][idx] for idx in range(3)]
        crop_size_mm = self.crop_size * self.spacing
        sp = np.array(origin, dtype=np.double)
        for i in range(3):
            if im_size_mm[i] > crop_size_mm[i]:
                sp[i] = origin[i] + np.random.uniform(0, im_size_mm[i] - crop_size_mm[i])
        return sp + crop_size_mm / 2

    def center_sample(self, case):
        """world coordinate of the image centre (dataset.py:129-142)"""
        spacing, origin, direction = (np.asarray(v, dtype=np.double) for v in case.seg_frame)
        end_voxel = np.array([case.seg_size[idx] - 1 for idx in range(3)], dtype=np.double)
        end_world = origin + direction.reshape(3, 3) @ (spacing * end_voxel)
        return np.array([(origin[idx] + end_world[idx]) / 2.0 for idx in range(3)], dtype=np.double)

    def _mask_sample(self, case):
        label = np.random.randint(1, self.num_classes)
        voxel = case.label_voxel(label, lambda n: np.random.randint(0, n))
        if voxel is None:                      # if no segmentation
            return self.global_sample(case)
        spacing, origin, direction = (np.asarray(v, dtype=np.double) for v in case.seg_frame)
        return origin + direction.reshape(3, 3) @ (spacing * np.array(voxel, dtype=np.double))

    def sample_crop_geometry(self, index):
        """(centre, crop spacing) of the next sample of case `index` -- consumes the RNG like dataset.py:166-200"""
        case = self.case(index)
        if self.sampling_method == 'CENTER':
            center = self.center_sample(case)
        elif self.sampling_method == 'GLOBAL':
            center = self.global_sample(case)
        elif self.sampling_method == 'MASK':
            center = self._mask_sample(case)
        else:  # HYBRID
            center = self.global_sample(case) if index % 2 else self._mask_sample(case)
        center = center + np.random.uniform(-self.random_translation, self.random_translation, size=[3])
        crop_spacing = self.spacing * np.random.uniform(self.random_scale[0], self.random_scale[1])
        return center, crop_spacing

    # ---- the sample ---------------------------------------------------------------------------------------------------
    def __getitem__(self, index):
        """-> (image crop [1, z, y, x], mask crop [1, z, y, x] float labels, frame (15 floats), case name); device tensors"""
        case = self.case(index)
        image_path = self.im_list[index]
        case_name = os.path.basename(os.path.dirname(image_path)) + '_' + os.path.basename(image_path)
        center, crop_spacing = self.sample_crop_geometry(index)
        im = image_tools.crop_image_device(case.image, case.frame, center, self.crop_size, crop_spacing, self.interpolation)
        if self.crop_normalizers[0] is not None:
            im = image_tools.normalize_crop_device(im, self.crop_normalizers[0])
        seg = image_tools.crop_image_device(case.seg, case.seg_frame, center, self.crop_size, crop_spacing, 'NN')
        origin = image_tools.crop_origin(center, self.crop_size, crop_spacing)
        frame = np.array(list(crop_spacing) + list(origin) + list(case.seg_frame[2]), dtype=np.float32)
        return im.unsqueeze(0), seg.unsqueeze(0), frame, case_name


class DeviceCropLoader(object):
    """batches of device-resident samples in sampler order: iterable of (crops [B,1,z,y,x], masks [B,1,z,y,x], frames,
    case names).  Replaces torch's DataLoader + worker processes (core/seg_train.py:69-70) for the GPU data path."""

    def __init__(self, dataset, sampler, batch_size, drop_last=False):
        self.dataset, self.sampler, self.batch_size, self.drop_last = dataset, sampler, int(batch_size), drop_last

    def __len__(self):
        n = len(self.sampler)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        ims, segs, frames, names = [], [], [], []
        for index in self.sampler:
            im, seg, frame, name = self.dataset[index]
            ims.append(im)
            segs.append(seg)
            frames.append(frame)
            names.append(name)
            if len(ims) == self.batch_size:
                yield torch.stack(ims), torch.stack(segs), np.stack(frames), names
                ims, segs, frames, names = [], [], [], []
        if ims and not self.drop_last:
            yield torch.stack(ims), torch.stack(segs), np.stack(frames), names
