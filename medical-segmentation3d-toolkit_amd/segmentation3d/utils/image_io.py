"""Image file IO without SimpleITK: `read_image` / `write_image` dispatch on the file
extension to MetaImage (.mha / .mhd, utils/mha_io.py, now also zlib-compressed data) or NIfTI-1 (.nii / .nii.gz, this
module).  The reference goes through `sitk.ReadImage` / `sitk.WriteImage` (core/seg_infer.py:414,467,
dataloader/dataset.py:160-164; intermediate crops are saved as .nii.gz, utils/image_tools.py:92-100).

NIfTI frames are converted the way ITK's NiftiImageIO does, so that an `Image3d` carries the same spacing / origin /
direction SimpleITK would report: NIfTI world coordinates are RAS, ITK's are LPS -> the first two rows of the affine
(rotation and offset) change sign.  The sform is used when sform_code > 0, else the qform quaternion, else the pixdim
spacing alone.  Parity unpinned: neither SimpleITK nor nibabel is installable here; the reader is checked against the
NIfTI-1 header layout and by round trips.
"""
import gzip
import struct

import numpy as np

from segmentation3d.utils.image3d import Image3d
from segmentation3d.utils.mha_io import read_mha, write_mha

_NIFTI_TYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16,
                768: np.uint32}
_NIFTI_CODES = {np.dtype(v).name: k for k, v in _NIFTI_TYPES.items()}


def _is_nifti(path):
    return path.endswith('.nii') or path.endswith('.nii.gz')


def _quaternion_rotation(b, c, d):
    a = np.sqrt(max(0.0, 1.0 - (b * b + c * c + d * d)))
    return np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                     [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                     [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])


def read_nifti(path, dtype=np.float32):
    opener = gzip.open if path.endswith('.gz') else open
    with opener(path, 'rb') as f:
        raw = f.read()
    end = '<' if struct.unpack('<i', raw[:4])[0] == 348 else '>'
    if struct.unpack(end + 'i', raw[:4])[0] != 348:
        raise ValueError('{}: not a NIfTI-1 file'.format(path))
    dim = struct.unpack(end + '8h', raw[40:56])
    if dim[0] < 3 or any(d > 1 for d in dim[4:1 + dim[0]]):
        raise ValueError('{}: only 3-D scalar volumes are supported (dim = {})'.format(path, dim))
    datatype = struct.unpack(end + 'h', raw[70:72])[0]
    if datatype not in _NIFTI_TYPES:
        raise ValueError('{}: unsupported NIfTI datatype {}'.format(path, datatype))
    pixdim = struct.unpack(end + '8f', raw[76:108])
    vox_offset = int(struct.unpack(end + 'f', raw[108:112])[0])
    slope, inter = struct.unpack(end + '2f', raw[112:120])
    qform_code, sform_code = struct.unpack(end + '2h', raw[252:256])
    nx, ny, nz = dim[1:4]
    et = np.dtype(_NIFTI_TYPES[datatype]).newbyteorder(end)
    data = np.frombuffer(raw, dtype=et, count=nx * ny * nz, offset=max(vox_offset, 352)).reshape(nz, ny, nx)
    if slope not in (0.0, 1.0) or inter != 0.0:
        data = data.astype(np.float64) * (slope if slope != 0.0 else 1.0) + inter
    array = np.array(data, dtype=dtype if dtype is not None else data.dtype.newbyteorder('='), order='C')
    spacing = np.array([abs(pixdim[1]) or 1.0, abs(pixdim[2]) or 1.0, abs(pixdim[3]) or 1.0])
    if sform_code > 0:
        affine = np.array([struct.unpack(end + '4f', raw[280 + 16 * r:296 + 16 * r]) for r in range(3)], dtype=np.float64)
        rot, offset = affine[:, :3], affine[:, 3]
        spacing = np.sqrt((rot ** 2).sum(0))
        direction = rot / spacing
    elif qform_code > 0:
        b, c, d, qx, qy, qz = struct.unpack(end + '6f', raw[256:280])
        direction = _quaternion_rotation(b, c, d)
        if pixdim[0] < 0:                       # qfac: the third axis is flipped
            direction[:, 2] = -direction[:, 2]
        offset = np.array([qx, qy, qz], dtype=np.float64)
    else:
        direction, offset = np.eye(3), np.zeros(3)
    lps = np.diag([-1.0, -1.0, 1.0])            # RAS (NIfTI) -> LPS (ITK)
    return Image3d(array, spacing.tolist(), (lps @ offset).tolist(), (lps @ direction).ravel().tolist())


def write_nifti(image, path):
    array = np.ascontiguousarray(image.array)
    if array.dtype.name not in _NIFTI_CODES:
        raise ValueError('unsupported dtype {}'.format(array.dtype.name))
    nz, ny, nx = array.shape
    spacing = np.asarray(image.GetSpacing(), dtype=np.float64)
    lps = np.diag([-1.0, -1.0, 1.0])
    rot = lps @ np.asarray(image.GetDirection(), dtype=np.float64).reshape(3, 3) * spacing   # columns scaled by spacing
    offset = lps @ np.asarray(image.GetOrigin(), dtype=np.float64)
    hdr = bytearray(352)
    struct.pack_into('<i', hdr, 0, 348)
    struct.pack_into('<8h', hdr, 40, 3, nx, ny, nz, 1, 1, 1, 1)
    struct.pack_into('<hh', hdr, 70, _NIFTI_CODES[array.dtype.name], array.dtype.itemsize * 8)
    struct.pack_into('<8f', hdr, 76, 1.0, spacing[0], spacing[1], spacing[2], 0.0, 0.0, 0.0, 0.0)
    struct.pack_into('<f', hdr, 108, 352.0)
    struct.pack_into('<2f', hdr, 112, 1.0, 0.0)
    hdr[123] = 2                                 # xyzt_units: millimetres
    struct.pack_into('<2h', hdr, 252, 0, 1)      # qform_code 0, sform_code 1 (scanner)
    for r in range(3):
        struct.pack_into('<4f', hdr, 280 + 16 * r, rot[r, 0], rot[r, 1], rot[r, 2], offset[r])
    hdr[344:348] = b'n+1\x00'
    opener = gzip.open if path.endswith('.gz') else open
    with opener(path, 'wb') as f:
        f.write(bytes(hdr))
        f.write(array.astype(array.dtype.newbyteorder('<'), copy=False).tobytes())


def read_image(path, dtype=np.float32):
    """Image3d from a .mha / .mhd / .nii / .nii.gz file (dtype=None keeps the stored element type)"""
    if _is_nifti(path):
        return read_nifti(path, dtype)
    if path.endswith('.mha') or path.endswith('.mhd'):
        return read_mha(path, dtype)
    raise ValueError('unsupported image format: {} (MetaImage .mha/.mhd and NIfTI .nii/.nii.gz are supported)'.format(path))


def write_image(image, path):
    if _is_nifti(path):
        return write_nifti(image, path)
    if path.endswith('.mha') or path.endswith('.mhd'):
        return write_mha(image, path)
    raise ValueError('unsupported image format: {}'.format(path))
