#!/usr/bin/env python
"""Writes the independent file-format fixtures under tests/golden/images/ (SURVEY.md section 8f row f3).

The files are assembled here byte by byte -- literal MetaImage header text + struct-packed payload, and the 348-byte
NIfTI-1 header filled field by field at the offsets of nifti1.h -- WITHOUT importing anything from the product package
(no mha_io / image_io), so that the product readers meet files they did not write.  `expected.json` holds what
SimpleITK reports for such a file according to the formats' definitions:

* MetaImage (ITK MetaImageIO): voxel (x, y, z) is element x + X*(y + Y*z) of the payload; `TransformMatrix` lists the
  direction cosines AXIS BY AXIS (first the physical direction of the x axis, then y, then z), i.e. the COLUMNS of the
  direction matrix whose row-major flattening `sitk.Image.GetDirection()` returns; `Offset` = origin, `ElementSpacing` =
  spacing; `CompressedData = True` = one zlib stream; `BinaryDataByteOrderMSB = True` = big-endian elements;
  `ElementDataFile = <name>` = payload in a separate file beside the header (.mhd).
* NIfTI-1 (nifti1.h + ITK NiftiImageIO): world frame RAS -> ITK's LPS negates the first two rows of the affine; sform
  rows at 280/296/312 (spacing = column norms); qform quaternion (b, c, d) at 256, offsets at 268, qfac = pixdim[0]
  flips the third axis; value = stored * scl_slope + scl_inter when scl_slope != 0.

Parity unpinned in the strict sense (SimpleITK is not installable here and the reference ships no image file): these are
known-answer vectors derived from the format definitions, not outputs of the reference.

    python tests/golden/make_image_fixtures.py
"""
import gzip
import json
import math
import os
import struct
import zlib

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'images')


def volume(shape_zyx, dtype, scale=1.0, offset=0.0):
    z, y, x = np.meshgrid(*[np.arange(n) for n in shape_zyx], indexing='ij')
    return ((x + 10 * y + 100 * z) * scale + offset).astype(dtype)


def rot_zx(az_deg, ax_deg):
    a, b = math.radians(az_deg), math.radians(ax_deg)
    rz = np.array([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]])
    rx = np.array([[1, 0, 0], [0, math.cos(b), -math.sin(b)], [0, math.sin(b), math.cos(b)]])
    return rz @ rx


def mha_header(size_xyz, etype, spacing, origin, direction, extra, datafile='LOCAL'):
    cols = [direction[:, i] for i in range(3)]                 # axis by axis = columns of the direction matrix
    lines = ['ObjectType = Image', 'NDims = 3', 'BinaryData = True'] + extra + [
        'TransformMatrix = ' + ' '.join('{:.17g}'.format(v) for c in cols for v in c),
        'Offset = ' + ' '.join('{:.17g}'.format(v) for v in origin),
        'CenterOfRotation = 0 0 0', 'AnatomicalOrientation = RAI',
        'ElementSpacing = ' + ' '.join('{:.17g}'.format(v) for v in spacing),
        'DimSize = {} {} {}'.format(*size_xyz), 'ElementType = ' + etype, 'ElementDataFile = ' + datafile]
    return ('\n'.join(lines) + '\n').encode('ascii')


def nifti_header(end, size_xyz, datatype, bitpix, pixdim, slope, inter, qform_code, sform_code, quatern, qoffset, srows):
    h = bytearray(352)
    struct.pack_into(end + 'i', h, 0, 348)                                   # sizeof_hdr
    struct.pack_into(end + '8h', h, 40, 3, size_xyz[0], size_xyz[1], size_xyz[2], 1, 1, 1, 1)   # dim[8]
    struct.pack_into(end + 'h', h, 70, datatype)                             # datatype
    struct.pack_into(end + 'h', h, 72, bitpix)                               # bitpix
    struct.pack_into(end + '8f', h, 76, *pixdim)                             # pixdim[8] (pixdim[0] = qfac)
    struct.pack_into(end + 'f', h, 108, 352.0)                               # vox_offset
    struct.pack_into(end + 'f', h, 112, slope)                               # scl_slope
    struct.pack_into(end + 'f', h, 116, inter)                               # scl_inter
    h[123] = 2                                                               # xyzt_units: mm
    struct.pack_into(end + 'h', h, 252, qform_code)
    struct.pack_into(end + 'h', h, 254, sform_code)
    struct.pack_into(end + '3f', h, 256, *quatern)                           # quatern_b, _c, _d
    struct.pack_into(end + '3f', h, 268, *qoffset)                           # qoffset_x, _y, _z
    for r in range(3):
        struct.pack_into(end + '4f', h, 280 + 16 * r, *srows[r])             # srow_x, srow_y, srow_z
    h[344:348] = b'n+1\x00'                                                  # magic
    return bytes(h)


def main():
    os.makedirs(OUT, exist_ok=True)
    expected = {}
    lps = np.diag([-1.0, -1.0, 1.0])

    # ---- MetaImage 1: float32, raw, oblique (non-symmetric) direction, anisotropic spacing
    arr = volume((3, 4, 5), np.float32, 0.25, -7.5)
    R = rot_zx(30.0, 20.0)
    sp, org = (0.5, 0.75, 2.0), (-10.5, 20.25, 3.0)
    with open(os.path.join(OUT, 'raw_f32_oblique.mha'), 'wb') as f:
        f.write(mha_header((5, 4, 3), 'MET_FLOAT', sp, org, R, ['BinaryDataByteOrderMSB = False', 'CompressedData = False']))
        f.write(struct.pack('<{}f'.format(arr.size), *arr.ravel().tolist()))
    expected['raw_f32_oblique.mha'] = dict(array=arr.tolist(), dtype='float32', spacing=sp, origin=org, direction=R.ravel().tolist())

    # ---- MetaImage 2: int16, zlib-compressed payload, identity frame
    arr = volume((4, 3, 6), np.int16, 3, -500)
    blob = zlib.compress(struct.pack('<{}h'.format(arr.size), *arr.ravel().tolist()), 6)
    with open(os.path.join(OUT, 'zlib_i16.mha'), 'wb') as f:
        f.write(mha_header((6, 3, 4), 'MET_SHORT', (1, 1, 1), (0, 0, 0), np.eye(3),
                           ['BinaryDataByteOrderMSB = False', 'CompressedData = True', 'CompressedDataSize = {}'.format(len(blob))]))
        f.write(blob)
    expected['zlib_i16.mha'] = dict(array=arr.tolist(), dtype='int16', spacing=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0),
                                    direction=np.eye(3).ravel().tolist())

    # ---- MetaImage 3: .mhd header + separate big-endian uint16 payload, axes permuted (x axis along physical y, ...)
    arr = volume((2, 3, 4), np.uint16, 257, 1)
    P = np.array([[0.0, 0.0, 1.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])     # x axis -> +y, y axis -> +z, z axis -> +x
    with open(os.path.join(OUT, 'msb_u16.mhd'), 'wb') as f:
        f.write(mha_header((4, 3, 2), 'MET_USHORT', (1.25, 1.5, 0.8), (1.0, -2.0, 3.5), P,
                           ['BinaryDataByteOrderMSB = True', 'CompressedData = False'], datafile='msb_u16.raw'))
    with open(os.path.join(OUT, 'msb_u16.raw'), 'wb') as f:
        f.write(struct.pack('>{}H'.format(arr.size), *arr.ravel().tolist()))
    expected['msb_u16.mhd'] = dict(array=arr.tolist(), dtype='uint16', spacing=(1.25, 1.5, 0.8), origin=(1.0, -2.0, 3.5),
                                   direction=P.ravel().tolist())

    # ---- NIfTI 1: int16 with scl_slope / scl_inter, sform only (oblique, with spacing), little-endian
    arr = volume((3, 4, 5), np.int16, 2, -40)
    R = rot_zx(-25.0, 10.0)
    sp = np.array([0.8, 0.9, 2.5])
    A = R * sp                                                   # columns scaled by the spacing
    off = np.array([12.0, -34.5, 56.25])
    with open(os.path.join(OUT, 'sform_i16_slope.nii'), 'wb') as f:
        f.write(nifti_header('<', (5, 4, 3), 4, 16, (1.0, 9.0, 9.0, 9.0, 0, 0, 0, 0), 0.5, -3.0, 0, 1, (0, 0, 0), (0, 0, 0),
                             [list(A[r]) + [off[r]] for r in range(3)]))      # pixdim deliberately wrong: the sform rules
        f.write(struct.pack('<{}h'.format(arr.size), *arr.ravel().tolist()))
    expected['sform_i16_slope.nii'] = dict(array=(arr.astype(np.float64) * 0.5 - 3.0).tolist(), dtype='int16 * 0.5 - 3',
                                           spacing=sp.tolist(), origin=(lps @ off).tolist(), direction=(lps @ R).ravel().tolist())

    # ---- NIfTI 2: float32, gzip, qform only: rotation by 90 degrees about z (a = d = sqrt(1/2)), qfac = -1
    arr = volume((4, 2, 3), np.float32, 0.5, 0.25)
    qoff = np.array([-5.0, 6.0, 7.5])
    with gzip.GzipFile(os.path.join(OUT, 'qform_f32.nii.gz'), 'wb', mtime=0) as f:      # mtime 0: reproducible bytes
        f.write(nifti_header('<', (3, 2, 4), 16, 32, (-1.0, 1.5, 0.7, 3.0, 0, 0, 0, 0), 0.0, 0.0, 1, 0,
                             (0.0, 0.0, math.sqrt(0.5)), tuple(qoff), [[0, 0, 0, 0]] * 3))
        f.write(struct.pack('<{}f'.format(arr.size), *arr.ravel().tolist()))
    Rq = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, -1.0]])      # Rz(90) with the third COLUMN flipped by qfac
    expected['qform_f32.nii.gz'] = dict(array=arr.tolist(), dtype='float32', spacing=(1.5, 0.7, 3.0), origin=(lps @ qoff).tolist(),
                                        direction=(lps @ Rq).ravel().tolist())

    # ---- NIfTI 3: uint8, BIG-endian header, sform and qform both present and in agreement (axis-aligned, flipped x)
    arr = volume((2, 3, 4), np.uint8, 1, 7)
    sp = np.array([2.0, 1.0, 4.0])
    F = np.diag([-1.0, 1.0, 1.0])                                # x axis runs towards -R = L
    off = np.array([100.0, -50.0, 25.0])
    A = F * sp
    with open(os.path.join(OUT, 'msb_u8_both_forms.nii'), 'wb') as f:
        # quaternion of diag(-1, 1, 1) x qfac: det = -1 -> qfac = -1 and the proper rotation is diag(-1, 1, -1) = 180 deg about y
        f.write(nifti_header('>', (4, 3, 2), 2, 8, (-1.0, 2.0, 1.0, 4.0, 0, 0, 0, 0), 1.0, 0.0, 1, 1, (0.0, 1.0, 0.0), tuple(off),
                             [list(A[r]) + [off[r]] for r in range(3)]))
        f.write(struct.pack('>{}B'.format(arr.size), *arr.ravel().tolist()))
    expected['msb_u8_both_forms.nii'] = dict(array=arr.tolist(), dtype='uint8', spacing=sp.tolist(), origin=(lps @ off).tolist(),
                                             direction=(lps @ F).ravel().tolist())

    with open(os.path.join(OUT, 'expected.json'), 'w') as f:
        json.dump(expected, f, indent=1, sort_keys=True)
    print('wrote', sorted(os.listdir(OUT)))


if __name__ == '__main__':
    main()
