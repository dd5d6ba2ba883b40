"""Minimal MetaImage (.mha / .mhd) reader and writer -- enough to run `seg_infer` on synthetic volumes without
SimpleITK (the reference reads/writes through `sitk.ReadImage` / `sitk.WriteImage`, core/seg_infer.py:414,467).
3-D scalar element types; raw or zlib-compressed data (read), raw (write)."""
import os
import zlib

import numpy as np

from segmentation3d.utils.image3d import Image3d

_MET = {'MET_FLOAT': np.float32, 'MET_DOUBLE': np.float64, 'MET_SHORT': np.int16, 'MET_USHORT': np.uint16,
        'MET_CHAR': np.int8, 'MET_UCHAR': np.uint8, 'MET_INT': np.int32, 'MET_UINT': np.uint32}
_MET_INV = {np.dtype(v).name: k for k, v in _MET.items()}


def read_mha(path, dtype=np.float32):
    with open(path, 'rb') as f:
        header = {}
        while True:
            line = f.readline()
            if not line:
                raise ValueError('{}: ElementDataFile not found in header'.format(path))
            key, _, value = line.decode('ascii', 'replace').partition('=')
            key, value = key.strip(), value.strip()
            header[key] = value
            if key == 'ElementDataFile':
                break
        compressed = header.get('CompressedData', 'False').lower() == 'true'   # zlib stream (sitk.WriteImage(..., True))
        if int(header.get('NDims', 3)) != 3:
            raise ValueError('only 3-D images are supported')
        size = [int(v) for v in header['DimSize'].split()]
        et = _MET[header['ElementType']]
        count = size[0] * size[1] * size[2]
        if header['ElementDataFile'] == 'LOCAL':
            blob = f.read()
        else:
            with open(os.path.join(os.path.dirname(path), header['ElementDataFile']), 'rb') as g:
                blob = g.read()
        if compressed:
            blob = zlib.decompress(blob)
        data = np.frombuffer(blob, dtype=et, count=count)
    if header.get('BinaryDataByteOrderMSB', 'False').lower() == 'true':
        data = data.byteswap()
    array = data.reshape(size[2], size[1], size[0])
    if dtype is not None:
        array = array.astype(dtype)
    spacing = [float(v) for v in header.get('ElementSpacing', '1 1 1').split()]
    origin = [float(v) for v in header.get('Offset', header.get('Position', '0 0 0')).split()]
    # ITK's MetaImageIO stores the direction cosines AXIS BY AXIS: the i-th triple of TransformMatrix is the physical
    # direction of image axis i = COLUMN i of the direction matrix, whose row-major flattening sitk's GetDirection() returns
    tm = [float(v) for v in header.get('TransformMatrix', header.get('Orientation', '1 0 0 0 1 0 0 0 1')).split()]
    direction = [tm[3 * c + r] for r in range(3) for c in range(3)]
    return Image3d(array, spacing, origin, direction)


def write_mha(image, path):
    array = np.ascontiguousarray(image.array)
    name = array.dtype.name
    if name not in _MET_INV:
        raise ValueError('unsupported dtype {}'.format(name))
    x, y, z = image.GetSize()
    lines = ['ObjectType = Image', 'NDims = 3', 'BinaryData = True', 'BinaryDataByteOrderMSB = False',
             'CompressedData = False',
             # axis by axis = the columns of the direction matrix (see read_mha)
             'TransformMatrix = ' + ' '.join(repr(float(image.GetDirection()[3 * r + c])) for c in range(3) for r in range(3)),
             'Offset = ' + ' '.join(repr(float(v)) for v in image.GetOrigin()),
             'CenterOfRotation = 0 0 0', 'AnatomicalOrientation = RAI',
             'ElementSpacing = ' + ' '.join(repr(float(v)) for v in image.GetSpacing()),
             'DimSize = {} {} {}'.format(x, y, z), 'ElementType = ' + _MET_INV[name], 'ElementDataFile = LOCAL']
    with open(path, 'wb') as f:
        f.write(('\n'.join(lines) + '\n').encode('ascii'))
        f.write(array.tobytes())
