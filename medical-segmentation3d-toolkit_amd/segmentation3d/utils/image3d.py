"""Image3d -- a minimal stand-in for the `SimpleITK.Image` objects the reference passes around.

SimpleITK is not a dependency of this package.  The hot path only needs voxel data plus the frame (size, spacing,
origin, direction) and the handful of accessors the reference calls (`GetSize`, `GetSpacing`, `GetOrigin`,
`GetDirection`, `CopyInformation`).  Voxel (x, y, z) of an image with size (X, Y, Z) is `array[z, y, x]`, the same
convention as `sitk.GetArrayFromImage` (utils/image_tools.py:279-281, 448 of the reference).
"""
import numpy as np


class Image3d(object):
    def __init__(self, array, spacing=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0),
                 direction=(1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0)):
        array = np.asarray(array)
        if array.ndim != 3:
            raise ValueError('Image3d needs a 3-D [z, y, x] array, got shape {}'.format(array.shape))
        self.array = array
        self.spacing = tuple(float(s) for s in spacing)
        self.origin = tuple(float(o) for o in origin)
        self.direction = tuple(float(d) for d in direction)

    # --- sitk-style accessors ---
    def GetSize(self):
        z, y, x = self.array.shape
        return (int(x), int(y), int(z))

    def GetSpacing(self):
        return self.spacing

    def GetOrigin(self):
        return self.origin

    def GetDirection(self):
        return self.direction

    def CopyInformation(self, other):
        self.spacing, self.origin, self.direction = tuple(other.GetSpacing()), tuple(other.GetOrigin()), tuple(other.GetDirection())

    def frame(self):
        return {'spacing': self.spacing, 'origin': self.origin, 'direction': self.direction}

    def like(self, array):
        return Image3d(array, self.spacing, self.origin, self.direction)


def image_size_spacing(image):
    """(size xyz, spacing xyz) of an Image3d / sitk-like object, or of a (size, spacing) pair"""
    if hasattr(image, 'GetSize'):
        return [int(v) for v in image.GetSize()], [float(v) for v in image.GetSpacing()]
    size, spacing = image
    return [int(v) for v in size], [float(v) for v in spacing]
