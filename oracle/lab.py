"""sRGB <-> CIE-Lab (D65, 2 degree observer), float64 restatement.

Follows scikit-image 0.18.3 ``skimage/color/colorconv.py`` (the library the
reference calls at methods/linear.py:25,26,40):

* ``rgb2xyz``  l.657-661   gamma expansion + ``arr @ xyz_from_rgb.T``
* ``xyz2lab``  l.950-969   white scaling, cbrt / linear toe, L a b
* ``lab2xyz``  l.1010-1032 inverse, ``z < 0 -> 0``, cube / linear toe
* ``xyz2rgb``  l.612-619   ``arr @ inv(M).T``, gamma compression, clip [0,1]

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import numpy as np

# sRGB -> XYZ matrix, skimage colorconv.py l.338-340
XYZ_FROM_RGB = np.array([[0.412453, 0.357580, 0.180423],
                         [0.212671, 0.715160, 0.072169],
                         [0.019334, 0.119193, 0.950227]], dtype=np.float64)
# skimage inverts it with scipy.linalg.inv at import (l.342); np.linalg.inv
# agrees to the last bit or two, far below any tolerance used here.
RGB_FROM_XYZ = np.linalg.inv(XYZ_FROM_RGB)
# D65 / 2 degree white point, colorconv.py l.426
WHITE_D65 = np.array([0.95047, 1.0, 1.08883], dtype=np.float64)


def rgb2xyz(rgb):
    arr = np.array(rgb, dtype=np.float64, copy=True)
    mask = arr > 0.04045
    arr[mask] = np.power((arr[mask] + 0.055) / 1.055, 2.4)
    arr[~mask] /= 12.92
    return arr @ XYZ_FROM_RGB.T


def xyz2lab(xyz):
    arr = np.asarray(xyz, dtype=np.float64) / WHITE_D65
    mask = arr > 0.008856
    arr[mask] = np.cbrt(arr[mask])
    arr[~mask] = 7.787 * arr[~mask] + 16.0 / 116.0
    x, y, z = arr[..., 0], arr[..., 1], arr[..., 2]
    L = (116.0 * y) - 16.0
    a = 500.0 * (x - y)
    b = 200.0 * (y - z)
    return np.stack([L, a, b], axis=-1)


def lab2xyz(lab):
    arr = np.array(lab, dtype=np.float64, copy=True)
    L, a, b = arr[..., 0], arr[..., 1], arr[..., 2]
    y = (L + 16.0) / 116.0
    x = (a / 500.0) + y
    z = y - (b / 200.0)
    z = np.where(z < 0, 0.0, z)          # skimage warns and zeroes (l.1016-1020)
    out = np.stack([x, y, z], axis=-1)
    mask = out > 0.2068966
    out[mask] = np.power(out[mask], 3.0)
    out[~mask] = (out[~mask] - 16.0 / 116.0) / 7.787
    out *= WHITE_D65
    return out


def xyz2rgb(xyz):
    arr = np.asarray(xyz, dtype=np.float64) @ RGB_FROM_XYZ.T
    mask = arr > 0.0031308
    arr[mask] = 1.055 * np.power(arr[mask], 1 / 2.4) - 0.055
    arr[~mask] *= 12.92
    np.clip(arr, 0, 1, out=arr)
    return arr


def rgb2lab(rgb):
    return xyz2lab(rgb2xyz(rgb))


def lab2rgb(lab):
    return xyz2rgb(lab2xyz(lab))
