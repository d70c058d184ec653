"""Pillow's LANCZOS resize for 8-bit images -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference calls `PIL_image.resize((w // 2, h // 2), PIL.Image.LANCZOS)` in the > 2.5 MP branch of FCN_LectureNet.binarize
(AccessMath/lecturenet_v1/FCN_lecturenet.py:434-437).  Pillow is a third-party dependency that is not vendored under
/root/reference and not version-pinned by it (no requirements file); this restates its published algorithm
(src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc; two passes,
horizontal first, 8-bit intermediate, 22-bit fixed-point coefficients) in numpy.
Parity status: pinned by tests/golden/g6b_lanczos.npz, recorded from Pillow 12.2.0 in the build container
(tests/golden/make_golden_resize.py; tests/test_oracle_golden.py::test_g6b_lanczos).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _sinc(x):
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def lanczos_filter(x):
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3)
    return 0.0


def precompute_coeffs(in_size, out_size, support=3.0, filt=lanczos_filter):
    """bounds [out][2] (xmin, count) and fixed-point coefficients [out][ksize] (precompute_coeffs + normalize_coeffs_8bpc)"""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    sup = support * filterscale
    ksize = int(math.ceil(sup)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - sup + 0.5), 0)
        xmax = min(int(center + sup + 0.5), in_size) - xmin
        w = [filt((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x, v in enumerate(w):
            k = v / ww if ww != 0.0 else v
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    img = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((len(bounds),) + img.shape[1:], np.uint8)
    for xx, (xmin, n) in enumerate(bounds):
        acc = np.tensordot(kk[xx, :n].astype(np.int64), img[xmin:xmin + n], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis)


def resize_lanczos(img, out_w, out_h):
    """uint8 [H, W, C] -> uint8 [out_h, out_w, C]: horizontal pass, then vertical pass on its 8-bit result"""
    h, w = img.shape[:2]
    if out_w != w:
        img = _pass(img, *precompute_coeffs(w, out_w), axis=1)
    if out_h != h:
        img = _pass(img, *precompute_coeffs(h, out_h), axis=0)
    return img
