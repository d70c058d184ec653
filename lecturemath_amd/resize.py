"""Frame pre / post-processing of FCN_LectureNet.binarize's > 2.5 MP branch on the device (csrc/lm_resize.hip):
PIL.Image.resize(..., LANCZOS) (FCN_lecturenet.py:434-437) and the INTER_NEAREST enlargement back (:481-486).

Pillow's resize applies, per axis, integer taps that depend on the two sizes only.  They are computed here once per size pair with
the formulas of Pillow's precompute_coeffs / normalize_coeffs_8bpc (float64 -> 22-bit fixed point) and cached on the device; all
arithmetic on pixels runs in liblecturemath_hip.so and reproduces Pillow byte for byte (tests/golden/g6b_lanczos.npz)."""
import math

import numpy as np

from . import _lib
from .device import Backend

PRECISION_BITS = 32 - 8 - 2


def _lanczos(x):
    def sinc(v):
        if v == 0.0:
            return 1.0
        v *= math.pi
        return math.sin(v) / v
    return sinc(x) * sinc(x / 3) if -3.0 <= x < 3.0 else 0.0


def coefficients(in_size, out_size):
    """(bounds int32 [out][2] = first input index and tap count, taps int32 [out][ksize]) of one axis"""
    scale = float(in_size) / out_size
    fscale = max(scale, 1.0)
    support = 3.0 * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    inv = 1.0 / fscale
    one = float(1 << PRECISION_BITS)
    cache = {}
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        n = min(int(center + support + 0.5), in_size) - xmin
        key = (center - xmin, n)                     # interior pixels of an integer ratio repeat the same taps
        if key not in cache:
            w = [_lanczos((x + xmin - center + 0.5) * inv) for x in range(n)]
            ww = 0.0
            for v in w:
                ww += v
            row = []
            for v in w:
                k = v / ww if ww != 0.0 else v
                row.append(int(-0.5 + k * one) if k < 0 else int(0.5 + k * one))
            cache[key] = row
        kk[xx, :n] = cache[key]
        bounds[xx] = (xmin, n)
    return bounds, kk


class DeviceResizer:
    """LANCZOS resize and NEAREST enlargement of device uint8 images [H, W] or [H, W, 3]"""

    def __init__(self, lib=None):
        self.lib = lib or _lib.load()
        self.be = Backend(self.lib)
        self._tables = {}

    def _axis(self, n_in, n_out):
        if n_in == n_out:
            return None, None, 0
        key = (n_in, n_out)
        if key not in self._tables:
            b, k = coefficients(n_in, n_out)
            self._tables[key] = (self.be.from_host(b.reshape(-1)), self.be.from_host(k.reshape(-1)), k.shape[1])
        return self._tables[key]

    def lanczos(self, img, out_w, out_h):
        """img: device (or host numpy) uint8 [H, W] / [H, W, 3] -> device uint8 of size out_h x out_w, Pillow's LANCZOS"""
        if isinstance(img, np.ndarray):
            img = self.be.from_host(img)
        h, w = int(img.shape[0]), int(img.shape[1])
        c = 1 if len(img.shape) == 2 else int(img.shape[2])
        bh, kh, nh = self._axis(w, out_w)
        bv, kv, nv = self._axis(h, out_h)
        shape = (out_h, out_w) if len(img.shape) == 2 else (out_h, out_w, c)
        out = self.be.empty(shape, np.uint8)
        tmp = self.be.empty((h * out_w * c,), np.uint8)
        self.lib.check(self.lib.lm_resample_rgb8(_lib.ptr(img), h, w, c, _lib.ptr(tmp), _lib.ptr(out), out_h, out_w, _lib.ptr(bh), _lib.ptr(kh), nh,
                                                 _lib.ptr(bv), _lib.ptr(kv), nv, self.be.stream()))
        return out

    def nearest(self, img, out_w, out_h):
        """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_NEAREST) for integer ratios: out[y, x] = img[y * h // out_h, x * w // out_w]"""
        if isinstance(img, np.ndarray):
            img = self.be.from_host(img)
        h, w = int(img.shape[0]), int(img.shape[1])
        c = 1 if len(img.shape) == 2 else int(img.shape[2])
        out = self.be.empty((out_h, out_w) if len(img.shape) == 2 else (out_h, out_w, c), np.uint8)
        self.lib.check(self.lib.lm_upsample_nearest_u8(_lib.ptr(img), h, w, c, _lib.ptr(out), out_h, out_w, self.be.stream()))
        return out
