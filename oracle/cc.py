"""ctypes front-end of oracle/cc_oracle.c (TEST INFRASTRUCTURE; see oracle/__init__.py).

Each wrapper names the reference code its C function restates (paths relative to
/root/reference/ACCESS2021_release).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MIN_CC_PIXELS = 20  # AccessMath/preprocessing/content/labeler.py:22

_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_f32p = ctypes.POINTER(ctypes.c_float)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.orc_label4.argtypes = [_u8p, ctypes.c_int, ctypes.c_int, _i32p]
        L.orc_label4.restype = ctypes.c_int
        L.orc_age_boundaries.argtypes = [_i32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         _i32p, _i32p, _i32p, _i32p, _i32p, _f32p]
        L.orc_age_boundaries.restype = ctypes.c_int
        L.orc_extract.argtypes = [_i32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                  _i32p, _i64p, _u8p, _i64p]
        L.orc_extract.restype = ctypes.c_int
        L.orc_overlap.argtypes = [_i32p, _u8p, _i32p, _u8p]
        L.orc_overlap.restype = ctypes.c_int64
        L.orc_threshold_invert.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int, _u8p]
        L.orc_threshold_invert.restype = None
        L.orc_stab_new.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int]
        L.orc_stab_new.restype = ctypes.c_void_p
        L.orc_stab_free.argtypes = [ctypes.c_void_p]
        L.orc_stab_free.restype = None
        L.orc_stab_add_frame.argtypes = [ctypes.c_void_p, _u8p, ctypes.c_int]
        L.orc_stab_add_frame.restype = ctypes.c_int
        for name, res in (("orc_stab_n_unique", ctypes.c_int32), ("orc_stab_n_frames", ctypes.c_int32),
                          ("orc_stab_tempo_count", ctypes.c_int64), ("orc_stab_n_log", ctypes.c_int64),
                          ("orc_stab_n_active", ctypes.c_int32)):
            getattr(L, name).argtypes = [ctypes.c_void_p]
            getattr(L, name).restype = res
        L.orc_stab_get_active.argtypes = [ctypes.c_void_p, _i32p]
        L.orc_stab_get_log.argtypes = [ctypes.c_void_p, _i32p, _i64p]
        L.orc_stab_get_unique_recs.argtypes = [ctypes.c_void_p, _i32p]
        L.orc_stab_get_unique_frames.argtypes = [ctypes.c_void_p, ctypes.c_int32, _i32p]
        L.orc_stab_get_unique_crop.argtypes = [ctypes.c_void_p, ctypes.c_int32, _u8p]
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


def label4(img):
    """scipy.ndimage.label(img) with the default structure -- labeler.py:126."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    labels = np.empty((h, w), dtype=np.int32)
    n = lib().orc_label4(_p(img, _u8p), w, h, _p(labels, _i32p))
    if n < 0:
        raise MemoryError
    return labels, n


def age_boundaries(labels, ages, n):
    """CC_AgeBoundaries -- accessmath_lib.c:357-413. Returns mins_y, maxs_y, mins_x, maxs_x, counts, ages."""
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    h, w = labels.shape
    outs = [np.zeros(n, dtype=np.int32) for _ in range(5)]
    oa = np.zeros(n, dtype=np.float32)
    ap = _p(np.ascontiguousarray(ages, dtype=np.float32), _f32p) if ages is not None else None
    lib().orc_age_boundaries(_p(labels, _i32p), ap, w, h, n, *[_p(o, _i32p) for o in outs], _p(oa, _f32p))
    return (*outs, oa)


def extract(labels, n, min_pixels=MIN_CC_PIXELS):
    """The kept-CC loop of Labeler.extractSpatioTemporalContent -- labeler.py:171-189.
    Returns (rec[k,6] = cc_id,min_x,max_x,min_y,max_y,size ; list of uint8 0/255 crops)."""
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    h, w = labels.shape
    if n == 0:
        return np.zeros((0, 6), np.int32), []
    nbytes = ctypes.c_int64(0)
    L = lib()
    kept = L.orc_extract(_p(labels, _i32p), w, h, n, min_pixels, None, None, None, ctypes.byref(nbytes))
    rec = np.zeros((max(kept, 1), 6), dtype=np.int32)
    off = np.zeros(max(kept, 1), dtype=np.int64)
    arena = np.zeros(max(nbytes.value, 1), dtype=np.uint8)
    L.orc_extract(_p(labels, _i32p), w, h, n, min_pixels, _p(rec, _i32p), _p(off, _i64p), _p(arena, _u8p),
                  ctypes.byref(nbytes))
    rec = rec[:kept]
    crops = []
    for k in range(kept):
        cw = rec[k, 2] - rec[k, 1] + 1
        ch = rec[k, 4] - rec[k, 3] + 1
        crops.append(arena[off[k]:off[k] + cw * ch].reshape(ch, cw).copy())
    return rec, crops


def extract_ccs(binary, min_pixels=MIN_CC_PIXELS):
    """Labeler.extractSpatioTemporalContent(binary, zeros) -- labeler.py:116-191 -- as plain arrays."""
    labels, n = label4(binary)
    rec, crops = extract(labels, n, min_pixels)
    return labels, n, rec, crops


def overlap(box_a, crop_a, box_b, crop_b):
    """match count of ConnectedComponent.getOverlapFMeasure -- connected_component.py:202-228.
    Boxes are (min_x, max_x, min_y, max_y)."""
    ba = np.asarray(box_a, dtype=np.int32)
    bb = np.asarray(box_b, dtype=np.int32)
    ca = np.ascontiguousarray(crop_a, dtype=np.uint8)
    cb = np.ascontiguousarray(crop_b, dtype=np.uint8)
    return int(lib().orc_overlap(_p(ba, _i32p), _p(ca, _u8p), _p(bb, _i32p), _p(cb, _u8p)))


def threshold_invert(logits, thr=128):
    """sigmoid -> *255 -> uint8 trunc -> >=thr -> 255-x  (FCN_lecturenet.py:452,461-467; FCN_lecturenet_binarizer.py:54)."""
    lg = np.ascontiguousarray(logits, dtype=np.float32)
    out = np.empty(lg.shape, dtype=np.uint8)
    lib().orc_threshold_invert(_p(lg, _f32p), lg.size, thr, _p(out, _u8p))
    return out


class Stability:
    """CCStabilityEstimator.__init__/add_frame state machine -- cc_stability_estimator.py:11-155."""

    def __init__(self, width, height, min_recall, min_precision, max_gap):
        self.width, self.height = width, height
        self._h = lib().orc_stab_new(width, height, min_recall, min_precision, max_gap)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_stab_free(self._h)
            self._h = None

    def add_frame(self, binary, min_pixels=MIN_CC_PIXELS):
        b = np.ascontiguousarray(binary, dtype=np.uint8)
        assert b.shape == (self.height, self.width)
        return lib().orc_stab_add_frame(self._h, _p(b, _u8p), min_pixels)

    @property
    def tempo_count(self):
        return int(lib().orc_stab_tempo_count(self._h))

    def active(self):
        n = lib().orc_stab_n_active(self._h)
        out = np.zeros(max(n, 1), np.int32)
        lib().orc_stab_get_active(self._h, _p(out, _i32p))
        return out[:n]

    def result(self, with_crops=True):
        """Plain-data view with the reference's attribute names:
        unique_recs[u] = (min_x,max_x,min_y,max_y,size), unique_cc_frames[u] = [(frame, raw_label)...],
        cc_idx_per_frame[f] = [(unique_idx, cc_id)...], unique_crops[u] = uint8 0/255 array."""
        L = lib()
        nu = L.orc_stab_n_unique(self._h)
        nf = L.orc_stab_n_frames(self._h)
        nl = L.orc_stab_n_log(self._h)
        rec = np.zeros((max(nu, 1), 6), np.int32)
        L.orc_stab_get_unique_recs(self._h, _p(rec, _i32p))
        rec = rec[:nu]
        frames = []
        crops = []
        for u in range(nu):
            fr = np.zeros((rec[u, 5], 2), np.int32)
            L.orc_stab_get_unique_frames(self._h, u, _p(fr, _i32p))
            frames.append([(int(a), int(b)) for a, b in fr])
            if with_crops:
                cw = rec[u, 1] - rec[u, 0] + 1
                ch = rec[u, 3] - rec[u, 2] + 1
                c = np.zeros((ch, cw), np.uint8)
                L.orc_stab_get_unique_crop(self._h, u, _p(c, _u8p))
                crops.append(c)
        log = np.zeros((max(nl, 1), 2), np.int32)
        foff = np.zeros(nf + 1, np.int64)
        L.orc_stab_get_log(self._h, _p(log, _i32p), _p(foff, _i64p))
        per_frame = [[(int(a), int(b)) for a, b in log[foff[f]:foff[f + 1]]] for f in range(nf)]
        return {
            "width": self.width, "height": self.height,
            "unique_recs": rec[:, :5].copy(),
            "unique_cc_frames": frames,
            "cc_idx_per_frame": per_frame,
            "unique_crops": crops,
            "tempo_count": self.tempo_count,
            "active": self.active(),
        }
