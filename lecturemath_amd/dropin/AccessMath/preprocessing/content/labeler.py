"""Labeler on the MI355X: same entry points as the reference's content/labeler.py (extractConnectedComponents :107-111,
extractSpatioTemporalContent :116-191, MIN_CC_PIXELS :22).  scipy.ndimage.label, CC_AgeBoundaries and the per-CC crop loop
are replaced by one pass through liblecturemath_hip.so (bit-packed run labelling, lm_cc_kernels.hip)."""
import numpy as np

from AM_CommonTools.data.connected_component import ConnectedComponent
from lecturemath_amd import _lib, device


class Labeler:
    MIN_CC_PIXELS = 20
    _streams = {}

    @staticmethod
    def _one_frame_stream(width, height):
        key = (width, height)
        fs = Labeler._streams.get(key)
        if fs is None:
            fs = device.FrameStream(width, height, 1, 2.0, 2.0, 1, Labeler.MIN_CC_PIXELS, max_batch=1,
                                    max_ccs=max(width * height // 2, 64), max_crop_words=max(width * height, 4096))
            Labeler._streams = {key: fs}       # keep one workspace
        return fs

    @staticmethod
    def extractConnectedComponents(content, filter_small=True, is_labeled=False):
        fake_age = np.zeros(content.shape, dtype=np.float32)
        return Labeler.extractSpatioTemporalContent(content, fake_age, filter_small, is_labeled)

    @staticmethod
    def extractSpatioTemporalContent(content, ages, filter_small=True, is_labeled=False):
        assert len(content.shape) == 2
        height, width = content.shape
        if is_labeled:
            return Labeler._from_labels(content, ages, filter_small)
        fs = Labeler._one_frame_stream(width, height)
        min_px = Labeler.MIN_CC_PIXELS if filter_small else 1
        if fs.min_pixels != min_px:
            fs.set_min_pixels(min_px)
        fs.reset()
        fs.push(fs.be.from_host(np.ascontiguousarray(content, np.uint8)[None]))
        r = fs.read(with_crops=True)
        age0 = None
        out = []
        for c in range(r["n_cc"]):
            cc_id, mnx, mxx, mny, mxy, size = (np.int32(v) for v in r["rec"][c, :6])
            nwords = ((int(mxx) >> 5) - (int(mnx) >> 5) + 1) * (int(mxy) - int(mny) + 1)
            o = int(r["crop_off"][c])
            img = device.decode_crop(r["crop"][o:o + nwords], int(mnx), int(mxx), int(mny), int(mxy))
            cc = ConnectedComponent(int(cc_id), mnx, mxx, mny, mxy, size, img)
            # minimum age over the CC's pixels (accessmath_lib.c:405-407); all-zero on the v3.0 path
            if ages is not None and np.any(ages):
                age = np.float32(ages[int(mny):int(mxy) + 1, int(mnx):int(mxx) + 1][img > 0].min())
            else:
                age = np.float32(0.0)
            cc.start_time = age
            cc.end_time = age
            out.append(cc)
        return out

    @staticmethod
    def _from_labels(labels, ages, filter_small):
        """is_labeled=True callers hand in an int32 label image (labeler.py:127-130): statistics through the literal
        CC_AgeBoundaries export of the library."""
        lib = _lib.load()
        labels = np.ascontiguousarray(labels, np.int32)
        n = int(labels.max())
        if n == 0:
            return []
        h, w = labels.shape
        ages = np.ascontiguousarray(ages, np.float32)
        o = [np.zeros(n, np.int32) for _ in range(5)]
        oa = np.zeros(n, np.float32)
        lib.CC_AgeBoundaries(labels.ctypes.data, ages.ctypes.data, w, h, n, *[a.ctypes.data for a in o], oa.ctypes.data)
        mny, mxy, mnx, mxx, cnt = o
        out = []
        for k in range(n):
            if filter_small and cnt[k] < Labeler.MIN_CC_PIXELS:
                continue
            img = (labels[mny[k]:mxy[k] + 1, mnx[k]:mxx[k] + 1] == k + 1).astype(np.uint8) * 255
            cc = ConnectedComponent(k, mnx[k], mxx[k], mny[k], mxy[k], cnt[k], img)
            cc.start_time = oa[k]
            cc.end_time = oa[k]
            out.append(cc)
        return out
