"""Thin Python layer over the C ABI: device buffers (torch on the GPU), contexts and frame streams.

PyTorch is plumbing only (device memory + streams); all arithmetic happens in liblecturemath_hip.so.
With the test-only emulated library (tests/hipemu) "device" memory is host memory and numpy arrays
stand in for torch tensors; the product never selects that path by itself (see _lib.load).
"""
import ctypes

import numpy as np

from . import _lib


class Backend:
    """Allocates buffers the library can address: CUDA(HIP) tensors, or numpy for the emulated build."""

    def __init__(self, lib):
        self.lib = lib
        self.device = lib.is_device_build
        if self.device:
            import torch
            if not torch.cuda.is_available():
                raise _lib.LecturemathLibraryError("liblecturemath_hip.so needs a GPU (torch.cuda.is_available() is False)")
            self.torch = torch

    _T = {np.uint8: "uint8", np.int32: "int32", np.float32: "float32", np.int64: "int64", np.int16: "int16"}

    def empty(self, shape, dtype):
        if self.device:
            return self.torch.empty(shape, dtype=getattr(self.torch, self._T[dtype]), device="cuda")
        # emulated build: 64-byte aligned like device allocations (packed record blocks hold 32-byte aligned structs)
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        raw = np.empty(nbytes + 64, np.uint8)
        off = (-raw.ctypes.data) % 64
        return raw[off:off + nbytes].view(dtype).reshape(shape)

    def from_host(self, a):
        a = np.ascontiguousarray(a)
        if self.device:
            return self.torch.from_numpy(a).cuda()
        return a.copy()

    def to_host(self, x):
        if self.device:
            return x.cpu().numpy()
        return np.asarray(x)

    def stream(self):
        if self.device:
            return self.torch.cuda.current_stream().cuda_stream
        return None

    def synchronize(self):
        if self.device:
            self.torch.cuda.synchronize()


class FrameLabeler:
    """Batched scipy.ndimage.label + CC_AgeBoundaries on device frames (labeler.py:126, accessmath_lib.c:357-413)."""

    def __init__(self, width, height, max_batch=1, lib=None):
        self.lib = lib or _lib.load()
        self.be = Backend(self.lib)
        self.width, self.height, self.max_batch = width, height, max_batch
        self.ctx = self.lib.lm_ctx_create(width, height, max_batch)
        if not self.ctx:
            raise _lib.LecturemathError(_lib.LM_ERR_HIP, self.lib.last_error())

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.lm_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def threshold_invert(self, logits, thr=128):
        """logits: device fp32 array -> device uint8 {0,255}, already inverted (255 = ink)."""
        out = self.be.empty(tuple(logits.shape), np.uint8)
        n = int(np.prod(logits.shape))
        self.lib.check(self.lib.lm_threshold_invert(_lib.ptr(logits), _lib.ptr(out), n, thr, self.be.stream()))
        return out

    def label(self, binary, want_labels=True):
        """binary: device uint8 [B,H,W] -> (device int32 labels [B,H,W] or None, host int32 counts [B])."""
        b = binary.shape[0]
        assert tuple(binary.shape[1:]) == (self.height, self.width) and b <= self.max_batch
        labels = self.be.empty((b, self.height, self.width), np.int32) if want_labels else None
        self.lib.check(self.lib.lm_label_batch(self.ctx, _lib.ptr(binary), b, _lib.ptr(labels), self.be.stream()))
        counts = np.zeros(b, np.int32)
        self.lib.check(self.lib.lm_label_counts(self.ctx, counts.ctypes.data, self.be.stream()))
        return labels, counts

    def label_logits(self, logits, thr=128, invert=True, want_binary=False, want_labels=True):
        """fp32 logits [B,H,W] -> (labels or None, counts, {0,255} frames or None): threshold (+ the worker's inversion) and labelling in
        one pass over the logits (lm_label_batch_logits)."""
        b = logits.shape[0]
        assert tuple(logits.shape[1:]) == (self.height, self.width) and b <= self.max_batch
        labels = self.be.empty((b, self.height, self.width), np.int32) if want_labels else None
        binary = self.be.empty((b, self.height, self.width), np.uint8) if want_binary else None
        self.lib.check(self.lib.lm_label_batch_logits(self.ctx, _lib.ptr(logits), b, thr, 1 if invert else 0, _lib.ptr(binary), _lib.ptr(labels),
                                                      self.be.stream()))
        counts = np.zeros(b, np.int32)
        self.lib.check(self.lib.lm_label_counts(self.ctx, counts.ctypes.data, self.be.stream()))
        return labels, counts, binary

    def stats(self, counts):
        """CC_AgeBoundaries arrays of every frame of the last batch: list of int32 [5, n] (mins_y, maxs_y, mins_x, maxs_x, counts)."""
        self.lib.check(self.lib.lm_cc_stats_batch(self.ctx, self.be.stream()))
        out = []
        for f, n in enumerate(counts):
            a = np.zeros((5, int(n)), np.int32)
            rows = [a[i].ctypes.data if n else None for i in range(5)]
            self.lib.check(self.lib.lm_cc_stats_read(self.ctx, f, int(n), *rows, self.be.stream()))
            out.append(a)
        return out


def frame_sums(frames, lib=None):
    """Exact per-frame sums of a device uint8 tensor [n, H, W] (step 04, VideoSegmenter.compute_binary_sums): int64 numpy."""
    lib = lib or _lib.load()
    be = Backend(lib)
    n = int(frames.shape[0])
    out = be.empty((n,), np.int64)
    lib.check(lib.lm_frame_sums(_lib.ptr(frames), n, int(frames.shape[1]) * int(frames.shape[2]), _lib.ptr(out), be.stream()))
    return be.to_host(out)


def image_pairs_overlap(boxes, images, lib=None):
    """Pairs (i < j), sorted, of {0,255} images placed at their boxes (min_x, max_x, min_y, max_y inclusive) that share an ink
    pixel (step 05: compute_overlapping_CC_groups / keyframe conflicts).  Box join + bit tests on the device."""
    lib = lib or _lib.load()
    n = len(images)
    if n < 2:
        return []
    hb = np.ascontiguousarray(np.asarray(boxes, np.int32).reshape(n, 4))
    off = np.zeros(n + 1, np.int64)
    off[1:] = np.cumsum([im.size for im in images])
    flat = np.ascontiguousarray(np.concatenate([np.asarray(im, np.uint8).ravel() for im in images]))
    be = Backend(lib)
    cap = max(1024, 8 * n)
    while True:
        pairs = np.zeros((cap, 2), np.int32)
        found = np.zeros(1, np.int64)
        rc = lib.lm_image_pairs_overlap(hb.ctypes.data, flat.ctypes.data, off.ctypes.data, n, pairs.ctypes.data, cap, found.ctypes.data, be.stream())
        if rc == _lib.LM_ERR_CAPACITY and int(found[0]) > cap:
            cap = int(found[0])
            continue
        lib.check(rc)
        return [(int(a), int(b)) for a, b in pairs[:int(found[0])]]


def decode_crop(words, min_x, max_x, min_y, max_y):
    """bit-row crop (absolute 32-px column alignment) -> uint8 0/255 (h, w) like ConnectedComponent.img."""
    wx0 = min_x >> 5
    nw = (max_x >> 5) - wx0 + 1
    h = max_y - min_y + 1
    w = np.asarray(words, dtype="<u4").reshape(h, nw)
    bits = np.unpackbits(w.view(np.uint8).reshape(h, nw * 4), axis=1, bitorder="little")
    x0 = min_x - wx0 * 32
    return (bits[:, x0:x0 + (max_x - min_x + 1)] * 255).astype(np.uint8)


class FrameStream:
    """Device-resident CCStabilityEstimator.add_frame state (cc_stability_estimator.py:11-155)."""

    def __init__(self, width, height, max_frames, min_recall=0.85, min_precision=0.85, max_gap=85, min_pixels=20,
                 max_batch=16, max_ccs=None, max_crop_words=None, max_uniques=None, lib=None):
        self.labeler = FrameLabeler(width, height, max_batch, lib)
        self.lib, self.be = self.labeler.lib, self.labeler.be
        self.width, self.height = width, height
        self.min_pixels = min_pixels
        self._create_args = None
        px = width * height
        max_ccs = max_ccs or max_frames * max(px // 256, 64)
        max_crop_words = max_crop_words or max_frames * max(px // 4, 4096)
        max_uniques = max_uniques or max_ccs
        self.handle = self.lib.lm_stream_create(self.labeler.ctx, max_frames, max_ccs, max_crop_words, max_uniques,
                                                min_recall, min_precision, max_gap, min_pixels)
        if not self.handle:
            raise _lib.LecturemathError(_lib.LM_ERR_HIP, self.lib.last_error())

    def close(self):
        if getattr(self, "handle", None):
            self.lib.lm_stream_destroy(self.handle)
            self.handle = None
        self.labeler.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self.lib.check(self.lib.lm_stream_reset(self.handle, self.be.stream()))

    def set_min_pixels(self, min_pixels):
        self.lib.check(self.lib.lm_stream_set_min_pixels(self.handle, int(min_pixels)))
        self.min_pixels = int(min_pixels)

    def push(self, binary, labels_out=None):
        """binary: device uint8 [n,H,W] (any n; split into batches internally). Asynchronous."""
        self.lib.check(self.lib.lm_stream_push(self.handle, _lib.ptr(binary), int(binary.shape[0]), _lib.ptr(labels_out),
                                               self.be.stream()))

    def push_records(self, binary, labels_out=None):
        """Per-frame half only (label, stats, records, crops): no temporal matching."""
        self.lib.check(self.lib.lm_stream_push_records(self.handle, _lib.ptr(binary), int(binary.shape[0]), _lib.ptr(labels_out),
                                                       self.be.stream()))

    def match(self, n_frames):
        """Sequential temporal matching over the next n_frames unmatched frames."""
        self.lib.check(self.lib.lm_stream_match(self.handle, int(n_frames), self.be.stream()))

    def pack(self, first_frame, n_frames):
        """Records + crops of frames [first_frame, first_frame + n_frames) as one flat device uint8 buffer (lm_stream_pack)."""
        nb = ctypes.c_int64(0)
        self.lib.check(self.lib.lm_stream_pack_size(self.handle, int(first_frame), int(n_frames), ctypes.addressof(nb), self.be.stream()))
        buf = self.be.empty((max(int(nb.value), 32),), np.uint8)
        self.lib.check(self.lib.lm_stream_pack(self.handle, int(first_frame), int(n_frames), _lib.ptr(buf), int(nb.value), self.be.stream()))
        return buf

    def append_packed(self, buf):
        """Append a packed block (device uint8 buffer from pack(), possibly received from another rank) behind the last frame."""
        self.lib.check(self.lib.lm_stream_append_packed(self.handle, _lib.ptr(buf), int(buf.shape[0]), self.be.stream()))

    def export_assign(self):
        """What the temporal matching added to the records, as one flat device uint8 buffer (lm_stream_export_assign): the hand-off of
        a matched stream to the rank that runs step 03, together with pack(0, n_frames)."""
        nb = ctypes.c_int64(0)
        self.lib.check(self.lib.lm_stream_assign_bytes(self.handle, ctypes.addressof(nb), self.be.stream()))
        buf = self.be.empty((int(nb.value),), np.uint8)
        self.lib.check(self.lib.lm_stream_export_assign(self.handle, _lib.ptr(buf), int(nb.value), self.be.stream()))
        return buf

    def import_assign(self, buf):
        self.lib.check(self.lib.lm_stream_import_assign(self.handle, _lib.ptr(buf), int(buf.shape[0]), self.be.stream()))

    def counters(self):
        k = np.zeros(7, np.int64)
        self.lib.check(self.lib.lm_stream_counters(self.handle, k.ctypes.data, self.be.stream()))
        return dict(n_frames=int(k[0]), n_cc=int(k[1]), n_crop_words=int(k[2]), n_unique=int(k[3]), n_active=int(k[4]),
                    tempo_count=int(k[5]))

    def read(self, with_crops=True):
        """Host copy of the stream: records, frame offsets, crops, active list."""
        k = self.counters()
        rec = np.zeros((max(k["n_cc"], 1), 8), np.int32)
        foff = np.zeros(k["n_frames"] + 1, np.int64)
        coff = np.zeros(max(k["n_cc"], 1), np.int64)
        crop = np.zeros(max(k["n_crop_words"], 1), np.uint32) if with_crops else None
        active = np.zeros(max(k["n_active"], 1), np.int32)
        self.lib.check(self.lib.lm_stream_read(self.handle, rec.ctypes.data, foff.ctypes.data, coff.ctypes.data,
                                               crop.ctypes.data if with_crops else None, active.ctypes.data, self.be.stream()))
        return dict(k, rec=rec[:k["n_cc"]], frame_off=foff, crop_off=coff[:k["n_cc"]], crop=crop, active=active[:k["n_active"]])

    def export_state(self):
        """Host snapshot (numpy) of everything needed to rebuild the stream: records, crops, active list."""
        r = self.read(with_crops=True)
        rec = r["rec"]
        nu = r["n_unique"]
        first = np.full(nu, -1, np.int64)
        last = np.zeros(nu, np.int32)
        if len(rec):
            order = np.arange(len(rec) - 1, -1, -1)
            first[rec[order, 7]] = order                  # smallest CC index wins (written last)
            np.maximum.at(last, rec[:, 7], rec[:, 6])
        act = r["active"].astype(np.int32)
        return {"width": self.width, "height": self.height, "rec": rec, "frame_off": r["frame_off"], "crop_off": r["crop_off"],
                "crop": r["crop"][:r["n_crop_words"]], "n_unique": nu, "tempo_count": r["tempo_count"], "active": act,
                "active_cc": first[act].astype(np.int32), "active_last": last[act].astype(np.int32), "uniq_first_cc": first}

    def import_state(self, st):
        """Inverse of export_state (capacities of this stream must suffice)."""
        rec = np.ascontiguousarray(st["rec"], np.int32)
        foff = np.ascontiguousarray(st["frame_off"], np.int64)
        coff = np.ascontiguousarray(st["crop_off"], np.int64)
        crop = np.ascontiguousarray(st["crop"], np.uint32)
        act = np.ascontiguousarray(st["active"], np.int32)
        acc = np.ascontiguousarray(st["active_cc"], np.int32)
        acl = np.ascontiguousarray(st["active_last"], np.int32)
        self.lib.check(self.lib.lm_stream_import(
            self.handle, rec.ctypes.data if len(rec) else None, foff.ctypes.data, coff.ctypes.data if len(rec) else None,
            crop.ctypes.data if len(crop) else (np.zeros(1, np.uint32).ctypes.data if len(rec) else None), len(foff) - 1, len(rec),
            len(crop), int(st["n_unique"]), int(st["tempo_count"]), act.ctypes.data if len(act) else None,
            acc.ctypes.data if len(act) else None, acl.ctypes.data if len(act) else None, len(act),
            int(st.get("n_matched", len(foff) - 1)), self.be.stream()))

    def result(self, with_crops=True):
        """Same plain-data view the oracle produces (reference attribute names):
        unique_recs, unique_crops, unique_cc_frames, cc_idx_per_frame, tempo_count, active."""
        r = self.read(with_crops)
        rec, foff = r["rec"], r["frame_off"]
        nu = r["n_unique"]
        first = np.full(nu, -1, np.int64)
        frames = [[] for _ in range(nu)]
        per_frame = []
        for f in range(r["n_frames"]):
            lst = []
            for c in range(foff[f], foff[f + 1]):
                u = int(rec[c, 7])
                if first[u] < 0:
                    first[u] = c
                frames[u].append((f, int(rec[c, 0]) + 1))
                lst.append((u, int(rec[c, 0])))
            per_frame.append(lst)
        urec = rec[first][:, [1, 2, 3, 4, 5]].copy() if nu else np.zeros((0, 5), np.int32)
        crops = []
        if with_crops:
            for u in range(nu):
                c = first[u]
                mnx, mxx, mny, mxy = (int(v) for v in rec[c, 1:5])
                nwords = ((mxx >> 5) - (mnx >> 5) + 1) * (mxy - mny + 1)
                o = int(r["crop_off"][c])
                crops.append(decode_crop(r["crop"][o:o + nwords], mnx, mxx, mny, mxy))
        return {"width": self.width, "height": self.height, "unique_recs": urec, "unique_crops": crops,
                "unique_cc_frames": frames, "cc_idx_per_frame": per_frame, "tempo_count": r["tempo_count"],
                "active": r["active"], "raw": r}


_G_DTYPES = {0: np.int32, 1: np.int64, 2: np.int32, 3: np.int32, 4: np.int32, 5: np.int32, 6: np.int32, 7: np.int32,
             8: np.int64, 9: np.int32, 10: np.float64, 11: np.float64, 12: np.int64, 13: np.int32, 14: np.int32, 15: np.int32,
             16: np.int32, 17: np.int64, 18: np.int32, 19: np.int32, 20: np.int64, 21: np.int32, 22: np.int64, 23: np.int32,
             24: np.int32, 25: np.int32, 26: np.int64, 27: np.int64, 28: np.int64, 29: np.float64, 30: np.int32, 31: np.int64,
             32: np.int64, 33: np.uint8, 34: np.int64}
_G_NAMES = ["uniq_cc", "ulist_off", "ulist_cc", "assign", "stable", "pair_a", "pair_b", "pair_match", "tov_off", "tov_other",
            "tov_recall", "tov_precision", "aov_off", "aov_other", "aov_matched", "aov_size_other", "aov_size_self", "grp_off",
            "grp_members", "gid", "ages_off", "ages", "gpf_off", "gpf", "conf_g1", "conf_g2", "conf_matched", "conf_unmatched",
            "conf_union", "conf_inter", "bounds", "gimg_off", "gimg_item_off", "gimg", "scalars"]


class Grouping:
    """Step 03 over a finished FrameStream (pre_ST3D_v3.0_03_cc_grouping.py:22-118)."""

    def __init__(self, stream, max_gap=85, min_times=3, t_window=5, min_recall=0.5, img_threshold=0.5, reconstruct=True):
        self.stream = stream
        self.lib, self.be = stream.lib, stream.be
        self.handle = self.lib.lm_group_run(stream.handle, max_gap, min_times, t_window, min_recall, img_threshold,
                                            1 if reconstruct else 0, self.be.stream())
        if not self.handle:
            raise _lib.LecturemathError(_lib.LM_ERR_HIP, self.lib.last_error())

    def close(self):
        if getattr(self, "handle", None):
            self.lib.lm_group_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def array(self, which):
        """Copy of one result array (see the id table in include/lecturemath_amd.h)."""
        if isinstance(which, str):
            which = _G_NAMES.index(which)
        p, n = ctypes.c_void_p(), ctypes.c_int64()
        self.lib.check(self.lib.lm_group_array(self.handle, which, ctypes.addressof(p), ctypes.addressof(n)))
        dt = np.dtype(_G_DTYPES[which])
        if n.value == 0:
            return np.zeros(0, dt)
        buf = (ctypes.c_char * (n.value * dt.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=dt).copy()

    def render(self, first, n, out=None):
        """frames_from_groups channel 0 for frames [first, first+n) -> device uint8 [n,H,W]."""
        out = out if out is not None else self.be.empty((n, self.stream.height, self.stream.width), np.uint8)
        self.lib.check(self.lib.lm_group_render(self.handle, first, n, _lib.ptr(out), self.be.stream()))
        return out

    def result(self, with_images=True, with_clean=True):
        """Plain-data view with the reference's names (same shape as oracle.grouping.run_step03)."""
        A = {name: self.array(i) for i, name in enumerate(_G_NAMES) if name != "gimg" or with_images}
        sc = A["scalars"]
        nu, ng, nf = int(sc[3]), int(sc[2]), int(sc[4])

        def csr(off, *cols):
            return [[tuple(c[j].item() for c in cols) if len(cols) > 1 else cols[0][j].item()
                     for j in range(off[i], off[i + 1])] for i in range(len(off) - 1)]

        out = {
            "n_split": int(sc[0]), "total_intersections": int(sc[1]),
            "stable_idxs": [int(v) for v in A["stable"]],
            "time_overlapping_cc": csr(A["tov_off"], A["tov_other"], A["tov_recall"], A["tov_precision"]),
            "all_overlapping_cc": csr(A["aov_off"], A["aov_other"], A["aov_matched"], A["aov_size_other"], A["aov_size_self"]),
            "cc_groups": csr(A["grp_off"], A["grp_members"]),
            "group_idx_per_cc": {u: int(g) for u, g in enumerate(A["gid"]) if g >= 0},
            "group_ages": {g: lst for g, lst in enumerate(csr(A["ages_off"], A["ages"]))},
            "groups_per_frame": csr(A["gpf_off"], A["gpf"]),
            "group_boundaries": {g: tuple(int(v) for v in A["bounds"][4 * g:4 * g + 4]) for g in range(ng)},
            "arrays": A,
        }
        conf = {g: {} for g in range(ng)}
        for i in range(len(A["conf_g1"])):
            conf[int(A["conf_g1"][i])][int(A["conf_g2"][i])] = {
                "matched": int(A["conf_matched"][i]), "unmatched": int(A["conf_unmatched"][i]),
                "area_union": int(A["conf_union"][i]), "area_intersection": float(A["conf_inter"][i])}
        out["conflicts"] = conf
        if with_images:
            imgs = {}
            for g in range(ng):
                x0, x1, y0, y1 = out["group_boundaries"][g]
                w, h = x1 - x0 + 1, y1 - y0 + 1
                lst = []
                for it in range(A["gimg_item_off"][g], A["gimg_item_off"][g + 1]):
                    o = A["gimg_off"][it]
                    lst.append(A["gimg"][o:o + w * h].reshape(h, w))
                imgs[g] = lst
            out["group_images"] = imgs
        if with_clean and nf:
            out["clean_binary"] = list(self.be.to_host(self.render(0, nf)))
        # unique_cc_frames / cc_idx_per_frame after the split
        rec = self.stream.read(with_crops=False)["rec"]
        out["unique_cc_frames"] = [[(int(rec[c, 6]), int(rec[c, 0]) + 1) for c in A["ulist_cc"][A["ulist_off"][u]:A["ulist_off"][u + 1]]]
                                   for u in range(nu)]
        foff = self.stream.read(with_crops=False)["frame_off"]
        out["cc_idx_per_frame"] = [[(int(A["assign"][c]), int(rec[c, 0])) for c in range(foff[f], foff[f + 1])] for f in range(nf)]
        return out
