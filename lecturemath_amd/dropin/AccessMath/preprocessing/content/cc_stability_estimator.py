"""CCStabilityEstimator on the MI355X.

Same constructor, methods and attributes (= pickle schema) as the reference's content/cc_stability_estimator.py; the state
lives in HBM (lecturemath_amd.device.FrameStream) and the Python attributes the reference exposes -- unique_cc_objects,
unique_cc_frames, cc_idx_per_frame, cc_active, cc_last_frame, cc_int_index_x/y, tempo_count, img_idx -- are materialised
from the device on first access.  add_frame (:41-155) buffers frames and pushes them in batches; the step-03 methods
(:181-681) are served from ONE fused device run (lm_group_run) whose results are cached per parameter set.
"""
import numpy as np

from AM_CommonTools.data.connected_component import ConnectedComponent
from AccessMath.preprocessing.tools.interval_index import IntervalIndex
from lecturemath_amd import device, png

_LAZY = ("unique_cc_objects", "unique_cc_frames", "cc_idx_per_frame", "cc_int_index_x", "cc_int_index_y", "cc_last_frame",
         "cc_active", "tempo_count", "img_idx")


class CCStabilityEstimator:
    BATCH = 32
    INITIAL_FRAMES = 512

    def __init__(self, width, height, min_recall, min_precision, max_gap, verbose=False):
        self.width, self.height = width, height
        self.min_recall, self.min_precision, self.max_gap = min_recall, min_precision, max_gap
        self.verbose = verbose
        self.fake_age = np.zeros((height, width), dtype=np.float32)
        self._reset_runtime()

    # ------------------------------------------------------------------ runtime (not pickled)
    def _reset_runtime(self):
        self._stream = None
        self._pending = []
        self._pushed = 0
        self._host = None        # materialised python structures
        self._groups = {}        # (max_gap, min_times, t_window, min_recall, thr) -> device.Grouping
        self._split = None       # (max_gap, min_times) once split_stable_cc_by_gaps ran
        self._imported = None    # reference-format state loaded from a pickle, not yet on the device

    def _ensure_stream(self, need_frames):
        cap = getattr(self, "_cap_frames", 0)
        if self._stream is not None and need_frames <= cap:
            return
        new_cap = max(self.INITIAL_FRAMES, cap * 2, need_frames)
        st = self._stream.export_state() if self._stream is not None else None
        if self._stream is not None:
            self._stream.close()
        self._stream = device.FrameStream(self.width, self.height, new_cap, self.min_recall, self.min_precision, self.max_gap, 20,
                                          max_batch=self.BATCH)
        self._cap_frames = new_cap
        if st is not None:
            self._stream.import_state(st)
        elif self._imported is not None:
            self._stream.import_state(self._imported)
            self._pushed = len(self._imported["frame_off"]) - 1
            self._imported = None

    def _flush(self):
        if self._imported is not None:
            self._ensure_stream(len(self._imported["frame_off"]) - 1 + len(self._pending))
        if not self._pending:
            if self._stream is None:
                self._ensure_stream(1)
            return
        frames = np.stack(self._pending)
        self._pending = []
        self._ensure_stream(self._pushed + len(frames))
        self._stream.push(self._stream.be.from_host(frames))
        self._pushed += len(frames)
        self._host = None
        for g in self._groups.values():
            g.close()
        self._groups = {}

    # ------------------------------------------------------------------ step 02
    def add_frame(self, img, input_binary=False):
        if not input_binary:
            from .binarizer import Binarizer      # legacy classical binarizer of the reference tree (not on the v3.0 path)
            img = Binarizer.backgroundSubtractionBinarization(img.astype("uint8"))
        self._pending.append(np.ascontiguousarray(img, np.uint8))
        if len(self._pending) >= self.BATCH:
            self._flush()
        if self.verbose:
            print("[" + str(self._pushed + len(self._pending)) + "]", end="\r")

    def add_frames_device(self, frames):
        """Extension: frames already in HBM (uint8 [n,H,W]); no host round trip."""
        self._flush()
        self._ensure_stream(self._pushed + int(frames.shape[0]))
        self._stream.push(frames)
        self._pushed += int(frames.shape[0])
        self._host = None

    def finish_processing(self):
        self._flush()
        if self.verbose:
            print(".")
        print("Total CC merges tested: " + str(self.tempo_count))
        self.fake_age = None

    def get_raw_cc_count(self):
        return sum(len(f) for f in self.cc_idx_per_frame)

    # ------------------------------------------------------------------ materialisation of the reference's attributes
    def _materialise(self):
        if self._host is not None:
            return self._host
        self._flush()
        r = self._stream.read(with_crops=True)
        rec, foff = r["rec"], r["frame_off"]
        objs = []
        for c in range(r["n_cc"]):
            cc_id, mnx, mxx, mny, mxy, size = (np.int32(v) for v in rec[c, :6])
            nwords = ((int(mxx) >> 5) - (int(mnx) >> 5) + 1) * (int(mxy) - int(mny) + 1)
            o = int(r["crop_off"][c])
            cc = ConnectedComponent(int(cc_id), mnx, mxx, mny, mxy, size,
                                    device.decode_crop(r["crop"][o:o + nwords], int(mnx), int(mxx), int(mny), int(mxy)))
            cc.start_time = np.float32(0.0)
            cc.end_time = np.float32(0.0)
            objs.append(cc)
        assign = rec[:, 7] if len(rec) else np.zeros(0, np.int32)
        if self._split is not None:
            g = self._grouping()
            assign = g.array("assign")
            uniq_cc = g.array("uniq_cc")
            off, lst = g.array("ulist_off"), g.array("ulist_cc")
            frames = [[(int(rec[c, 6]), int(rec[c, 0]) + 1) for c in lst[off[u]:off[u + 1]]] for u in range(len(uniq_cc))]
            uobjs = [objs[c] for c in uniq_cc]
        else:
            nu = r["n_unique"]
            frames = [[] for _ in range(nu)]
            first = [None] * nu
            for c in range(len(rec)):
                u = int(assign[c])
                if first[u] is None:
                    first[u] = objs[c]
                frames[u].append((int(rec[c, 6]), int(rec[c, 0]) + 1))
            uobjs = first
        per_frame = [[(int(assign[c]), objs[c]) for c in range(foff[f], foff[f + 1])] for f in range(r["n_frames"])]
        last = [fl[-1][0] for fl in frames]
        ix, iy = IntervalIndex(True), IntervalIndex(True)
        for u in r["active"]:
            cc = uobjs[int(u)]
            ix.add(int(cc.min_x), int(cc.max_x) + 1, int(u))
            iy.add(int(cc.min_y), int(cc.max_y) + 1, int(u))
        self._host = {"unique_cc_objects": uobjs, "unique_cc_frames": frames, "cc_idx_per_frame": per_frame, "cc_int_index_x": ix,
                      "cc_int_index_y": iy, "cc_last_frame": last, "cc_active": [int(u) for u in r["active"]],
                      "tempo_count": r["tempo_count"], "img_idx": r["n_frames"]}
        return self._host

    def __getattr__(self, name):
        if name in _LAZY:
            return self._materialise()[name]
        raise AttributeError(name)

    # ------------------------------------------------------------------ pickling (reference schema, both directions)
    def __getstate__(self):
        h = self._materialise()
        d = {k: getattr(self, k) for k in ("width", "height", "min_recall", "min_precision", "max_gap", "fake_age", "verbose")}
        d.update(h)
        return d

    def __setstate__(self, d):
        for k in ("width", "height", "min_recall", "min_precision", "max_gap", "fake_age", "verbose"):
            setattr(self, k, d.get(k))
        self._reset_runtime()
        self._imported = _state_from_objects(d)

    # ------------------------------------------------------------------ step 03
    def rebuilt_binary_images(self):
        """(:166-179) kept for API completeness; its result is unused by the v3.0 scripts."""
        out = []
        for frame_ccs in self.cc_idx_per_frame:
            canvas = np.zeros((self.height, self.width), dtype=np.uint8)
            for _, cc in frame_ccs:
                canvas[cc.min_y:cc.max_y + 1, cc.min_x:cc.max_x + 1] += cc.img
            out.append(canvas)
        return out

    def _grouping(self, t_window=5, min_recall=0.5, thr=0.5):
        self._flush()
        max_gap, min_times = self._split if self._split is not None else (1 << 30, 3)
        key = (max_gap, min_times, t_window, min_recall, thr)
        if key not in self._groups:
            self._groups[key] = device.Grouping(self._stream, max_gap=max_gap, min_times=min_times, t_window=t_window,
                                                min_recall=min_recall, img_threshold=thr, reconstruct=True)
            self._last_key = key
        return self._groups[key]

    def split_stable_cc_by_gaps(self, max_gap, stable_min_frames):
        self._split = (int(max_gap), int(stable_min_frames))
        self._host = None
        return int(self._grouping().array("scalars")[0])

    def get_stable_cc_idxs(self, min_stable_frames):
        if self._split is not None and self._split[1] == min_stable_frames:
            return [int(v) for v in self._grouping().array("stable")]
        return [u for u, fl in enumerate(self.unique_cc_frames) if len(fl) >= min_stable_frames]

    def get_temporal_index(self):
        return [[u for u, _ in fr] for fr in self.cc_idx_per_frame]

    def _check_stable(self, stable_idxs, g):
        if list(stable_idxs) != [int(v) for v in g.array("stable")]:
            raise NotImplementedError("custom stable_idxs lists are not supported by the fused device path: pass "
                                      "get_stable_cc_idxs(min_times) after split_stable_cc_by_gaps(max_gap, min_times)")

    def compute_overlapping_stable_cc(self, stable_idxs, temporal_window):
        if self._split is None:
            self._split = (1 << 30, 3)
        g = self._grouping(t_window=int(temporal_window))
        self._tw = int(temporal_window)
        self._check_stable(stable_idxs, g)
        r = g.result(with_images=False, with_clean=False)
        return r["time_overlapping_cc"], r["total_intersections"], r["all_overlapping_cc"]

    def compute_groups(self, stable_idxs, overlapping_cc, min_recall, t_fmeasure, t_time_IOU):
        g = self._grouping(t_window=getattr(self, "_tw", 5), min_recall=float(min_recall))
        self._mr = float(min_recall)
        self._check_stable(stable_idxs, g)
        r = g.result(with_images=False, with_clean=False)
        return r["cc_groups"], r["group_idx_per_cc"]

    def _cur(self, thr=None):
        return self._grouping(t_window=getattr(self, "_tw", 5), min_recall=getattr(self, "_mr", 0.5),
                              thr=0.5 if thr is None else float(thr))

    def compute_groups_temporal_information(self, cc_groups):
        r = self._cur().result(with_images=False, with_clean=False)
        return r["group_ages"], r["groups_per_frame"]

    def compute_conflicting_groups(self, stable_idxs, all_overlapping_cc, n_groups, group_idx_per_cc):
        return self._cur().result(with_images=False, with_clean=False)["conflicts"]

    def compute_group_images(self, cc_groups, group_ages, segment_threshold):
        self._thr = float(segment_threshold)
        r = self._cur(self._thr).result(with_images=True, with_clean=False)
        return r["group_images"], r["group_boundaries"]

    def frames_from_groups(self, cc_groups, group_boundaries, groups_per_frame, group_ages, group_images, save_prefix=None,
                           stable_min_frames=3, show_unstable=True):
        g = self._cur(getattr(self, "_thr", 0.5))
        n = self.img_idx
        out = []
        for f0 in range(0, n, self.BATCH):
            m = min(self.BATCH, n - f0)
            for frame in g.be.to_host(g.render(f0, m)):
                out.append(png.encode_gray8(frame))
        return out

    def frames_from_groups_device(self, first, count):
        """Extension: reconstructed clean frames [first, first+count) as a device uint8 tensor (no PNG)."""
        return self._cur(getattr(self, "_thr", 0.5)).render(first, count)


def _state_from_objects(d):
    """Reference-format estimator attributes (after unpickling) -> lecturemath_amd.device.FrameStream.import_state input."""
    per_frame = d["cc_idx_per_frame"]
    recs, offs, crops, coff = [], [0], [], []
    words = 0
    for f, lst in enumerate(per_frame):
        for u, cc in lst:
            recs.append((int(cc.cc_id), int(cc.min_x), int(cc.max_x), int(cc.min_y), int(cc.max_y), int(cc.size), f, int(u)))
            wx0 = int(cc.min_x) >> 5
            nw = (int(cc.max_x) >> 5) - wx0 + 1
            h = int(cc.max_y) - int(cc.min_y) + 1
            bits = np.zeros((h, nw * 32), np.uint8)
            x0 = int(cc.min_x) - wx0 * 32
            bits[:, x0:x0 + cc.img.shape[1]] = cc.img > 0
            crops.append(np.packbits(bits, axis=1, bitorder="little").view("<u4").reshape(-1))
            coff.append(words)
            words += h * nw
        offs.append(len(recs))
    rec = np.asarray(recs, np.int32).reshape(-1, 8)
    frames = d["unique_cc_frames"]
    nu = len(frames)
    active = np.asarray(d.get("cc_active", []), np.int32)
    first = np.full(nu, -1, np.int64)
    if len(rec):
        order = np.arange(len(rec) - 1, -1, -1)
        first[rec[order, 7]] = order
    last = np.asarray([fl[-1][0] for fl in frames], np.int32) if nu else np.zeros(0, np.int32)
    return {"rec": rec, "frame_off": np.asarray(offs, np.int64), "crop_off": np.asarray(coff, np.int64),
            "crop": np.concatenate(crops) if crops else np.zeros(0, np.uint32), "n_unique": nu,
            "tempo_count": int(d.get("tempo_count", 0)), "active": active, "active_cc": first[active].astype(np.int32),
            "active_last": last[active].astype(np.int32)}
