"""Canonical SHA-256 digests of the products of steps 02 and 03, so that results of streams too long to ship as fixtures
(BASELINE configs[2]: 10,000 frames at 1080p) can be compared bit for bit between this library, the oracle and the
reference (tests/golden/make_golden_stream1080p.py runs the reference and commits the digests it prints).

Every product is serialised the same way whatever produced it: a ragged list becomes an int64 offset array followed by a
little-endian int32 data array; the digest is sha256(offsets || data).

  unique_cc_frames   per unique: (frame, raw label) entries         cc_stability_estimator.py:58,102,114 (after the split :181-228)
  cc_idx_per_frame   per frame: (unique index, cc_id) entries       :59,103,115
  cc_groups          per group: member unique indices, in order     :308-413
  group_ages         per group: sorted ages                         :415-444
  groups_per_frame   per frame: live groups                         :415-444
  group_boundaries   per group: (min_x, max_x, min_y, max_y)        :575-636
  clean_binary       reconstructed frames, channel 0, in order      :638-681
"""
import hashlib

import numpy as np


def ragged_digest(offsets, data):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(offsets, dtype="<i8").tobytes())
    h.update(np.ascontiguousarray(data, dtype="<i4").tobytes())
    return h.hexdigest()


def lists_digest(list_of_lists, width):
    """Python lists of ints (width 1) or of width-tuples, e.g. the reference's own attributes."""
    off = np.zeros(len(list_of_lists) + 1, np.int64)
    off[1:] = np.cumsum([len(x) for x in list_of_lists])
    flat = [v for lst in list_of_lists for v in lst]
    data = np.asarray(flat, np.int64).reshape(-1, width) if flat else np.zeros((0, width), np.int64)
    return ragged_digest(off, data)


def sums_digest(sums):
    """sha256 over the per-frame byte sums (int64) of the reconstructed frames: a cheap stand-in for `clean_binary` where hashing
    20 GB of frames on the host is too slow (bench.py); the sums come from lm_frame_sums on the device."""
    return hashlib.sha256(np.ascontiguousarray(sums, dtype="<i8").tobytes()).hexdigest()


class FrameChain:
    """sha256 over uint8 frames fed in order (any chunking)."""

    def __init__(self):
        self.h = hashlib.sha256()
        self.n = 0

    def update(self, frames):
        a = np.ascontiguousarray(frames, dtype=np.uint8)
        self.h.update(a.tobytes())
        self.n += 1 if a.ndim == 2 else a.shape[0]

    def hexdigest(self):
        return self.h.hexdigest()


def from_python(unique_cc_frames, cc_idx_per_frame, cc_groups, group_ages, groups_per_frame, group_boundaries):
    """Digests of reference-shaped Python objects (the reference's estimator attributes / the oracle's results).
    cc_idx_per_frame entries are (unique index, cc_id) pairs; group_ages / group_boundaries are dicts keyed 0..n_groups-1."""
    ng = len(cc_groups)
    return {
        "unique_cc_frames": lists_digest(unique_cc_frames, 2),
        "cc_idx_per_frame": lists_digest(cc_idx_per_frame, 2),
        "cc_groups": lists_digest(cc_groups, 1),
        "group_ages": lists_digest([group_ages[g] for g in range(ng)], 1),
        "groups_per_frame": lists_digest(groups_per_frame, 1),
        "group_boundaries": lists_digest([[tuple(int(v) for v in group_boundaries[g])] for g in range(ng)], 4),
    }


def from_device(stream, grouping):
    """Digests of a finished device.FrameStream + device.Grouping, built from the flat arrays (no Python lists: a 10k-frame
    stream holds ~10^7 CC records)."""
    A = {k: grouping.array(k) for k in ("ulist_off", "ulist_cc", "assign", "grp_off", "grp_members", "ages_off", "ages", "gpf_off",
                                        "gpf", "bounds", "scalars")}
    r = stream.read(with_crops=False)
    rec, foff = r["rec"], r["frame_off"]
    ucc = A["ulist_cc"]
    ucf = np.stack([rec[ucc, 6], rec[ucc, 0] + 1], axis=1) if len(ucc) else np.zeros((0, 2), np.int32)
    cipf = np.stack([A["assign"][:len(rec)], rec[:, 0]], axis=1) if len(rec) else np.zeros((0, 2), np.int32)
    ng = int(A["scalars"][2])
    return {
        "unique_cc_frames": ragged_digest(A["ulist_off"], ucf),
        "cc_idx_per_frame": ragged_digest(foff, cipf),
        "cc_groups": ragged_digest(A["grp_off"], A["grp_members"]),
        "group_ages": ragged_digest(A["ages_off"], A["ages"]),
        "groups_per_frame": ragged_digest(A["gpf_off"], A["gpf"]),
        "group_boundaries": ragged_digest(np.arange(ng + 1, dtype=np.int64), A["bounds"].reshape(-1, 4)),
    }
