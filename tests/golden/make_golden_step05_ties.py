#!/usr/bin/env python3
"""G8b: tie-heavy inputs for step 05 (KeyframeExtractor.GenerateFromST3DForIntervals, keyframe_extractor.py:13-145), run through
THE REFERENCE in this container.  Synthetic SpaceTimeStructs in which many groups START AT THE SAME FRAME and overlap each other
in chains and cliques, with hundreds of groups per segment -- so that which group of a conflict set is drawn is decided by the
reference's tie-break (position inside `list(set)` of the component, CPython set order for ints beyond the table size).
Stored: the structure (ages, boxes, images) and the reference's keyframes / times for several segmentations."""
import contextlib
import io
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_env  # noqa: E402

ref_env.enter()
if not hasattr(np, "bool"):
    np.bool = bool          # numpy >= 1.24 lacks np.bool, which the reference still uses (:69)
from AccessMath.preprocessing.content.keyframe_extractor import KeyframeExtractor  # noqa: E402
from AccessMath.data.space_time_struct import SpaceTimeStruct  # noqa: E402


def build(seed, n_groups, h=96, w=640, n_frames=40):
    rng = np.random.default_rng(seed)
    ages, images, bounds = {}, {}, {}
    for g in range(n_groups):
        a0 = int(rng.choice([0, 0, 0, 5, 5, 12]))                   # few distinct starts: ties everywhere
        a1 = int(rng.integers(a0 + 1, n_frames))
        mids = sorted(set(int(v) for v in rng.integers(a0 + 1, a1 + 1, int(rng.integers(0, 3)))) - {a0, a1})
        ag = [a0] + mids + [a1]
        gw, gh = int(rng.integers(6, 30)), int(rng.integers(6, 20))
        x0 = int(rng.integers(0, w - gw)) if g % 3 else int((g * 7) % (w - gw))     # every third group on a diagonal of neighbours
        y0 = int(rng.integers(0, h - gh))
        imgs = []
        for _ in range(len(ag) - 1):
            im = ((rng.random((gh, gw)) < 0.55) * 255).astype(np.uint8)
            imgs.append(im)
        ages[g], images[g], bounds[g] = ag, imgs, (x0, x0 + gw - 1, y0, y0 + gh - 1)
    return ages, images, bounds, h, w, n_frames


def make():
    out = {}
    specs = [(1, 120), (2, 400), (3, 700)]
    out["n_cases"] = np.int64(len(specs))
    for c, (seed, n_groups) in enumerate(specs):
        ages, images, bounds, h, w, n = build(seed, n_groups)
        st3d = SpaceTimeStruct([1000.0 * i for i in range(n)], list(range(n)), h, w, ages, images, bounds)
        segs = [[(0, n - 1)], [(0, 9), (10, 24), (25, n - 1)]]
        out["meta_%d" % c] = np.frombuffer(json.dumps({"h": h, "w": w, "n": n, "segs": segs, "n_groups": n_groups}).encode(), np.uint8)
        out["ages_%d" % c] = np.asarray([v for g in range(n_groups) for v in ages[g]], np.int64)
        out["ages_off_%d" % c] = np.cumsum([0] + [len(ages[g]) for g in range(n_groups)]).astype(np.int64)
        out["bounds_%d" % c] = np.asarray([bounds[g] for g in range(n_groups)], np.int64)
        out["images_%d" % c] = np.packbits(np.concatenate([im.ravel() > 0 for g in range(n_groups) for im in images[g]]))
        for k, sg in enumerate(segs):
            with contextlib.redirect_stdout(io.StringIO()):
                keyframes, cc_times = KeyframeExtractor.GenerateFromST3DForIntervals(st3d, sg, False)
            kf = np.stack(keyframes)
            out["keyframes_%d_%d" % (c, k)] = np.packbits(kf[..., 0] == 255, axis=2)
            out["times_%d_%d" % (c, k)] = np.asarray([(s, *t) for s, lst in enumerate(cc_times) for t in lst], np.float64).reshape(-1, 6)
            print("case", c, "segments", sg, "groups drawn", [len(lst) for lst in cc_times])
    np.savez_compressed(os.path.join(HERE, "g8b_step05_ties.npz"), **out)


if __name__ == "__main__":
    make()
