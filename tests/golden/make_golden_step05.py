#!/usr/bin/env python3
"""G8: step 05 core of the reference, KeyframeExtractor.GenerateFromST3DForIntervals (keyframe_extractor.py:13-145), run in
THIS container on the three golden streams with the video segments of G7 (parameter set 2) and with one whole-stream segment.
Stored: the step-03 inputs that matter (group ages, boundaries, group images) and the outputs (keyframes, times).
numpy >= 1.24 lacks np.bool, which the reference still uses (:69): aliased to bool here (environment stand-in)."""
import contextlib
import io
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_env  # noqa: E402
from lecturemath_amd import synth  # noqa: E402

ref_env.enter()
if not hasattr(np, "bool"):
    np.bool = bool
from AccessMath.preprocessing.content.cc_stability_estimator import CCStabilityEstimator  # noqa: E402
from AccessMath.preprocessing.content.keyframe_extractor import KeyframeExtractor  # noqa: E402
from AccessMath.data.space_time_struct import SpaceTimeStruct  # noqa: E402


def make(name):
    g = np.load(os.path.join(HERE, "g3_stream_%s.npz" % name))
    g7 = np.load(os.path.join(HERE, "g7_step04_%s.npz" % name))
    spec = json.loads(bytes(g["spec"]).decode())
    h, w = spec["h"], spec["w"]
    frames = list(synth.binary_stream(spec["n"], h, w, **spec["gen"]))
    est = CCStabilityEstimator(w, h, 0.85, 0.85, spec["gap2"], False)
    for f in frames:
        est.add_frame(f, True)
    with contextlib.redirect_stdout(io.StringIO()):
        est.split_stable_cc_by_gaps(spec["gap3"], 3)
        stable = est.get_stable_cc_idxs(3)
        tov, total, aov = est.compute_overlapping_stable_cc(stable, 5)
        groups, gid = est.compute_groups(stable, tov, 0.5, None, None)
        ages, gpf = est.compute_groups_temporal_information(groups)
        gimg, gb = est.compute_group_images(groups, ages, 0.5)
    n = len(frames)
    times = [1000.0 * i for i in range(n)]
    st3d = SpaceTimeStruct(times, list(range(n)), est.height, est.width, ages, gimg, gb)
    out = {"n_frames": np.int64(n), "h": np.int64(h), "w": np.int64(w)}
    seg_sets = [[tuple(int(v) for v in iv) for iv in g7["intervals_2"]], [(0, n - 1)], [(0, n // 3), (n // 3 + 1, n - 1)]]
    out["segments"] = np.frombuffer(json.dumps(seg_sets).encode(), np.uint8)
    for k, segs in enumerate(seg_sets):
        with contextlib.redirect_stdout(io.StringIO()):
            keyframes, cc_times = KeyframeExtractor.GenerateFromST3DForIntervals(st3d, segs, False)
        kf = np.stack(keyframes)                                    # [segments][H][W][3] uint8
        assert set(np.unique(kf)) <= {0, 255} and (kf[..., 0] == kf[..., 1]).all() and (kf[..., 0] == kf[..., 2]).all()
        out["keyframes_%d" % k] = np.packbits(kf[..., 0] == 255, axis=2)
        flat = [(s, *t) for s, lst in enumerate(cc_times) for t in lst]
        out["times_%d" % k] = np.asarray(flat, np.float64).reshape(-1, 6)
        print(name, "segments", segs, "ink px per keyframe", [int((kf[i, :, :, 0] == 0).sum()) for i in range(len(segs))])
    np.savez_compressed(os.path.join(HERE, "g8_step05_%s.npz" % name), **out)


if __name__ == "__main__":
    for nm in ("accumulate_erase", "occluder_return", "short_gap_jitter"):
        make(nm)
