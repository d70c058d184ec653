#!/usr/bin/env python3
"""G7: step 04 (deletion-event video segmentation, VIDEO_SEGMENTATION_METHOD = 3) of the reference, run in THIS container
on the three golden streams, for the shipped parameters and two more sensitive parameter sets.

For every stream the reference's own step-03 calls (same order and parameters as pre_ST3D_v3.0_03_cc_grouping.py:41-118)
build the inputs, then the reference's pre_ST3D_v3.0_04_vid_segmentation.process_input runs unmodified (matplotlib is a
no-op stand-in, tests/golden/_ref_shims/matplotlib).  Stored: inputs of step 04 (group ages, group boundaries, frame size,
reconstructed frames as bits) and its outputs (intervals; the binary sums VideoSegmenter.compute_binary_sums returns).
"""
import contextlib
import importlib.util
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
import cv2  # noqa: E402  (the stand-in)
from AccessMath.preprocessing.content.cc_stability_estimator import CCStabilityEstimator  # noqa: E402
from AccessMath.preprocessing.content.video_segmenter import VideoSegmenter  # noqa: E402
from AccessMath.data.space_time_struct import SpaceTimeStruct  # noqa: E402

spec04 = importlib.util.spec_from_file_location("ref_step04", os.path.join(ref_env.REF_ROOT, "pre_ST3D_v3.0_04_vid_segmentation.py"))
ref_step04 = importlib.util.module_from_spec(spec04)
spec04.loader.exec_module(ref_step04)

PARAM_SETS = [
    {"VIDEO_SEGMENTATION_DEL_EVENT_ADD_THRESHOLD": 10, "VIDEO_SEGMENTATION_DEL_EVENT_MIN_LENGTH": 15,
     "VIDEO_SEGMENTATION_DEL_EVENT_THRESHOLD": 0.25},                     # shipped defaults (04_vid_segmentation.py:52-54)
    {"VIDEO_SEGMENTATION_DEL_EVENT_ADD_THRESHOLD": 10, "VIDEO_SEGMENTATION_DEL_EVENT_MIN_LENGTH": 3,
     "VIDEO_SEGMENTATION_DEL_EVENT_THRESHOLD": 0.001},
    {"VIDEO_SEGMENTATION_DEL_EVENT_ADD_THRESHOLD": 0.02, "VIDEO_SEGMENTATION_DEL_EVENT_MIN_LENGTH": 2,
     "VIDEO_SEGMENTATION_DEL_EVENT_THRESHOLD": 0.005},
]


class _Conf:
    def __init__(self, values):
        self.values = dict(values, VIDEO_SEGMENTATION_METHOD=3)

    def get_int(self, key, default=None):
        return int(self.values.get(key, default))

    def get_float(self, key, default=None):
        return float(self.values.get(key, default))

    def get(self, key, default=None):
        return self.values.get(key, default)


class _Lecture:
    title = "golden"


class _Process:
    def __init__(self, values):
        self.configuration = _Conf(values)
        self.img_dir = "."
        self.current_lecture = _Lecture()
        self.params = {}


def ragged(lists):
    flat = [v for lst in lists for v in lst]
    off = np.cumsum([0] + [len(lst) for lst in lists])
    return np.asarray(flat, np.int64), np.asarray(off, np.int64)


def make(name):
    g = np.load(os.path.join(HERE, "g3_stream_%s.npz" % name))
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
        conf = est.compute_conflicting_groups(stable, aov, len(groups), gid)
        gimg, gb = est.compute_group_images(groups, ages, 0.5)
        clean = est.frames_from_groups(groups, gb, gpf, ages, gimg, None, 3, True)
    n = len(frames)
    frame_times = [float(i) for i in range(n)]
    frame_indices = list(range(n))
    st3d = SpaceTimeStruct(frame_times, frame_indices, est.height, est.width, ages, gimg, gb)
    dec = np.stack([cv2.imdecode(c, cv2.IMREAD_GRAYSCALE) for c in clean])
    out = {"name": np.frombuffer(name.encode(), np.uint8), "n_frames": np.int64(n), "h": np.int64(h), "w": np.int64(w),
           "bounds": np.asarray([gb[k] for k in range(len(groups))], np.int64).reshape(-1, 4),
           "clean_values": dec.reshape(n, -1).astype(np.uint8) if dec.size < (1 << 22) else np.zeros(0, np.uint8),
           "sums": np.asarray(VideoSegmenter.compute_binary_sums(list(dec)), np.float64),
           "params": np.frombuffer(json.dumps(PARAM_SETS).encode(), np.uint8)}
    out["ages"], out["ages_off"] = ragged([ages[k] for k in range(len(groups))])
    for k, values in enumerate(PARAM_SETS):
        with contextlib.redirect_stdout(io.StringIO()):
            intervals = ref_step04.process_input(_Process(values), [(frame_times, frame_indices, clean), (ages, conf), st3d])
        out["intervals_%d" % k] = np.asarray(intervals, np.int64).reshape(-1, 2)
        print(name, "params", k, "->", [tuple(int(v) for v in iv) for iv in intervals])
    np.savez_compressed(os.path.join(HERE, "g7_step04_%s.npz" % name), **out)


if __name__ == "__main__":
    for nm in ("accumulate_erase", "occluder_return", "short_gap_jitter"):
        make(nm)
