"""Container-only (skipped where /root/reference is absent, e.g. on the GPU box): the drop-in VideoSegmenter's peak finder and
recursive splitter -- own formulations on the sign of the first difference / an explicit stack -- against the reference's
implementations (AccessMath/preprocessing/content/video_segmenter.py:133-182, 499-520) on thousands of random signals full of
ties and plateaus, including what they print."""
import contextlib
import importlib.util
import io
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import ref_env  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_env.available(), reason="needs the reference (build container only)")


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_peaks_and_splits_match_the_reference():
    cwd = os.getcwd()
    try:
        ref_env.enter()
        ref = _load("lm_ref_video_segmenter", os.path.join(ref_env.REF_ROOT, "AccessMath/preprocessing/content/video_segmenter.py"))
    finally:
        os.chdir(cwd)
    mine = _load("lm_dropin_video_segmenter", os.path.join(os.path.dirname(HERE), "lecturemath_amd/dropin/AccessMath/preprocessing/content/video_segmenter.py"))
    rng = np.random.default_rng(0)
    for trial in range(2000):
        n = int(rng.integers(1, 60))
        sig = (rng.integers(0, 4, n).astype(float), np.round(rng.random(n), 1), np.cumsum(rng.integers(-1, 2, n)).astype(float))[trial % 3]
        a = int(rng.integers(0, n))
        b = int(rng.integers(a, n))
        r = [tuple(int(v) for v in t) for t in ref.VideoSegmenter.find_signal_peaks(a, b, sig)]
        assert mine.VideoSegmenter.find_signal_peaks(a, b, sig) == r
        ml, thr = int(rng.integers(0, 6)), float(rng.choice([0.0, 0.5, 1.0, 2.0]))
        with contextlib.redirect_stdout(io.StringIO()) as o1:
            s1 = ref.VideoSegmenter.split_video_from_group_deletes(sig, 0, n - 1, ml, thr)
        with contextlib.redirect_stdout(io.StringIO()) as o2:
            s2 = mine.VideoSegmenter.split_video_from_group_deletes(sig, 0, n - 1, ml, thr)
        assert [tuple(int(v) for v in t) for t in s1] == s2 and o1.getvalue() == o2.getvalue()
