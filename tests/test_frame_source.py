"""The image-folder frame source (dropin ImageListProcessor / ImageListGenerator) against the calls the reference's own class made
on the same folder (tests/golden/g10_frame_source.json, recorded by tests/golden/make_golden_frame_source.py), and feeding the
step-01 worker.  CPU only: host I/O."""
import json
import os
import sys

import numpy as np
import pytest

import dropin_checks
import frame_source_fixture as fx

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g10_frame_source.json")


def _processor(folder):
    if dropin_checks.DROPIN not in sys.path:
        sys.path.insert(0, dropin_checks.DROPIN)
    from AccessMath.preprocessing.video_processor.image_list_processor import ImageListProcessor
    return ImageListProcessor(folder, img_extension=".png")


@pytest.mark.parametrize("case,limit,forced", [("all", 0, None), ("limit2", 2, None), ("forced_same", 0, (fx.W, fx.H))])
def test_worker_calls_equal_the_reference(tmp_path, case, limit, forced):
    fx.build(str(tmp_path))
    p = _processor(str(tmp_path))
    if forced:
        p.force_resolution(*forced)
    w = fx.RecordingWorker()
    p.doProcessing(w, limit=limit, verbose=False)
    want = json.load(open(GOLD))[case]
    assert json.loads(json.dumps(w.log)) == want


def test_generator_protocol(tmp_path):
    """The VideoCapture look-alike: ascending ids, first image not skipped, get()/index2frameID() describe the frame read last,
    unknown properties are None, reading past the end fails."""
    fx.build(str(tmp_path))
    if dropin_checks.DROPIN not in sys.path:
        sys.path.insert(0, dropin_checks.DROPIN)
    from AccessMath.preprocessing.video_processor.image_list_processor import ImageListGenerator
    for preload in (False, True):
        g = ImageListGenerator(os.path.join(str(tmp_path), "JPEGImages"), "png", preload=preload)
        assert len(g) == len(fx.FRAME_IDS) and (g.width, g.height, g.channels) == (fx.W, fx.H, 3)
        seen = []
        while True:
            ok, frame = g.read()
            if not ok:
                break
            fid = g.index2frameID()
            assert (frame == fx.frame_pixels(fid)).all() and g.get("abs_time") == fid * 33.25 + 0.5 and g.get("no_such") is None
            seen.append(fid)
        assert seen == sorted(fx.FRAME_IDS) and g.read() == (False, None)


def test_forced_resolution_needs_opencv(tmp_path):
    fx.build(str(tmp_path))
    p = _processor(str(tmp_path))
    p.force_resolution(fx.W * 2, fx.H * 2)
    try:
        import cv2  # noqa: F401
        pytest.skip("OpenCV present: the resize is cv2's own")
    except ImportError:
        pass
    with pytest.raises(NotImplementedError):
        p.doProcessing(fx.RecordingWorker())


def test_missing_export_is_reported(tmp_path):
    with pytest.raises(Exception, match="not in the correct export format"):
        _processor(str(tmp_path)).doProcessing(fx.RecordingWorker())
