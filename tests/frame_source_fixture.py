"""The exported-lecture folder used by the frame-source tests (tests/test_frame_source.py) and by the script that recorded the
reference's calls on it (tests/golden/make_golden_frame_source.py): small colour PNG frames + index.json, deterministic."""
import hashlib
import json
import os

import numpy as np

FRAME_IDS = [30, 60, 150, 90, 1200, 7]        # written out of order on purpose; the processors sort them
H, W = 24, 40


def frame_pixels(fid):
    rng = np.random.default_rng(1000 + fid)
    return rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)         # BGR, as the workers receive it


def build(folder):
    from PIL import Image
    img_dir = os.path.join(folder, "JPEGImages")
    os.makedirs(img_dir, exist_ok=True)
    meta = {}
    for k, fid in enumerate(FRAME_IDS):
        Image.fromarray(frame_pixels(fid)[:, :, ::-1].copy()).save(os.path.join(img_dir, "%d.png" % fid))
        meta[str(fid)] = {"video_time": fid * 33.25, "frame_idx": fid, "abs_time": fid * 33.25 + 0.5, "video_idx": 0}
    with open(os.path.join(img_dir, "index.json"), "w") as f:
        json.dump(meta, f)
    return folder


class RecordingWorker:
    """The video-worker protocol, logging every call in a JSON-friendly form."""

    def __init__(self):
        self.log = []

    @staticmethod
    def _sha(a):
        return None if a is None else hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]

    def getWorkName(self):
        return "recording"

    def initialize(self, width, height):
        self.log.append(["initialize", int(width), int(height)])

    def handleFrame(self, frame, last_frame, v_index, abs_time, rel_time, abs_frame_idx):
        self.log.append(["handleFrame", list(frame.shape), str(frame.dtype), self._sha(frame), self._sha(last_frame), int(v_index),
                         float(abs_time), float(rel_time), int(abs_frame_idx), type(abs_frame_idx).__name__])

    def finalize(self):
        self.log.append(["finalize"])
