"""Records what the reference's ImageListProcessor (AccessMath/preprocessing/video_processor/image_list_processor.py:82-199) does
to a video worker on the folder of tests/frame_source_fixture.py -> tests/golden/g10_frame_source.json.
Container only (imports /root/reference through ref_env; cv2.imread is the PIL-backed shim: PNG is lossless)."""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_env
import frame_source_fixture as fx

ref_env.enter()
from AccessMath.preprocessing.video_processor.image_list_processor import ImageListProcessor

out = {}
with tempfile.TemporaryDirectory() as d:
    fx.build(d)
    for name, limit, forced in (("all", 0, None), ("limit2", 2, None), ("forced_same", 0, (fx.W, fx.H))):
        w = fx.RecordingWorker()
        p = ImageListProcessor(d, img_extension=".png")
        if forced:
            p.force_resolution(*forced)
        p.doProcessing(w, limit=limit, verbose=False)
        out[name] = w.log
json.dump(out, open(os.path.join(HERE, "g10_frame_source.json"), "w"), indent=1)
print({k: len(v) for k, v in out.items()})
