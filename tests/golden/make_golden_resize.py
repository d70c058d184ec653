"""G6b: Pillow's LANCZOS resize (FCN_lecturenet.py:436, `PIL_image.resize((w // 2, h // 2), PIL.Image.LANCZOS)`, the > 2.5 MP branch of
binarize()) on small RGB images, recorded from the Pillow installed in the build container (third-party arithmetic that the
reference's own tests do not pin).  Inputs and outputs are committed; lecturemath_amd/resize.py + lm_resample_rgb8
must reproduce the outputs byte for byte.  Run from the repository root:  python tests/golden/make_golden_resize.py"""
import os

import numpy as np
import PIL
import PIL.Image

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(606)
out = {"pil_version": np.asarray(PIL.__version__)}
cases = []
# (h, w, out_h, out_w): halvings of even and odd sizes (int(w / 2)), a 3:1 reduction, an enlargement, a one-axis change
for i, (h, w, oh, ow) in enumerate(((256, 384, 128, 192), (203, 301, 101, 150), (96, 130, 32, 43), (40, 50, 70, 90), (64, 200, 64, 100))):
    if i % 2 == 0:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)                        # noise: every coefficient matters, clipping at both ends
    else:
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx // 7 + yy // 5) % 2) * 255], axis=2).astype(np.uint8)   # ramps + checker
    res = np.asarray(PIL.Image.fromarray(img).resize((ow, oh), PIL.Image.LANCZOS))
    out["in%d" % i], out["out%d" % i] = img, res
    cases.append((h, w, oh, ow))
out["cases"] = np.asarray(cases, np.int32)
np.savez_compressed(os.path.join(HERE, "g6b_lanczos.npz"), **out)
print("wrote g6b_lanczos.npz:", cases, "Pillow", PIL.__version__)
