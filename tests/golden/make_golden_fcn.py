#!/usr/bin/env python3
"""G5: FCN-LectureNet golden vectors produced by the REFERENCE module (container-only; see make_golden.py).

For tiny configurations (so the fixture stays small) with randomised BN statistics:
  state_dict + RGB uint8 input -> forward() outputs (binarization logit, text-mask logit, reconstruction),
  binarize(return_others=True, force_binary=True) outputs, a few intermediate activations.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import ref_env  # noqa: E402

assert ref_env.available()
ref_env.enter()
import torch  # noqa: E402
from PIL import Image  # noqa: E402
from AccessMath.lecturenet_v1.FCN_lecturenet import FCN_LectureNet  # noqa: E402
from lecturemath_amd import synth  # noqa: E402
from oracle import fcn as ofcn  # noqa: E402

CASES = [
    dict(name="k7_70x94", widths=(8, 16, 16, 32, 32, 32, 16, 16, 16, 16, 16, 16, 8, 8, 8, 8, 8, 8), pk=7, h=70, w=94, seed=1),
    dict(name="k3_135x240", widths=(8, 16, 16, 32, 32, 32, 16, 16, 16, 16, 16, 16, 8, 8, 8, 8, 8, 8), pk=3, h=135, w=240, seed=2),
    dict(name="k7_66x130_wide", widths=(16, 16, 32, 32, 48, 48, 32, 32, 16, 32, 16, 16, 16, 16, 16, 16, 32, 16), pk=7, h=66, w=130, seed=3),
]


def build_reference(widths, pk):
    d1, d2, d3, d4, d5, mid, u5, c5, u4, c4, u3, c3, u2, c2, u1, c1, pm1, pm2 = widths
    return FCN_LectureNet(3, d1, d2, d3, d4, d5, mid, u5, c5, u4, c4, u3, c3, u2, c2, u1, c1, 3, pm1, pm2, pk, False)


for case in CASES:
    sd = ofcn.random_state_dict(case["widths"], pixel_kernel=case["pk"], seed=case["seed"])
    net = build_reference(case["widths"], case["pk"])
    missing = net.load_state_dict(sd, strict=True)
    net.eval()
    rgb, _ = synth.whiteboard_rgb(case["h"], case["w"], n_glyphs=25, seed=case["seed"])
    pil = Image.fromarray(rgb)
    with torch.no_grad():
        x0 = FCN_LectureNet.prepare_image(pil)
        out, text, rec = net.forward(x0)
        x_up1 = net.encode_decode(x0)
    binary, text_mask, rec_img = net.binarize(pil, return_others=True, force_binary=True)
    o = {"widths": np.asarray(case["widths"]), "pk": np.int64(case["pk"]), "rgb": rgb,
         "out": out.numpy(), "text": text.numpy(), "rec": rec.numpy(), "x_up1": x_up1.numpy(),
         "binary": binary, "text_mask": text_mask, "rec_img": rec_img}
    for k, v in sd.items():
        o["sd." + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "g5_fcn_%s.npz" % case["name"]), **o)
    # the oracle restatement must agree with the module
    with torch.no_grad():
        o2, t2, r2 = ofcn.forward(sd, ofcn.prepare_image(rgb))
    print(case["name"], "params", sum(v.numel() for v in sd.values()), "oracle-vs-module max abs",
          float((o2 - out).abs().max()), float((t2 - text).abs().max()), float((r2 - rec).abs().max()),
          "binary ink frac %.3f" % (binary == 0).mean())
