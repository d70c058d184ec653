#!/usr/bin/env python3
"""Per-layer error attribution of the fp16-split operand formats.

For every layer of the shipped-width network at 1920x1080 (3 seeds: weights xavier + randomised BatchNorm statistics, a whiteboard
frame per seed) ONE layer at a time is switched from f16x3 to f16x2 / f16 while every other layer stays f16x3, and the change of the
three outputs is recorded (max and rms of |logit - logit_f16x3|; the f16x3 pass itself is within 2e-6 of the fp32 torch oracle).
Then cumulative assignments are evaluated against the all-f16x3 output: layers sorted by their single-layer error, cheapest first.
    python tools/fcn_layer_precision.py out.json [seeds]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lecturemath_amd import _lib, fcn, synth

NAMES = {0: "conv_down_1", 1: "conv_down_2", 2: "conv_down_3", 3: "conv_down_4", 4: "conv_down_5", 5: "mid", 6: "tconv_5", 7: "tconv_4",
         8: "tconv_3", 9: "tconv_2", 10: "tconv_1", 11: "conv_up_5", 12: "conv_up_4", 13: "conv_up_3", 14: "conv_up_2", 15: "conv_up_1",
         16: "text+rec heads", 18: "conv_pixels_1", 19: "conv_pixels_2", 20: "conv_out"}
out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/fcn_layer_precision.json"
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
H, W = 1080, 1920
lib = _lib.load()
res = {"frame": [H, W], "seeds": seeds, "layers": {}, "cumulative": []}
per = {l: {"f16x2": [], "f16": []} for l in NAMES}
engines = []
for seed in range(seeds):
    sd = synth.fcn_random_state_dict(synth.FCN_SHIPPED_WIDTHS, pixel_kernel=7, seed=seed)
    eng = fcn.FcnEngine(synth.FCN_SHIPPED_WIDTHS, 7, 3, H, W, lib, precision="f16x3")
    eng.load_state_dict(sd)
    rgb, _ = synth.whiteboard_rgb(H, W, 1500, seed=20211 + seed)
    d = torch.from_numpy(rgb).cuda()
    ref = [t.clone() for t in eng.forward(d)]
    torch.cuda.synchronize()
    engines.append((eng, d, ref))
    for l in NAMES:
        for prec in ("f16x2", "f16"):
            eng.set_layer_precision(l, prec)
            got = eng.forward(d)
            torch.cuda.synchronize()
            e = (got[0] - ref[0]).abs()
            per[l][prec].append({"max": float(e.max()), "rms": float((e * e).mean().sqrt()), "text_max": float((got[1] - ref[1]).abs().max()),
                                 "rec_max": float((got[2] - ref[2]).abs().max())})
            eng.set_layer_precision(l, "f16x3")
    print("seed", seed, "logit range", float(ref[0].min()), float(ref[0].max()), "std", float(ref[0].std()), flush=True)
for l in NAMES:
    res["layers"][NAMES[l]] = {p: {"max": max(x["max"] for x in per[l][p]), "rms": max(x["rms"] for x in per[l][p]),
                                   "text_max": max(x["text_max"] for x in per[l][p]), "rec_max": max(x["rec_max"] for x in per[l][p])}
                               for p in ("f16x2", "f16")}
    print("%-16s f16x2 max %.2e rms %.2e | f16 max %.2e rms %.2e" % (NAMES[l], res["layers"][NAMES[l]]["f16x2"]["max"], res["layers"][NAMES[l]]["f16x2"]["rms"],
                                                                    res["layers"][NAMES[l]]["f16"]["max"], res["layers"][NAMES[l]]["f16"]["rms"]), flush=True)
# cumulative: switch layers in ascending order of their single-layer f16x2 error; at every step measure the whole assignment
for prec in ("f16x2", "f16"):
    order = sorted(NAMES, key=lambda l: res["layers"][NAMES[l]][prec]["max"])
    for eng, d, ref in engines:
        for l in NAMES:
            eng.set_layer_precision(l, "f16x3")
    chosen = []
    for l in order:
        chosen.append(l)
        worst = {"max": 0.0, "text_max": 0.0, "rec_max": 0.0}
        for eng, d, ref in engines:
            eng.set_layer_precision(l, prec)
            got = eng.forward(d)
            torch.cuda.synchronize()
            worst["max"] = max(worst["max"], float((got[0] - ref[0]).abs().max()))
            worst["text_max"] = max(worst["text_max"], float((got[1] - ref[1]).abs().max()))
            worst["rec_max"] = max(worst["rec_max"], float((got[2] - ref[2]).abs().max()))
        res["cumulative"].append({"format": prec, "layers": [NAMES[x] for x in chosen], **worst})
        print("cumulative %s +%-16s -> logit %.2e text %.2e rec %.2e" % (prec, NAMES[l], worst["max"], worst["text_max"], worst["rec_max"]), flush=True)
os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
json.dump(res, open(out_path, "w"), indent=1)
