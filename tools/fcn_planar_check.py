#!/usr/bin/env python3
"""Planar FCN engine (csrc/lm_fcn2.hip) against the torch fp32 oracle: shipped widths, an odd-sized frame and 1080p.
    python tools/fcn_planar_check.py [--no-1080p]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lecturemath_amd import _lib, fcn, synth
from oracle import fcn as ofcn
lib = _lib.load()
cases = [(270, 478, 120, 4)] + ([] if "--no-1080p" in sys.argv else [(1080, 1920, 1500, 20211)])
for H, W, ng, seed in cases:
    sd = ofcn.random_state_dict(ofcn.SHIPPED_WIDTHS, pixel_kernel=7, seed=0)
    rgb, _ = synth.whiteboard_rgb(H, W, n_glyphs=ng, seed=seed)
    torch.set_num_threads(os.cpu_count())
    t = time.time()
    with torch.no_grad():
        o, tt, r = ofcn.forward(sd, ofcn.prepare_image(rgb))
    print("oracle %dx%d: %.1f s" % (W, H, time.time() - t), flush=True)
    for prec in ("mixed", "planar-f16x3", "planar-f16", "f16x3"):
        eng = fcn.FcnEngine(ofcn.SHIPPED_WIDTHS, 7, 3, H, W, lib, precision=prec)
        eng.load_state_dict(sd)
        out, text, rec = (x.cpu().numpy() for x in eng.forward(rgb))
        out2 = eng.forward(rgb)[0].cpu().numpy()
        print("%dx%d %-13s logit %.2e text %.2e rec %.2e  (second pass identical: %s)" % (
            W, H, prec, np.abs(out - o[0, 0].numpy()).max(), np.abs(text - tt[0, 0].numpy()).max(), np.abs(rec - r[0].numpy()).max(),
            bool((out == out2).all())), flush=True)
        eng.close()
