#!/usr/bin/env python3
"""Times lm_fcn_forward alone on one 1080p frame (shipped widths) with per-kernel-class totals left to rocprofv3.
LM_LIB_PATH selects a variant build (tools/variants); results of experiment builds are timing-only.
    python tools/fcn_microbench.py [precision] [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lecturemath_amd import _lib, fcn, synth
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lib = _lib.load(os.environ.get("LM_LIB_PATH") or None)
H, W = 1080, 1920
sd = synth.fcn_random_state_dict(synth.FCN_SHIPPED_WIDTHS, pixel_kernel=7, seed=0)
eng = fcn.FcnEngine(synth.FCN_SHIPPED_WIDTHS, 7, 3, H, W, lib, precision=prec)
eng.load_state_dict(sd)
rgb, _ = synth.whiteboard_rgb(H, W, 1500, seed=20211)
d = torch.from_numpy(rgb).cuda()
for _ in range(3):
    eng.forward(d)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    eng.forward(d)
e1.record(); torch.cuda.synchronize()
print("lib=%s precision=%s: %.3f ms/frame" % (os.path.basename(os.environ.get("LM_LIB_PATH") or "default"), prec, e0.elapsed_time(e1) / n))
