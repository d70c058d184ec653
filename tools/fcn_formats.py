#!/usr/bin/env python3
"""Per-layer operand-format assignments of the planar FCN engine, judged by what the path does with the logits: max |logit - oracle|,
BINARY FLIPS against the oracle's binarization (oracle/fcn.py forward + oracle/cc.py threshold_invert: the pixels of the
binarised frame that differ), and ms per frame (HIP events over `passes` forward passes).  Shipped widths, 1920x1080.
    python tools/fcn_formats.py out.json [seeds] [passes] [only=name,name]
The assignments are overrides on top of precision="mixed" (lecturemath_amd/fcn.py FcnEngine(formats=...))."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lecturemath_amd import _lib, fcn, synth
from oracle import fcn as ofcn
from oracle import cc as occ

D1, U1, TXT, PX1, PX2, OUT = 0, 15, 16, 18, 19, 20
ASSIGN = {       # overrides on top of precision="mixed" (fcn.FcnEngine.MIXED_FORMATS: conv_up_1, text / rec heads, conv_pixels_1 on w2 since round 4)
    "mixed": {},
    "r3-mixed (all six f16x3)": {U1: "f16x3", TXT: "f16x3", PX1: "f16x3"},
    "up1=w2": {TXT: "f16x3", PX1: "f16x3"},
    "up1,px1=w2": {TXT: "f16x3"},
    "px1=a2": {PX1: "a2"},
    "px2=a2": {PX2: "a2"},
    "px2=w2": {PX2: "w2"},
    "px1=a2,px2=a2": {PX1: "a2", PX2: "a2"},
    "out=a2": {OUT: "a2"},
    "d1=w2": {D1: "w2"},
    "px2=a2,out=a2,d1=w2": {PX2: "a2", OUT: "a2", D1: "w2"},
    "planar-f16": None,
}
out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/fcn_formats.json"
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 30
only = None
for a in sys.argv[4:]:
    if a.startswith("only="):
        only = a[5:].split(";")
H, W = 1080, 1920
lib = _lib.load()
torch.set_num_threads(os.cpu_count())
cases = []
for seed in range(seeds):
    sd = ofcn.random_state_dict(ofcn.SHIPPED_WIDTHS, pixel_kernel=7, seed=seed)
    rgb, _ = synth.whiteboard_rgb(H, W, n_glyphs=1500, seed=20211 + seed)
    t = time.time()
    with torch.no_grad():
        o, tt, r = ofcn.forward(sd, ofcn.prepare_image(rgb))
    o = o[0, 0].numpy()
    cases.append((sd, rgb, o, occ.threshold_invert(o)))
    print("oracle seed %d: %.1f s, logit std %.3f, foreground %.3f" % (seed, time.time() - t, o.std(), (cases[-1][3] == 0).mean()), flush=True)
res = {"frame": [H, W], "seeds": seeds, "passes": passes, "assignments": {}}
for name, fm in ASSIGN.items():
    if only and name not in only:
        continue
    worst, flips, ms = 0.0, [], []
    for sd, rgb, o, obin in cases:
        eng = fcn.FcnEngine(ofcn.SHIPPED_WIDTHS, 7, 3, H, W, lib, precision="planar-f16" if fm is None else "mixed", formats=fm)
        eng.load_state_dict(sd)
        d = torch.from_numpy(rgb).cuda()
        out = eng.forward(d)[0]
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        worst = max(worst, float(np.abs(got - o).max()))
        flips.append(int((occ.threshold_invert(got) != obin).sum()))
        if len(ms) == 0:
            outs = [torch.empty_like(out) for _ in range(3)]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                eng.forward_raw(_lib.ptr(d), H, W, _lib.ptr(outs[0]), _lib.ptr(outs[1]), None)
            e0.record()
            for _ in range(passes):
                eng.forward_raw(_lib.ptr(d), H, W, _lib.ptr(outs[0]), _lib.ptr(outs[1]), None)
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1) / passes)
        eng.close()
    res["assignments"][name] = {"max_abs_logit_diff_vs_oracle": worst, "binary_flips_vs_oracle": flips, "flip_fraction": max(flips) / float(H * W),
                                "ms_per_frame": ms[0]}
    print("%-24s max|dlogit| %.2e  flips %s  %.3f ms/frame" % (name, worst, flips, ms[0]), flush=True)
os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
json.dump(res, open(out_path, "w"), indent=1)
