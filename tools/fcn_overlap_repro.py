"""Two FCN forward passes in flight on two HIP streams (LM_FCN_NO_CHAIN=1 lifts the library's serialisation): do they disturb
each other?  Prints, per output, how many pixels differ from the single-pass result and where (tile geometry of the head kernels)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lecturemath_amd import _lib, fcn, synth

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
quiet = len(sys.argv) > 2
precision = os.environ.get("LM_FCN_PRECISION", "f16x3")
h, w = 1080, 1920
lib = _lib.load(os.environ.get("LM_LIB_PATH") or None)
sd = synth.fcn_random_state_dict(synth.FCN_SHIPPED_WIDTHS, pixel_kernel=7, seed=0)
engines = []
for _ in range(2):
    e = fcn.FcnEngine(synth.FCN_SHIPPED_WIDTHS, 7, 3, h, w, lib, precision=precision)
    e.load_state_dict(sd)
    engines.append(e)
rgb, _ = synth.whiteboard_rgb(h, w, 1500, seed=20211)
d = torch.from_numpy(rgb).cuda()
gold = [t.clone() for t in engines[0].forward(d)]
gold1 = [t.clone() for t in engines[1].forward(d)]
torch.cuda.synchronize()
print("engines agree alone:", all(bool((a == b).all()) for a, b in zip(gold, gold1)))
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
names = ["out", "text", "rec"]
total = {n: 0 for n in names}
for it in range(iters):
    res = []
    for e, st in zip(engines, streams):
        with torch.cuda.stream(st):
            res.append(e.forward(d))
    torch.cuda.synchronize()
    for k, r in enumerate(res):
        for n, a, b in zip(names, gold, r):
            bad = (a != b)
            nb = int(bad.sum())
            total[n] += nb
            if nb and not quiet:
                idx = bad.nonzero()
                err = float((a - b).abs().max())
                ys = idx[:, -2].cpu().numpy()
                xs = idx[:, -1].cpu().numpy()
                print("iter %d engine %d %s: %d px differ, max err %.3e; rows %s; x%%4 histogram %s; x//64 tiles %s" %
                      (it, k, n, nb, err, sorted(set(ys.tolist()))[:8], np.bincount(xs % 4, minlength=4).tolist(), sorted(set((xs // 64).tolist()))[:10]))
print("TOTAL differing pixels over %d iterations:" % iters, total)
