#!/usr/bin/env python3
"""Where a workgroup of lm_k_g2 spends its life, from the per-wave cycle stamps of the diagnostic build (-DLM_G2_STAMPS,
tools/variants/liblm_stamps.so): prologue (first patch + weights landed), MFMA groups, waits at the group barriers (vmcnt + s_barrier),
chunk switches of single-buffered patches, epilogue (activation, conversion, stores drained).  One layer per process:
    LM_G2_STAMP_LAYER=18 python tools/fcn_stamps.py [formats, e.g. 15=w2,18=a2]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lecturemath_amd import _lib, fcn, synth
layer = int(os.environ["LM_G2_STAMP_LAYER"])
fm = {}
if len(sys.argv) > 1 and sys.argv[1]:
    for kv in sys.argv[1].split(","):
        k, v = kv.split("=")
        fm[int(k)] = v
here = os.path.dirname(os.path.abspath(__file__))
lib = _lib.load(os.path.join(here, "variants", "liblm_stamps.so"))
H, W = 1080, 1920
sd = synth.fcn_random_state_dict(synth.FCN_SHIPPED_WIDTHS, pixel_kernel=7, seed=0)
eng = fcn.FcnEngine(synth.FCN_SHIPPED_WIDTHS, 7, 3, H, W, lib, precision="mixed", formats=fm)
eng.load_state_dict(sd)
rgb, _ = synth.whiteboard_rgb(H, W, 1500, seed=20211)
d = torch.from_numpy(rgb).cuda()
for _ in range(4):
    eng.forward(d)
torch.cuda.synchronize()
NW, NS = 8192 * 4 * 2, 12
buf = np.zeros(NW * NS, np.uint64)
raw = ctypes.CDLL(lib.path)
raw.lm_debug_g2_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int64]
assert raw.lm_debug_g2_read_stamps(buf.ctypes.data, buf.size) == 0
s = buf.reshape(NW, NS).astype(np.int64)
r = eng.recipes[layer]
s = s[s[:, 0] > 0]
tot = s[:, 6] - s[:, 0]
pro = s[:, 1] - s[:, 0]
epi = s[:, 6] - s[:, 5]
comp, wait, sw, issue, bar = s[:, 2], s[:, 3], s[:, 4], s[:, 9], s[:, 10]
# s_memtime counts shader cycles, s_memrealtime 100 MHz ticks: the clock held during the kernel from the two
real = (s[:, 7] - s[:, 8]).astype(np.float64)
ok = real > 50
mhz = float(np.median(tot[ok] / real[ok]) * 100.0) if ok.any() else 2400.0
span = (s[:, 7].max() - s[:, 8].min()) / 100.0
print("layer %d  %s  waves %d  shader clock %.0f MHz" % (layer, {k: r[k] for k in ("kh", "kw", "terms", "mt", "chunks", "groups", "slices", "lds_bytes")}, len(s), mhz))
print("  kernel span %.1f us;  per wave (mean / median, us):" % span)
for name, v in (("total", tot), ("prologue", pro), ("slice loops", comp), ("group-end barrier", bar), ("chunk switches", sw), ("epilogue", epi)):
    print("    %-16s %8.2f %8.2f   %5.1f %%" % (name, v.mean() / mhz, np.median(v) / mhz, 100.0 * v.sum() / tot.sum()))
mf = r["slices"] * r["mt"] * 4 * {1: 1, 2: 2, 3: 3, 4: 2}[r["terms"]] * 16
print("  MFMA issue cycles per wave %d = %.2f us at this clock (alone on its SIMD)" % (mf, mf / mhz))
# concurrency: waves alive over time -> resident workgroups per CU on average
ev = np.concatenate([np.stack([s[:, 0], np.ones(len(s), np.int64)], 1), np.stack([s[:, 6], -np.ones(len(s), np.int64)], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
alive = np.cumsum(ev[:, 1])
dtv = np.diff(ev[:, 0])
print("  mean waves alive %.0f (%.2f per SIMD)" % ((alive[:-1] * dtv).sum() / max(dtv.sum(), 1), (alive[:-1] * dtv).sum() / max(dtv.sum(), 1) / 1024.0))
