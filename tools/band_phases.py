"""Phase timing inside lm_k_band from its wall-clock stamps (LM_DEBUG_BAND_STAMPS=<file>, 100 MHz constant clock):
python tools/band_phases.py [batch]   (runs one labelled batch of dense 1080p frames and prints per-phase means)"""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = os.path.join(tempfile.gettempdir(), "lm_band_stamps.bin")
os.environ["LM_DEBUG_BAND_STAMPS"] = path
import torch
from lecturemath_amd import _lib, device, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
first = 5000
frames = np.stack([f for i, f in enumerate(synth.binary_stream(first + B, 1080, 1920, seed=20213)) if i >= first])
lab = device.FrameLabeler(1920, 1080, B)
d = torch.from_numpy(frames).cuda()
for _ in range(3):
    labels, counts = lab.label(d)
s = np.fromfile(path, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
t = s[:, :5]
tick_us = 0.01          # wall_clock64: 100 MHz
names = ["tables -> LDS", "row offsets + init", "unions", "flatten + store"]
print("workgroups %d, runs per band mean %.0f max %d" % (len(s), s[:, 5].mean(), s[:, 5].max()))
for k, nm in enumerate(names):
    dt = (t[:, k + 1] - t[:, k]) * tick_us
    print("%-22s mean %6.2f us  p95 %6.2f us" % (nm, dt.mean(), np.percentile(dt, 95)))
tot = (t[:, 4] - t[:, 0]) * tick_us
print("%-22s mean %6.2f us  p95 %6.2f us" % ("workgroup total", tot.mean(), np.percentile(tot, 95)))
print("kernel span %.1f us (first start to last end)" % ((t[:, 4].max() - t[:, 0].min()) * tick_us))
start = (t[:, 0] - t[:, 0].min()) * tick_us
print("workgroup start times: p50 %.1f us, p90 %.1f us, max %.1f us" % (np.percentile(start, 50), np.percentile(start, 90), start.max()))
