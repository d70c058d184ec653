#!/usr/bin/env python3
"""Times lm_label_batch alone (torch events on the launching stream) on bench-like frames.
LM_DEBUG_BAND_PHASES=1|2 truncates lm_k_band to see where its time goes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lecturemath_amd import _lib, device, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
frames = np.stack(list(synth.binary_stream(256, 1080, 1920, seed=20213)))[-B:]     # late (dense) part of the stream
d = torch.from_numpy(frames).cuda()
lab = device.FrameLabeler(1920, 1080, B)
labels = torch.empty((B, 1080, 1920), dtype=torch.int32, device="cuda")
lib = lab.lib
st = torch.cuda.current_stream().cuda_stream
for want in (labels, None):
    for _ in range(3):
        lib.check(lib.lm_label_batch(lab.ctx, d.data_ptr(), B, want.data_ptr() if want is not None else None, st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.check(lib.lm_label_batch(lab.ctx, d.data_ptr(), B, want.data_ptr() if want is not None else None, st))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("phases=%s labels=%s B=%d: %.1f us/launch, %.2f us/frame, %.0f GB/s algorithmic" % (
        os.environ.get("LM_DEBUG_BAND_PHASES", "3"), want is not None, B, ms * 1e3, ms * 1e3 / B, 5 * 1920 * 1080 * B / (ms * 1e-3) / 1e9))
