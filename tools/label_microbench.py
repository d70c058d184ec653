#!/usr/bin/env python3
"""Times lm_label_batch alone (torch events on the launching stream) on frames of the bench stream.
    python tools/label_microbench.py [batch] [height width] [first_frame]
first_frame picks the part of the stream (default: frames 5000.. of the 10,000-frame stream are dense; 192.. are the sparse
start).  LM_LIB_PATH selects a variant build (tools/variants); LM_DEBUG_BAND_PHASES=2 leaves lm_k_band's unions to L2."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lecturemath_amd import _lib, device, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1080, 1920)
first = int(sys.argv[4]) if len(sys.argv) > 4 else 5000
scale = 4 if H > 1080 else 1
gen = synth.binary_stream(first + B, H, W, seed=20213, glyphs_per_add=40 * scale, max_ext=28 * (2 if scale > 1 else 1))
frames = np.stack([f for i, f in enumerate(gen) if i >= first])
d = torch.from_numpy(frames).cuda()
lib = _lib.load(os.environ.get("LM_LIB_PATH") or None)
lab = device.FrameLabeler(W, H, B, lib)
labels = torch.empty((B, H, W), dtype=torch.int32, device="cuda")
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
    print("%dx%d frames %d..%d lib=%s labels=%s B=%d: %.1f us/launch, %.2f us/frame, %.0f GB/s algorithmic = %.3f of 8 TB/s" % (
        W, H, first, first + B, os.path.basename(os.environ.get("LM_LIB_PATH") or "default"), want is not None, B, ms * 1e3, ms * 1e3 / B,
        5 * W * H * B / (ms * 1e-3) / 1e9, 5 * W * H * B / (ms * 1e-3) / 8e12))

# the fused launch: fp32 logits -> labels (lm_label_batch_logits), 4 B/px in + 4 B/px out
logits = torch.from_numpy(synth.logits_from_binary(frames, seed=1)).cuda()
for _ in range(3):
    lib.check(lib.lm_label_batch_logits(lab.ctx, logits.data_ptr(), B, 128, 1, None, labels.data_ptr(), st))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    lib.check(lib.lm_label_batch_logits(lab.ctx, logits.data_ptr(), B, 128, 1, None, labels.data_ptr(), st))
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print("%dx%d frames %d..%d FUSED logits -> labels (fused=%d) B=%d: %.1f us/launch, %.0f GB/s on 8 B/px = %.3f of 8 TB/s (%.3f on SURVEY's 5 B/px)" % (
    W, H, first, first + B, lib.lm_label_was_fused(lab.ctx), B, ms * 1e3, 8 * W * H * B / (ms * 1e-3) / 1e9, 8 * W * H * B / (ms * 1e-3) / 8e12,
    5 * W * H * B / (ms * 1e-3) / 8e12))
assert (labels.cpu().numpy() == np.stack([__import__("scipy.ndimage").ndimage.label(f)[0] for f in frames[:2]] + [labels[i].cpu().numpy() for i in range(2, B)])).all()

# the same launches timed the way bench.py times them inside its timed region: one HIP event pair PER launch (lm_ctx_set_profiling)
import ctypes
lib.check(lib.lm_ctx_set_profiling(lab.ctx, 1))
for _ in range(20):
    lib.check(lib.lm_label_batch(lab.ctx, d.data_ptr(), B, labels.data_ptr(), st))
ms, calls, nfr = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_int64(0)
lib.check(lib.lm_ctx_profile_read(lab.ctx, ctypes.addressof(ms), ctypes.addressof(calls), ctypes.addressof(nfr)))
lib.check(lib.lm_ctx_set_profiling(lab.ctx, 0))
print("per-launch events: %.1f us/launch over %d launches (%.3f of 8 TB/s)" % (ms.value * 1e3 / calls.value, calls.value,
      5 * W * H * B / (ms.value * 1e-3 / calls.value) / 8e12))
