#!/usr/bin/env python3
"""The two chains of steps 01-02 ALONE on the 10,000-frame bench stream (for rocprofv3 --kernel-trace --stats): first the wide chain
(logits -> labels -> stats -> records / crops, no matching), then the temporal matching of the whole stream in batches of 64, then
step 03 + rendering.  Wall times per chain are printed; per-kernel averages come from the trace.
    python tools/chain_profile.py [frames [first frame of the stream to take]]      LM_VARIANT_LIB=<path of another build of the library>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from lecturemath_amd import _lib, device, synth
F = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
H, W, B = 1080, 1920, 64
F0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib = _lib.load(os.environ.get("LM_VARIANT_LIB"))
logits = bench.make_logits(torch, synth, F0 + F, H, W, 20213, F0, F0 + F)
fs = device.FrameStream(W, H, F, 0.85, 0.85, 85, 20, max_batch=B, max_ccs=F * 4096, max_crop_words=F * max(1 << 17, (W * H) // 16), lib=lib)
labels = torch.empty((B, H, W), dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for rep in range(2):
    fs.reset()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lib.check(lib.lm_stream_run_logits(fs.handle, logits.data_ptr(), F, B, None, labels.data_ptr(), 128, 0, 0, st, st))
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for f0 in range(0, F, B):
        fs.match(min(B, F - f0))
    torch.cuda.synchronize(); t2 = time.perf_counter()
    gr = device.Grouping(fs, max_gap=85, min_times=3, t_window=5, min_recall=0.5, img_threshold=0.5, reconstruct=True)
    clean = torch.empty((B, H, W), dtype=torch.uint8, device="cuda")
    for f0 in range(0, F, B):
        gr.render(f0, min(B, F - f0), clean[:min(B, F - f0)])
    torch.cuda.synchronize(); t3 = time.perf_counter()
    gr.close()
    print("rep %d: wide chain %.1f ms, matching %.1f ms, step 03 + render %.1f ms" % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), flush=True)
