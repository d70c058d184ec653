#!/usr/bin/env python3
"""Wall-clock breakdown of one bench step (synchronising after each stage)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lecturemath_amd import _lib, device, synth
H, W, F, B = 1080, 1920, 256, 32
lib = _lib.load()
frames = torch.from_numpy(np.stack(list(synth.binary_stream(F, H, W, seed=20213)))).cuda()
logits = torch.where(frames > 0, -4.0, 4.0).to(torch.float32)
fs = device.FrameStream(W, H, F, 0.85, 0.85, 85, 20, max_batch=B, lib=lib)
binary = torch.empty((F, H, W), dtype=torch.uint8, device="cuda")
labels = torch.empty((B, H, W), dtype=torch.int32, device="cuda")
clean = torch.empty((B, H, W), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for rep in range(3):
    t0 = sync(); fs.reset()
    lib.check(lib.lm_threshold_invert(logits.data_ptr(), binary.data_ptr(), F * H * W, 128, st)); t1 = sync()
    for f0 in range(0, F, B):
        lib.check(lib.lm_stream_push(fs.handle, binary[f0:f0 + B].data_ptr(), B, labels.data_ptr(), st))
        if rep == 2 and os.environ.get("LM_MATCH_STATS"):
            k = np.zeros(5, np.int64)
            lib.check(lib.lm_stream_match_stats(fs.handle, k.ctypes.data, st))
            print("   batch at frame %d: sources %d tiles %d pairsA %d pairsB %d" % (f0, k[0], k[1], k[2], k[3]), fs.counters()["n_active"])
    t2h = time.perf_counter(); t2 = sync()
    gr = device.Grouping(fs, reconstruct=True); t3 = sync()
    for f0 in range(0, F, B): gr.render(f0, B, clean)
    t4 = sync(); gr.close(); t5 = sync()
    print("rep %d: threshold %.2f ms | push %.2f ms (host enqueue %.2f) | group %.2f ms | render %.2f ms | close %.2f ms | total %.2f" % (
        rep, (t1-t0)*1e3, (t2-t1)*1e3, (t2h-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, (t5-t4)*1e3, (t5-t0)*1e3))
