#!/usr/bin/env python3
"""Throughput of the FCN forward pass with k engines on k HIP streams (frames dealt round-robin) vs one: the deep layers have only
288-540 workgroups, so a second pass in flight fills CUs the first leaves idle.   python tools/fcn_two_streams.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lecturemath_amd import _lib, fcn, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
lib = _lib.load()
H, W = 1080, 1920
sd = synth.fcn_random_state_dict(synth.FCN_SHIPPED_WIDTHS, pixel_kernel=7, seed=0)
rgb, _ = synth.whiteboard_rgb(H, W, 1500, seed=20211)
d = torch.from_numpy(rgb).cuda()
out = torch.empty((4, H, W), dtype=torch.float32, device="cuda")
for k in (1, 2, 3):
    engs = []
    for _ in range(k):
        e = fcn.FcnEngine(synth.FCN_SHIPPED_WIDTHS, 7, 3, H, W, lib)
        e.load_state_dict(sd)
        engs.append(e)
    streams = [torch.cuda.Stream() for _ in range(k)]
    def run(m):
        for i in range(m):
            j = i % k
            engs[j].forward_raw(d.data_ptr(), H, W, out[j].data_ptr(), None, None, streams[j].cuda_stream)
    run(2 * k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%d engine(s) / stream(s): %.1f frames/s (%.3f ms per frame)" % (k, n / dt, dt / n * 1e3))
    for e in engs:
        e.close()
