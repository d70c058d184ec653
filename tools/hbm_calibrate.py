#!/usr/bin/env python3
"""Achievable HBM rates for the access shapes of the labelling kernels (pure write / pure read / copy), via torch."""
import torch
n = 32 * 1080 * 1920
x = torch.empty(n, dtype=torch.int32, device="cuda")
y = torch.empty(n, dtype=torch.int32, device="cuda")
u8 = torch.empty(n, dtype=torch.uint8, device="cuda")
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
s = t(lambda: x.zero_());        print("fill  int32 %6.1f MB: %6.1f us  %.2f TB/s write" % (n*4/1e6, s*1e6, n*4/s/1e12))
s = t(lambda: x.fill_(7));       print("fill_ int32 %6.1f MB: %6.1f us  %.2f TB/s write" % (n*4/1e6, s*1e6, n*4/s/1e12))
s = t(lambda: y.copy_(x));       print("copy  int32 %6.1f MB: %6.1f us  %.2f TB/s read+write" % (n*4/1e6, s*1e6, 2*n*4/s/1e12))
s = t(lambda: u8.sum());         print("read  uint8 %6.1f MB: %6.1f us  %.2f TB/s read" % (n/1e6, s*1e6, n/s/1e12))
s = t(lambda: x.sum());          print("read  int32 %6.1f MB: %6.1f us  %.2f TB/s read" % (n*4/1e6, s*1e6, n*4/s/1e12))
