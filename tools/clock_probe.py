#!/usr/bin/env python3
"""Shader clock held by the chip idle vs under the FCN forward pass (tools/ubench/clock_probe.hip built as tools/variants/libclockprobe.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lecturemath_amd import _lib, fcn, synth
lp = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", "libclockprobe.so"))
lp.lm_clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
out = torch.zeros(2, dtype=torch.int64, device="cuda")
side = torch.cuda.Stream()
def probe(ms=20):
    lp.lm_clock_probe_launch(out.data_ptr(), int(ms * 1e5), side.cuda_stream)       # wall clock: 100 MHz
    side.synchronize()
    c, w = out.tolist()
    return c / w * 100.0        # MHz
print("idle: %.0f MHz" % probe())
lib = _lib.load()
H, W = 1080, 1920
eng = fcn.FcnEngine(synth.FCN_SHIPPED_WIDTHS, 7, 3, H, W, lib, precision=(sys.argv[1] if len(sys.argv) > 1 else "f16x3"))
eng.load_state_dict(synth.fcn_random_state_dict(synth.FCN_SHIPPED_WIDTHS, pixel_kernel=7, seed=0))
rgb, _ = synth.whiteboard_rgb(H, W, 1500, seed=20211)
d = torch.from_numpy(rgb).cuda()
for _ in range(40):
    eng.forward(d)                      # ~190 ms of FCN work queued on the current stream
print("under the FCN (%s): %.0f MHz" % (eng.precision, probe(60)))
torch.cuda.synchronize()
