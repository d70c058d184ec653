#!/usr/bin/env python3
"""bench.py -- frames/sec of the hot path on BASELINE.json configs[2]: ONE synthetic 10,000-frame 1080p stream per step,
fp32 logits resident in HBM -> threshold+invert -> CC labelling (int32 label image written) -> CC statistics / records / bit
crops -> temporal matching over the whole stream (state carried across all 10,000 frames) -> step 03 (grouping, ages, group
images) -> reconstruction of all 10,000 frames.

    python bench.py --gpus N --steps K --warmup W        (N > 1: one rank per GPU -- under torch.distributed.run as the driver starts
                                                          it, or by itself: without WORLD_SIZE in the environment this process starts
                                                          `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child
                                                          BEFORE touching the GPU and relays its output and exit code)

N = 1: a "step" is one pass over the whole stream.  Batches of the stream pipeline inside a step (matching of batch k runs on
its own HIP stream under the labelling of batch k+1) and step 03 of step i overlaps steps 01-02 of step i+1 (--depth slots;
--depth 1 disables that).  N > 1 (configs[3]): the SAME stream is frame-range sharded (contiguous blocks of ceil(F/N) frames
per rank, lecturemath_amd/sharded.py): every rank thresholds + labels its block, the packed CC records travel to rank 0 in
one RCCL transfer per rank, rank 0 replays the matching and runs step 03; strong scaling, digests equal to N = 1.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      labelling launch sequence (lm_label_batch_logits), HBM bound: 8 B/px (4 B fp32 logit in + 4 B int32 label out; the
                byte frame of SURVEY 8(d)'s 5 B/px figure is never materialised -- that accounting is printed beside it as
                frac_survey_5Bpx) x W x H x frames per launch / mean launch duration, HIP events on the launching stream INSIDE the
                timed region
  cpu_baseline  the CPU oracle (single thread) on a bounded DENSE window of the same stream, the same work as `value` (steps 01-03 +
                reconstruction); the sparse first frames (steps 01-02 only) as a sub-key
  parity        sha256 digests of the step 02/03 products of the last timed step vs the digests the reference produced on the
                same stream (tests/golden/g9_stream1080p_digests.json)
  fcn           configs[1] measured in the same process: FCN-LectureNet forward at 1080p (ms/frame, algorithmic TFLOP/s,
                max |logit - oracle|) and the CPU oracle's rate
  e2e_rgb       RGB frames -> FCN -> threshold -> ... -> step 03 -> reconstructed frames, frames/s
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ALGO_BYTES_PER_PX = 5           # SURVEY.md 8(d)'s byte-frame accounting (1 B uint8 in + 4 B int32 label out), reported as frac_survey_5Bpx; the
                                # launch that runs reads fp32 logits and is priced on its own 8 B/px (roofline.algorithmic_bytes_per_px)
MFMA_F32_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32 matrix peak
MFMA_F16_PEAK_TFLOPS = 2500.0   # dense f16/bf16 MFMA peak
FCN_MFMA_PER_PRODUCT = {"f16": 1, "f16x2": 2, "f16x3": 3}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10, help="timed steps; a step is one pass over the whole stream")
    p.add_argument("--warmup", type=int, default=4, help="untimed steps; each of the --depth pipeline slots needs two to be warm (its step-03 arena is "
                                                         "sized by the first pass and re-allocated after it), so fewer than 2 x depth leave set-up cost in the timed steps")
    p.add_argument("--frames", type=int, default=10000, help="frames of the stream (BASELINE configs[2]: 10,000)")
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--batch", type=int, default=64, help="frames per labelling launch")
    p.add_argument("--cpu-frames", type=int, default=200, help="prefix of the stream timed on the CPU oracle (0 = skip)")
    p.add_argument("--no-labels", action="store_true", help="do not materialise the int32 label image")
    p.add_argument("--depth", type=int, default=2, help="N = 1: streams in flight (step 03 of one step under steps 01-02 of the next); 1 = none")
    p.add_argument("--seed", type=int, default=20213)
    p.add_argument("--fcn-precision", default=os.environ.get("LM_FCN_PRECISION", "mixed"),
                   choices=["mixed", "planar-f16x3", "planar-f16", "f16", "f16x2", "f16x3", "fp32"],
                   help="MFMA operand format of the FCN conv stack (fp32 accumulate in all); mixed = per layer (lecturemath_amd/fcn.py)")
    p.add_argument("--fcn-frames", type=int, default=50, help="frames timed for the `fcn` object (0 = skip fcn and e2e_rgb)")
    p.add_argument("--e2e-frames", type=int, default=64, help="RGB frames of the `e2e_rgb` measurement (0 = skip)")
    p.add_argument("--no-fcn-oracle", action="store_true", help="skip the CPU oracle forward pass (max |logit diff|, FCN cpu baseline)")
    p.add_argument("--workload", default="stream", choices=["stream", "fcn"],
                   help="stream = configs[2]/[3] (headline metric); fcn = configs[1] alone")
    return p.parse_args()


def fcn_flops(widths, pk, kk, h, w):
    """2*Cin*Cout*k*k*Hout*Wout per conv (transposed convs: Hin*Win), SURVEY.md 8(d) config 2."""
    d1, d2, d3, d4, d5, mid, u5, c5, u4, c4, u3, c3, u2, c2, u1, c1, pm1, pm2 = widths
    hs, ws = [h], [w]
    for _ in range(5):
        hs.append(hs[-1] // 2)
        ws.append(ws[-1] // 2)
    f = 0
    cin = 3
    for n, co in enumerate((d1, d2, d3, d4, d5)):
        f += 2 * cin * co * kk * kk * hs[n] * ws[n]
        cin = co
    f += 2 * d5 * mid * kk * kk * hs[5] * ws[5]
    prev = mid
    for n, (u, c, skip) in enumerate(((u5, c5, d5), (u4, c4, d4), (u3, c3, d3), (u2, c2, d2), (u1, c1, d1))):
        g = 4 - n
        f += 2 * prev * u * 4 * hs[g + 1] * ws[g + 1]
        f += 2 * (u + skip) * c * kk * kk * hs[g] * ws[g]
        prev = c
    px = h * w
    f += 2 * c1 * 1 * pk * pk * px + 2 * c1 * 3 * kk * kk * px
    f += 2 * (3 + c1) * pm1 * pk * pk * px + 2 * (3 + pm1) * pm2 * pk * pk * px + 2 * (3 + pm2) * 1 * pk * pk * px
    return f


# ----------------------------------------------------------------------------------------------------------------------
# configs[1]: FCN-LectureNet inference at 1080p, measured in this process
# ----------------------------------------------------------------------------------------------------------------------
MAX_FCN_PIXELS = 2.5e6       # FCN_LectureNet.MAX_PIXELS (FCN_lecturenet.py:430-437): larger frames are LANCZOS-halved first


def fcn_frame_size(H, W):
    while W * H > MAX_FCN_PIXELS:
        W, H = int(W / 2), int(H / 2)
    return H, W


def measure_fcn(a, lib, H, W, n_frames, with_oracle):
    import torch
    from lecturemath_amd import fcn, synth
    H, W = fcn_frame_size(H, W)      # configs[4]: the network always runs at <= 2.5 MP
    widths, pk = synth.FCN_SHIPPED_WIDTHS, 7
    sd = synth.fcn_random_state_dict(widths, pixel_kernel=pk, seed=0)
    eng = fcn.FcnEngine(widths, pk, 3, H, W, lib, precision=a.fcn_precision)
    eng.load_state_dict(sd)
    rgb, _ = synth.whiteboard_rgb(H, W, 1500, seed=20211)
    d_rgb = torch.from_numpy(rgb).cuda()
    for _ in range(2):
        out, text, rec = eng.forward(d_rgb)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n_frames):
        out, text, rec = eng.forward(d_rgb)
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    gpu_ms = e0.elapsed_time(e1) / n_frames
    fl = fcn_flops(widths, pk, 3, H, W)
    tflops = fl / (gpu_ms * 1e-3) / 1e12
    res = {"workload": "configs[1]: FCN-LectureNet (shipped widths, 15.8 M params, random init + randomised BN), one %dx%d frame" % (W, H),
           "precision": a.fcn_precision, "frames_timed": n_frames, "ms_per_frame": round(gpu_ms, 3), "frames_per_s": round(n_frames / wall, 2),
           "gflop_per_frame": round(fl / 1e9, 1), "algorithmic_tflops": round(tflops, 2)}
    if a.fcn_precision == "fp32":
        res.update({"peak_tflops": MFMA_F32_PEAK_TFLOPS, "frac_of_peak_algorithmic": round(tflops / MFMA_F32_PEAK_TFLOPS, 4)})
    else:
        res.update({"peak_tflops": MFMA_F16_PEAK_TFLOPS, "frac_of_peak_algorithmic": round(tflops / MFMA_F16_PEAK_TFLOPS, 4)})
        if eng.planar:
            # per layer: products per operand pair and the K walk chosen by lecturemath_amd/fcn2.py
            res["engine"] = "planar (csrc/lm_fcn2.hip)"
            names = {1: "f16 (1 MFMA per product)", 2: "a2: activations hi+lo (2)", 3: "f16x3: both split (3)", 4: "w2: weights hi+lo (2)"}
            res["layer_formats"] = {str(k): names[v["terms"]] + (", 16x32 tiles" if v["nc"] == 2 else "") + (", loader wave" if v["loader"] else "")
                                    for k, v in sorted(eng.recipes.items())}
            res["layer_formats_source"] = "profiles/r04_fcn_formats.json (error and binary flips per assignment, 3 seeds), profiles/r03_fcn_layer_precision.json"
            ex = eng.executed_gflop(H, W)
            res.update({"executed_gflop_per_frame": round(ex, 1), "executed_tflops": round(ex / gpu_ms, 2),
                        "frac_of_peak_executed": round(ex / gpu_ms / MFMA_F16_PEAK_TFLOPS, 4),
                        "mfma_per_product": "per layer (layer_formats): 3 or 2 in the full-resolution layers, 1 below"})
        else:
            k = FCN_MFMA_PER_PRODUCT[a.fcn_precision]
            res.update({"mfma_per_product": k, "executed_tflops": round(k * tflops, 2)})
    if with_oracle:
        from oracle import fcn as ofcn          # the checker / CPU baseline leg only
        torch.set_num_threads(os.cpu_count())
        t0 = time.perf_counter()
        with torch.no_grad():
            o, t, r = ofcn.forward(sd, ofcn.prepare_image(rgb))
        cdt = time.perf_counter() - t0
        res["max_abs_logit_diff_vs_oracle"] = float((out.cpu() - o[0, 0]).abs().max())
        res["max_abs_text_diff_vs_oracle"] = float((text.cpu() - t[0, 0]).abs().max())
        res["max_abs_rec_diff_vs_oracle"] = float((rec.cpu() - r[0]).abs().max())
        res["tolerance"] = 1e-3
        # what the path does with the logits: pixels of the binarised frame that differ from the oracle's binarization of ITS logits
        from oracle import cc as occ
        flips = int((occ.threshold_invert(out.cpu().numpy()) != occ.threshold_invert(o[0, 0].numpy())).sum())
        res["binary_flips_vs_oracle"] = {"count": flips, "fraction": flips / float(H * W), "logit_std": float(o.std()),
                                         "note": "random-init logits crowd the threshold; three seeds per assignment: profiles/r04_fcn_formats.json"}
        res["cpu_baseline"] = {"value": round(1.0 / cdt, 4), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": "one 1080p frame, oracle/fcn.py (torch fp32 CPU restatement of FCN_lecturenet.py:260-403)"}
    return res, eng, sd


def measure_e2e_rgb(a, lib, eng, H, W, n_frames, sd=None):
    """RGB uint8 frames resident in HBM -> FCN logits -> threshold+invert -> label/records/matching -> step 03 -> all frames
    reconstructed.  Random-init weights: the binarization is whatever the network emits (CC counts are reported).
    The frames are dealt to TWO engines on two HIP streams (the layers below 1/8 resolution launch 288-480 workgroups on 256 CUs: a
    second forward pass in flight fills what one leaves idle; profiles/r04_fcn_two_streams.txt: 437 -> 475 frames/s)."""
    import torch
    from lecturemath_amd import _lib, device, fcn, synth
    fh, fw = fcn_frame_size(H, W)
    engines = [eng]
    if sd is not None and eng.planar and not os.environ.get("LM_BENCH_ONE_FCN_STREAM"):
        e2 = fcn.FcnEngine(eng.widths, eng.pk, eng.kk, fh, fw, lib, precision=a.fcn_precision)
        e2.load_state_dict(sd)
        engines.append(e2)
    side = [torch.cuda.Stream() for _ in engines[1:]]
    big = (fh, fw) != (H, W)
    if big:
        n_frames = min(n_frames, 32)
    rgb = torch.from_numpy(np.stack(list(synth.whiteboard_stream(n_frames, H, W)))).cuda()
    logits = torch.empty((n_frames, fh, fw), dtype=torch.float32, device="cuda")
    clean = torch.empty((n_frames, H, W), dtype=torch.uint8, device="cuda")
    if big:         # configs[4]: LANCZOS halving -> network -> threshold -> NEAREST enlargement -> CC pipeline at the frame's own size
        from lecturemath_amd import resize
        rs = resize.DeviceResizer(lib)
        small = torch.empty((fh, fw), dtype=torch.uint8, device="cuda")
        binary = torch.empty((n_frames, H, W), dtype=torch.uint8, device="cuda")
    labels = torch.empty((min(a.batch, n_frames), H, W), dtype=torch.int32, device="cuda")
    fs = device.FrameStream(W, H, n_frames, 0.85, 0.85, 85, 20, max_batch=min(a.batch, n_frames), max_ccs=n_frames * 131072,
                            max_crop_words=n_frames * (1 << 21), lib=lib)
    st = torch.cuda.current_stream().cuda_stream

    def once():
        fs.reset()
        if big:
            for i in range(n_frames):
                half = rs.lanczos(rgb[i], fw, fh)
                eng.forward_raw(half.data_ptr(), fh, fw, logits[i].data_ptr(), None, None, st)
                lib.check(lib.lm_threshold_invert(logits[i].data_ptr(), small.data_ptr(), fh * fw, 128, st))
                lib.check(lib.lm_upsample_nearest_u8(small.data_ptr(), fh, fw, 1, binary[i].data_ptr(), H, W, st))
            for f0 in range(0, n_frames, a.batch):
                n = min(a.batch, n_frames - f0)
                lib.check(lib.lm_stream_push(fs.handle, binary[f0:f0 + n].data_ptr(), n, labels.data_ptr(), st))
        else:
            main = torch.cuda.current_stream()
            for sx in side:
                sx.wait_stream(main)
            for i in range(n_frames):
                k = i % len(engines)
                engines[k].forward_raw(rgb[i].data_ptr(), H, W, logits[i].data_ptr(), None, None, st if k == 0 else side[k - 1].cuda_stream)
            for sx in side:
                main.wait_stream(sx)
            lib.check(lib.lm_stream_run_logits(fs.handle, logits.data_ptr(), n_frames, min(a.batch, n_frames), None, labels.data_ptr(), 128, 1, 0, st, st))
        gr = device.Grouping(fs, max_gap=85, min_times=3, t_window=5, min_recall=0.5, img_threshold=0.5, reconstruct=True)
        gr.render(0, n_frames, clean)
        torch.cuda.synchronize()
        sc = gr.array("scalars")
        gr.close()
        return sc

    try:
        once()
        t0 = time.perf_counter()
        reps = 2
        for _ in range(reps):
            sc = once()
        dt = (time.perf_counter() - t0) / reps
        k = fs.counters()
        return {"workload": "%d RGB %dx%d frames of one evolving synthetic whiteboard, resident in HBM: %sFCN (%s)%s -> threshold+invert -> %slabel -> records -> "
                            "matching -> step 03 -> %d reconstructed frames" % (n_frames, W, H, "LANCZOS 1/2 (device) -> " if big else "", a.fcn_precision,
                                                                                " at %dx%d" % (fw, fh) if big else "", "NEAREST x2 (device) -> " if big else "", n_frames),
                "frames": n_frames, "value": round(n_frames / dt, 2), "unit": "frames/s", "ms_per_frame": round(dt / n_frames * 1e3, 3),
                "fcn_engines_in_flight": 1 if big else len(engines),
                "stream": {"n_cc": k["n_cc"], "n_unique": k["n_unique"], "n_groups": int(sc[2])}}
    except _lib.LecturemathError as e:
        return {"error": str(e)}
    finally:
        fs.close()
        for e in engines[1:]:
            e.close()


def main_fcn(a):
    import torch
    from lecturemath_amd import _lib
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    lib = _lib.load()
    res, eng, _ = measure_fcn(a, lib, a.height, a.width, max(a.steps, 1), not a.no_fcn_oracle and a.cpu_frames > 0)
    fp32 = a.fcn_precision == "fp32"
    tfl = res["algorithmic_tflops"] if fp32 else res["executed_tflops"]
    peak = MFMA_F32_PEAK_TFLOPS if fp32 else MFMA_F16_PEAK_TFLOPS
    out = {"metric": "frames/sec FCN-LectureNet binarizer inference @1080p", "value": res["frames_per_s"], "unit": "frames/s", "n_gpus": 1,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": res["ms_per_frame"], "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32" if fp32 else a.fcn_precision + " operands, fp32 accumulate", "data": "synthetic",
           "config": {"workload": res["workload"], "gflop_per_frame": res["gflop_per_frame"]},
           "roofline": {"bound": "mfma", "kernel": "lm_fcn_forward[conv stack]", "achieved": tfl, "peak": peak, "unit": "TFLOP/s",
                        "frac": round(tfl / peak, 4), "traffic": None, "launch_ms": res["ms_per_frame"],
                        "algorithmic_tflops": res["algorithmic_tflops"],
                        "note": "achieved counts executed MFMA flops (%s per algorithmic flop)" % (1 if fp32 else res["mfma_per_product"])},
           "cpu_baseline": res.get("cpu_baseline"), "fcn": res}
    print(json.dumps(out))


# ----------------------------------------------------------------------------------------------------------------------
# configs[2] / configs[3]: the stream
# ----------------------------------------------------------------------------------------------------------------------
def make_logits(torch, synth, F, H, W, seed, f0, f1, chunk=250):
    """fp32 logits of frames [f0, f1) of the stream, resident in HBM (83 GB for 10,000 frames at 1080p): +-4 around the ink
    mask plus uniform noise, so that threshold+invert reproduces the generator's frame.  The generator
    is sequential, so every rank walks the whole stream and keeps its block."""
    logits = torch.empty((f1 - f0, H, W), dtype=torch.float32, device="cuda")
    gen = torch.Generator(device="cuda")
    buf = []

    def flush(first):
        if not buf:
            return
        mask = torch.from_numpy(np.stack(buf)).cuda()
        gen.manual_seed(1234 + first)
        dst = logits[first - f0:first - f0 + len(buf)]
        dst.copy_(torch.where(mask > 0, -4.0, 4.0))
        dst += torch.rand(dst.shape, generator=gen, device="cuda", dtype=torch.float32) - 0.5
        buf.clear()

    first = f0
    for t, fr in enumerate(synth.binary_stream(min(F, f1), H, W, seed=seed)):
        if t < f0:
            continue
        buf.append(fr)
        if len(buf) == chunk:
            flush(first)
            first = t + 1
    flush(first)
    return logits


# ----------------------------------------------------------------------------------------------------------------------
# N > 1: configs[3] (the logits stream, strong scaling) and the RGB stream (FCN on every rank, weak scaling)
# ----------------------------------------------------------------------------------------------------------------------
def main_sharded(a, torch, dist, lib, world, rank, rehearse, logits, gen_s):
    """One process per GPU.  Both workloads run through lecturemath_amd.sharded.ShardedStream: per-frame half on every rank in pieces,
    matching on rank 0 as the pieces arrive, the matched stream handed to rank 1 for step 03 + reconstruction (so matching of step
    i + 1 overlaps step 03 of step i).  `value` = configs[3]: the SAME 10,000-frame logits stream as N = 1, strong scaling -- bounded
    by its sequential halves (printed as `amdahl`); `rgb_sharded` = RGB frames -> FCN -> ... on every rank, weak scaling."""
    from lecturemath_amd import device, digests, fcn, sharded, synth
    H, W, F, B = a.height, a.width, a.frames, a.batch
    dev = "cpu" if rehearse else "cuda"
    pieces = int(os.environ.get("LM_BENCH_PIECES", "4"))
    s_wide = torch.cuda.Stream(priority=-1)
    s_match, s_back = torch.cuda.Stream(priority=0), torch.cuda.Stream(priority=0)
    labels = None if a.no_labels else torch.empty((B, H, W), dtype=torch.int32, device="cuda")

    def run(sh, logits_of, n_steps, n_total, timed):
        """n_steps steps; on the group rank every step ends with step 03 + all frames rendered.  Returns (seconds, info of the last step)"""
        clean = torch.empty((B, H, W), dtype=torch.uint8, device="cuda") if rank == sh.group_rank else None
        gr, rdone, info, marks = None, torch.cuda.Event(), None, []
        if timed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_steps):
            if gr is not None:                      # the previous step's tables lean on the stream object that is about to be reused
                rdone.synchronize()
                gr.close()
                gr = None
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(s_wide):
                e0.record(s_wide)
                gs = sh.step(logits_of, labels=labels, stream_wide=s_wide.cuda_stream, stream_match=s_match.cuda_stream)
                e1.record(s_wide)
            marks.append((e0, e1))
            if gs is not None:
                with torch.cuda.stream(s_back):
                    s_back.wait_stream(s_wide)
                    e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e2.record(s_back)
                    gr = device.Grouping(gs, max_gap=85, min_times=3, t_window=5, min_recall=0.5, img_threshold=0.5, reconstruct=True)
                    for f0 in range(0, n_total, B):
                        gr.render(f0, min(B, n_total - f0), clean[:min(B, n_total - f0)])
                    e3.record(s_back)
                    rdone.record(s_back)
                    marks[-1] = marks[-1] + (e2, e3)
                    info = {"scalars": [int(v) for v in gr.array("scalars")], "gs": gs, "gr": gr}
        if gr is not None:
            rdone.synchronize()
        sh.finish()
        torch.cuda.synchronize()
        if timed:
            dist.barrier()
        dt = time.perf_counter() - t0
        if timed:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        per_step = {"front_and_matching_ms": round(sum(m[0].elapsed_time(m[1]) for m in marks) / max(len(marks), 1), 3)}
        if marks and len(marks[-1]) == 4:
            per_step["step03_and_render_ms"] = round(sum(m[2].elapsed_time(m[3]) for m in marks) / len(marks), 3)
        return dt, info, per_step

    def collect(obj, src):
        box = [obj if rank == src else None]
        dist.broadcast_object_list(box, src=src)
        return box[0]

    # ---------------- configs[3]
    sh = sharded.ShardedStream(W, H, F, B, lib=lib, pieces=pieces)
    lib.check(lib.lm_ctx_set_profiling(sh.fs.labeler.ctx, 0))
    run(sh, lambda lo, hi: logits[lo:hi], max(a.warmup, 0), F, False)
    lib.check(lib.lm_ctx_set_profiling(sh.fs.labeler.ctx, 1))
    dt, info, per_step = run(sh, lambda lo, hi: logits[lo:hi], a.steps, F, True)
    ms, calls, nfr = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_int64(0)
    lib.check(lib.lm_ctx_profile_read(sh.fs.labeler.ctx, ctypes.addressof(ms), ctypes.addressof(calls), ctypes.addressof(nfr)))
    fused = bool(lib.lm_label_was_fused(sh.fs.labeler.ctx))
    parity = None
    if rank == sh.group_rank and info is not None:
        dg = digests.from_device(info["gs"], info["gr"])
        k1 = info["gs"].counters()
        parity = {"digests": dg, "counters": {k: k1[k] for k in ("n_cc", "n_unique", "tempo_count")}, "n_groups": info["scalars"][2], "reference": None, "match": None}
        ref, gname = golden_digests(W, H, a.seed, F)
        if ref:
            keys = [k for k in dg if k in ref]
            parity["reference"] = "%s[%d]" % (gname, F)
            parity["match"] = bool(all(dg[k] == ref[k] for k in keys) and ref["tempo_count"] == k1["tempo_count"] and ref["n_cc"] == k1["n_cc"])
        info["gr"].close()
    steps_all = [collect(per_step, r) for r in range(world)]
    parity = collect(parity, sh.group_rank)
    sh.close()
    # the CPU baseline is a property of the workload, not of N: rank 0 times the oracle on the frames it holds (the other ranks wait in
    # the next collective), after the timed region
    cpu = measure_cpu_baseline(logits, a.cpu_frames, 0, F, W, H) if rank == 0 and a.cpu_frames > 0 else None
    del logits
    torch.cuda.empty_cache()
    seq0 = steps_all[0]["front_and_matching_ms"]
    seq1 = steps_all[sh.group_rank].get("step03_and_render_ms", 0.0) + steps_all[sh.group_rank]["front_and_matching_ms"]
    amdahl = {"rank0_front_plus_matching_ms_per_step": seq0, "group_rank_front_plus_step03_ms_per_step": round(seq1, 3),
              "bound_frames_per_s": round(F / (max(seq0, seq1) * 1e-3), 1) if max(seq0, seq1) > 0 else None,
              "note": "the temporal matching replays on rank 0 (first-match-wins against first-seen masks, cc_stability_estimator.py:90-123) and step 03 "
                      "on rank %d: a step cannot be shorter than the longer of the two, however many ranks label" % sh.group_rank}

    # ---------------- RGB frames -> FCN -> ... on every rank (weak scaling)
    rgb_res = None
    if a.e2e_frames > 0:
        n_rank = a.e2e_frames
        n_tot = n_rank * world
        sd = synth.fcn_random_state_dict(synth.FCN_SHIPPED_WIDTHS, pixel_kernel=7, seed=0) if rank == 0 else None
        sd = sharded.broadcast_state_dict(sd, src=0, device_name="cuda")          # ONE contiguous buffer over RCCL
        eng = fcn.FcnEngine(synth.FCN_SHIPPED_WIDTHS, 7, 3, H, W, lib, precision=a.fcn_precision)
        eng.load_state_dict({k: v.cpu() for k, v in sd.items()})
        lo_r, hi_r = sharded.frame_range(n_tot, rank, world)
        rgb = torch.from_numpy(np.stack([f for t, f in enumerate(synth.whiteboard_stream(n_tot, H, W)) if lo_r <= t < hi_r])).cuda()
        lg = torch.empty((hi_r - lo_r, H, W), dtype=torch.float32, device="cuda")

        def fcn_logits(lo, hi):
            st = torch.cuda.current_stream().cuda_stream
            for i in range(lo, hi):
                eng.forward_raw(rgb[i].data_ptr(), H, W, lg[i].data_ptr(), None, None, st)
            return lg[lo:hi]
        sh2 = sharded.ShardedStream(W, H, n_tot, min(B, n_rank), lib=lib, pieces=pieces, max_ccs_per_frame=131072, max_words_per_frame=1 << 21)
        run(sh2, fcn_logits, 1, n_tot, False)
        dt2, info2, per2 = run(sh2, fcn_logits, 2, n_tot, True)
        k2 = None
        if rank == sh2.group_rank and info2 is not None:
            k2 = dict(info2["gs"].counters(), n_groups=info2["scalars"][2])
            info2["gr"].close()
        k2 = collect(k2, sh2.group_rank)
        sh2.close()
        eng.close()
        rgb_res = {"workload": "%d RGB %dx%d frames per rank (one evolving whiteboard of %d frames over %d ranks), resident in HBM: FCN (%s) -> threshold+invert -> "
                               "label -> records on the rank; matching on rank 0, step 03 + reconstruction on rank %d" % (n_rank, W, H, n_tot, world, a.fcn_precision, sh2.group_rank),
                   "value": round(n_tot * 2 / dt2, 2), "unit": "frames/s", "scaling": "weak", "frames_per_rank": n_rank, "steps": 2,
                   "ms_per_frame_per_rank": round(dt2 / 2 / n_rank * 1e3, 3), "stream": {k: k2[k] for k in ("n_cc", "n_unique", "n_groups")} if k2 else None,
                   "n1_reference": "e2e_rgb of the N = 1 line is the same pipeline on one rank"}

    if rank == 0:
        launch_ms = ms.value / max(calls.value, 1)
        fpl = nfr.value / max(calls.value, 1)
        bpp = (4 if fused else 1) + (4 if labels is not None else 0)
        ach = bpp * W * H * fpl / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
        traffic, traffic_note = label_traffic(W, H, fused, fpl)
        out = {"metric": "frames/sec end-to-end binarize+CC+group @1080p", "value": round(F * a.steps / dt, 2), "unit": "frames/s", "n_gpus": world,
               "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "config": {"workload": "configs[3]: the configs[2] stream (ONE synthetic %dx%d binary-board stream of %d frames per step, fp32 logits in HBM) frame-range "
                                      "sharded over %d ranks: threshold+label+records on the owning rank in %d pieces, temporal matching on rank 0 as the pieces arrive, "
                                      "grouping (step 03) + all %d frames reconstructed on rank 1" % (W, H, F, world, pieces, F),
                          "frames_per_step": F, "batch": B, "parallelism": "frame-range shards of one stream; point-to-point piece transfers to rank 0 (%s)" % ("gloo rehearsal on one GPU" if rehearse else "RCCL"),
                          "stages_not_in_timed_region": ["fcn conv stack (see rgb_sharded)"]},
               "roofline": {"bound": "hbm", "kernel": "labelling launches of rank 0", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_note": traffic_note, "launch_ms": round(launch_ms, 4),
                            "launches": calls.value, "frames_per_launch": round(fpl, 2), "algorithmic_bytes_per_px": bpp},
               "cpu_baseline": cpu, "parity": parity, "amdahl": amdahl, "per_rank_step_ms": steps_all, "rgb_sharded": rgb_res, "gen_seconds": round(gen_s, 2)}
        print(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()


def golden_digests(W, H, seed, F):
    """(digests of the reference / the oracle for this stream, file name) or (None, None): 1080p and 4K streams of the bench's generator"""
    name = {(1920, 1080): "g9_stream1080p_digests.json", (3840, 2160): "g9_stream4k_digests.json"}.get((W, H))
    if name is None or seed != 20213:
        return None, None
    path = os.path.join(ROOT, "tests", "golden", name)
    if not os.path.exists(path):
        return None, None
    return json.load(open(path)).get(str(F)), "tests/golden/" + name


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as a child torchrun BEFORE anything in this
    process touches the GPU (no torch import here), relay the child's output and exit with its code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # the ranks' stdout is relayed line by line: the JSON line to stdout, anything else a library printed there (gloo's connection
    # notes in the one-GPU rehearsal) to stderr, so that stdout carries the ONE line the contract asks for
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
        sys.stdout.flush()
    sys.exit(proc.wait())


def label_traffic(W, H, fused, frames_per_launch):
    """HBM bytes of one labelling launch from the committed rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE in separate runs of
    tools/label_microbench.py, gfx950 corrections applied by tools/pmc_traffic.py), scaled to this run's frames per launch.  PMC passes
    cannot be collected inside this process: the figure is QUOTED from the newest matching profile, not measured by this run."""
    import glob
    for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_label_traffic_pmc*.json")), reverse=True):
        tr = json.load(open(tpath))
        if (tr.get("width"), tr.get("height"), tr.get("fused")) == (W, H, fused) and tr.get("frames_per_launch"):
            return (int(tr["hbm_bytes_per_launch"] * frames_per_launch / tr["frames_per_launch"]),
                    "quoted from %s (separate --pmc passes of tools/label_microbench.py, %d frames per launch), scaled to this run's frames per launch; "
                    "not collected by this process" % (os.path.relpath(tpath, ROOT), tr["frames_per_launch"]))
    return None, "PMC passes cannot be collected inside this process and profiles/ holds none for this frame size"


def measure_cpu_baseline(logits, n, first_frame, F, W, H):
    """The CPU oracle (oracle/cc_oracle.c + oracle/grouping.py, ONE thread) on bounded samples of the stream `logits` holds (frames
    first_frame .. first_frame + len(logits) of an F-frame stream).  `value` = the same work as the bench's `value` -- steps 01-03 with all
    frames reconstructed -- on a DENSE window (the board is full around frame 5000) run as a stream of its own; `sparse_prefix` = steps
    01-02 on the first frames.  The checker, timed on the host after the timed region."""
    from oracle import cc as occ
    from oracle import grouping as ogr
    have = int(logits.shape[0])
    n = min(n, have)
    d0 = min(5000 - first_frame, have - n)          # window start inside `logits`
    lg = logits[:n].cpu().numpy()
    t0 = time.perf_counter()
    st = occ.Stability(W, H, 0.85, 0.85, 85)
    for i in range(n):
        st.add_frame(occ.threshold_invert(lg[i]))
    cdt = time.perf_counter() - t0
    sparse = {"value": round(n / cdt, 3), "unit": "frames/s", "cores": 1, "kind": "port",
              "sample": "frames %d..%d of the same stream: threshold+invert, label, stats, crops, temporal matching only (oracle/cc_oracle.c, "
                        "single thread)" % (first_frame, first_frame + n)}
    quoted = {"value": 3.7, "unit": "frames/s", "cores": 1, "quoted_not_measured": True, "source": "BASELINE.md section 4",
              "note": "the reference itself (steps 02 + 03, first 1,000 frames of this stream, 8-vCPU build container); the oracle runs step 02 "
                      "4.0x faster than the reference on the same frames"}
    if d0 <= 0:
        return dict(sparse, sparse_prefix=None, reference_in_build_container=quoted, note="stream too short for a dense window: steps 01-02 only")
    lg = logits[d0:d0 + n].cpu().numpy()
    t0 = time.perf_counter()
    st = occ.Stability(W, H, 0.85, 0.85, 85)
    for i in range(n):
        st.add_frame(occ.threshold_invert(lg[i]))
    t1 = time.perf_counter()
    ogr.run_step03(st.result(), max_gap=85, min_times=3, t_window=5, min_recall=0.5, img_threshold=0.5, reconstruct=True)
    t2 = time.perf_counter()
    return {"value": round(n / (t2 - t0), 3), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "frames %d..%d of the same stream as a stream of their own, the same work as `value`: steps 01-02 %.1f s (oracle/cc_oracle.c) + "
                      "step 03 with all frames reconstructed %.1f s (oracle/grouping.py, numpy); single thread, host has %d cores"
                      % (first_frame + d0, first_frame + d0 + n, t1 - t0, t2 - t1, os.cpu_count()),
            "sparse_prefix": sparse, "reference_in_build_container": quoted}


def main():
    a = parse()
    if a.workload == "fcn":
        return main_fcn(a)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(a)
    import concurrent.futures

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, "--gpus %d inside a launcher that set WORLD_SIZE=%d" % (a.gpus, world)
    # LM_BENCH_REHEARSE=1: all ranks on cuda:0 with the gloo backend -- exercises the multi-process path on a one-GPU box
    rehearse = bool(os.environ.get("LM_BENCH_REHEARSE"))
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)               # before the process group: RCCL binds to the current device
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from lecturemath_amd import _lib, device, digests, sharded, synth
    lib = _lib.load()
    assert lib.is_device_build

    H, W, F, B = a.height, a.width, a.frames, a.batch
    f_lo, f_hi = sharded.frame_range(F, rank, world)
    n_mine = f_hi - f_lo
    t0 = time.time()
    logits = make_logits(torch, synth, F, H, W, a.seed, f_lo, f_hi)
    torch.cuda.synchronize()
    gen_s = time.time() - t0

    if world > 1:
        return main_sharded(a, torch, dist, lib, world, rank, rehearse, logits, gen_s)

    # capacities of a stream (records 32 B, crop words 4 B; sized from the generator's densities with headroom -- a capacity
    # error is raised by the library, never silent)
    cap_frames = F if rank == 0 else max(n_mine, 1)
    max_ccs = cap_frames * 4096
    max_words = cap_frames * max(1 << 17, (W * H) // 16)

    depth = max(1, a.depth) if world == 1 else 1
    prio = os.environ.get("LM_BENCH_PRIO", "front")
    split = not os.environ.get("LM_BENCH_NO_SPLIT")
    # The bandwidth-bound half of steps 01-02 (threshold, labelling, records: "wide") of ALL slots goes through ONE high-priority
    # HIP stream, so two labelling launches never share the GPU; the temporal matching of a slot (small latency-bound kernels)
    # runs on the slot's own normal-priority stream behind an event per batch, under the wide kernels of the next batches.
    # (The streams must differ in priority: with both high the runtime puts them on one hardware queue and nothing overlaps.)
    s_wide = torch.cuda.Stream(priority=-1 if prio == "front" else 0)
    back_prio = int(os.environ.get("LM_BENCH_BACK_PRIO", "0"))      # priority of the step-03 streams
    match_prio = int(os.environ.get("LM_BENCH_MATCH_PRIO", "0"))    # ... of the matching streams
    slots = []
    for _ in range(depth):
        fs = device.FrameStream(W, H, cap_frames, 0.85, 0.85, 85, 20, max_batch=B, max_ccs=max_ccs, max_crop_words=max_words, lib=lib)
        slots.append({"fs": fs, "clean": torch.empty((B, H, W), dtype=torch.uint8, device="cuda"),
                      "s_match": torch.cuda.Stream(priority=match_prio), "s_back": torch.cuda.Stream(priority=back_prio),
                      "done": torch.cuda.Event(), "rdone": torch.cuda.Event(), "gr": None, "recorded": []})
    # binary frames and the label image live for one batch: written by threshold / the labeller, consumed in order on s_wide
    binary = torch.empty((B, H, W), dtype=torch.uint8, device="cuda")
    labels = None if a.no_labels else torch.empty((B, H, W), dtype=torch.int32, device="cuda")
    pool = concurrent.futures.ThreadPoolExecutor(max_workers=max(1, depth))

    schedule = os.environ.get("LM_BENCH_SCHEDULE", "free")        # free (default) | gated | gated-py (the Python-driven loop below)
    gated = schedule == "gated-py"

    def front(sl, match=True, use_split=split):
        """steps 01 (threshold) + 02 (label, records; matching when `match`) of this rank's block of the stream.
        Schedule: "free" (default) hands the whole loop to lm_stream_run_logits, matching of batch k behind batch k's records on its
        own queue.  LM_BENCH_SCHEDULE=gated: the labelling launches -- the bandwidth-bound kernels the roofline is quoted on -- never
        share the GPU with the temporal matching: matching of batch k-1 starts when batch k has been labelled and runs under batch
        k's statistics / record emission; batch k+1 is labelled when it has finished.  Round 3, measured on one lease with the final
        kernels (profiles/r03_s2_operating_points.txt): free + one label part 68.3 k frames/s at 0.30 of the HBM peak in-region,
        gated + two parts 58.6 k at 0.39 -- the gaps of the labelling middle are where the rest of the pipeline runs."""
        fs = sl["fs"]
        ctx = fs.labeler.ctx
        with torch.cuda.stream(s_wide):
            ws, ms = s_wide.cuda_stream, sl["s_match"].cuda_stream
            if sl["gr"] is not None:                # the previous step of this slot: its rendering must be done before
                sl["rdone"].synchronize()           # its tables go away and its buffers are reused
                sl["gr"].close()
                sl["gr"] = None
            fs.reset()
            lp = labels.data_ptr() if labels is not None else None
            if not gated:
                # the whole launch loop in one library call (per batch: threshold -> label -> records on the wide stream, matching
                # on the slot's own stream behind an event): ~5,000 launches that a Python loop issued at half the GPU's pace
                two = match and use_split
                lib.check(lib.lm_stream_run_logits(fs.handle, logits.data_ptr(), n_mine, B, binary.data_ptr(), lp, 128, 1 if match else 0,
                                                   1 if schedule == "gated" else 0, ws, ms if two else ws))
                sl["done"].record(sl["s_match"] if two else s_wide)
                return
            batches = [(f0, min(B, n_mine - f0)) for f0 in range(0, n_mine, B)]
            ev = sl["recorded"]
            while len(ev) < 3 * len(batches):
                ev.append(torch.cuda.Event())
            for k, (f0, n) in enumerate(batches):
                evL, evR, evM = ev[3 * k], ev[3 * k + 1], ev[3 * k + 2]
                lib.check(lib.lm_threshold_invert(logits[f0:f0 + n].data_ptr(), binary.data_ptr(), n * H * W, 128, ws))
                if match and not use_split:
                    lib.check(lib.lm_stream_push(fs.handle, binary.data_ptr(), n, lp, ws))
                    continue
                if match and gated and k >= 2:
                    s_wide.wait_event(ev[3 * (k - 2) + 2])          # matching of batch k-2 has left the GPU
                lib.check(lib.lm_label_batch(ctx, binary.data_ptr(), n, lp, ws))
                evL.record(s_wide)
                lib.check(lib.lm_stream_push_labelled(fs.handle, n, ws))
                evR.record(s_wide)
                if not match:
                    continue
                if gated:
                    if k >= 1:                                      # matching of batch k-1 under batch k's records
                        sl["s_match"].wait_event(evL)
                        lib.check(lib.lm_stream_match(fs.handle, batches[k - 1][1], ms))
                        ev[3 * (k - 1) + 2].record(sl["s_match"])
                else:
                    sl["s_match"].wait_event(evR)
                    lib.check(lib.lm_stream_match(fs.handle, n, ms))
            if match and use_split and gated and batches:
                sl["s_match"].wait_event(ev[3 * (len(batches) - 1) + 1])
                lib.check(lib.lm_stream_match(fs.handle, batches[-1][1], ms))
            if match and use_split:
                sl["done"].record(sl["s_match"])
            else:
                sl["done"].record(s_wide)

    def gather_and_match(sl):
        """N > 1: the packed blocks of ranks 1.. travel to rank 0 (one transfer per rank), which appends and matches them"""
        fs = sl["fs"]
        with torch.cuda.stream(sl["s_match"]):
            sl["s_match"].wait_event(sl["done"])
            if rank != 0:
                blk = fs.pack(0, n_mine)
                torch.cuda.current_stream().synchronize()
                sharded.gather_blocks(blk, dst=0)
                return
            blocks = sharded.gather_blocks(torch.empty(32, dtype=torch.uint8, device="cuda"), dst=0)
            for r in range(1, world):
                fs.append_packed(blocks[r])
                r0, r1 = sharded.frame_range(F, r, world)
                fs.match(r1 - r0)
            sl["done"].record(sl["s_match"])

    def back(sl):
        """step 03: grouping + reconstruction of every frame (frames_from_groups), rendered batch by batch"""
        torch.cuda.set_device(local_rank)
        tq = time.perf_counter()
        with torch.cuda.stream(sl["s_back"]):
            sl["s_back"].wait_event(sl["done"])
            if os.environ.get("LM_BENCH_VERBOSE"):
                sl["done"].synchronize()
                sys.stderr.write("   steps 01-02 finished on the GPU %.1f ms after step 03 was submitted\n" % ((time.perf_counter() - tq) * 1e3))
                tq = time.perf_counter()
            gr = device.Grouping(sl["fs"], max_gap=85, min_times=3, t_window=5, min_recall=0.5, img_threshold=0.5, reconstruct=True)
            for f0 in range(0, F, B):
                n = min(B, F - f0)
                gr.render(f0, n, sl["clean"][:n])
            info = gr.array("scalars")
            sl["rdone"].record(sl["s_back"])        # rendering is only enqueued here; awaited before the slot is reused / at the end
            sl["gr"] = gr
            if os.environ.get("LM_BENCH_VERBOSE"):
                t1 = time.perf_counter()
                sl["rdone"].synchronize()
                sys.stderr.write("   step 03: tables %.1f ms, rendering done %.1f ms later\n" % ((t1 - tq) * 1e3, (time.perf_counter() - t1) * 1e3))
        return info

    def run_steps(k):
        pending = [None] * depth
        info = None
        verbose = bool(os.environ.get("LM_BENCH_VERBOSE"))
        for i in range(k):
            sl = slots[i % depth]
            ta = time.perf_counter()
            if pending[i % depth] is not None:
                info = pending[i % depth].result()          # the slot's previous step must be finished before it is reused
            tb = time.perf_counter()
            front(sl, match=(rank == 0))
            tc = time.perf_counter()
            if world > 1:
                gather_and_match(sl)
            if rank == 0:
                pending[i % depth] = pool.submit(back, sl)
            if verbose:
                sys.stderr.write("rank %d step %d: waited %.1f ms for its slot, enqueued steps 01-02 in %.1f ms, gather %.1f ms\n"
                                 % (rank, i, (tb - ta) * 1e3, (tc - tb) * 1e3, (time.perf_counter() - tc) * 1e3))
        for p in pending:
            if p is not None:
                info = p.result()
        for sl in slots:
            if sl["gr"] is not None:
                sl["rdone"].synchronize()
        return info

    def close_groups():
        for sl in slots:
            if sl["gr"] is not None:
                sl["rdone"].synchronize()
                sl["gr"].close()
                sl["gr"] = None

    # Set-up that is not a property of the steady state: every pipeline slot sizes its step-03 arena by its first pass and re-allocates
    # it after that pass, so a slot is warm after two passes.  When --warmup asks for fewer than 2 x depth untimed steps, the missing ones
    # are run here, untimed as well, and reported as `untimed_priming_steps`.
    priming = max(0, 2 * depth - max(a.warmup, 0)) if world == 1 else 0
    run_steps(priming + max(a.warmup, 0))
    torch.cuda.synchronize()
    k0 = slots[0]["fs"].counters() if a.warmup > 0 else None      # also surfaces capacity errors before timing
    for sl in slots:
        lib.check(lib.lm_ctx_set_profiling(sl["fs"].labeler.ctx, 1))

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ginfo = run_steps(a.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cpu" if rehearse else "cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- roofline of the labelling launch sequence (events recorded inside the timed region)
    tot_ms, tot_calls, tot_fr = 0.0, 0, 0
    for sl in slots:
        ms, calls, nfr = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_int64(0)
        lib.check(lib.lm_ctx_profile_read(sl["fs"].labeler.ctx, ctypes.addressof(ms), ctypes.addressof(calls), ctypes.addressof(nfr)))
        lib.check(lib.lm_ctx_set_profiling(sl["fs"].labeler.ctx, 0))
        tot_ms += ms.value
        tot_calls += calls.value
        tot_fr += nfr.value

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    k1 = slots[0]["fs"].counters()
    assert k0 is None or k1 == k0, "steps are not reproducible: %r vs %r" % (k0, k1)

    # ---- parity: digests of what the last timed step of every slot left, against the reference's digests of the same stream
    used = slots[:min(depth, max(a.steps, 1))]
    dg = [digests.from_device(sl["fs"], sl["gr"]) for sl in used if sl["gr"] is not None]
    sums = []
    if used and used[0]["gr"] is not None:
        sl = used[0]
        with torch.cuda.stream(sl["s_back"]):
            for f0 in range(0, F, B):
                n = min(B, F - f0)
                sl["gr"].render(f0, n, sl["clean"][:n])
                sums.append(device.frame_sums(sl["clean"][:n], lib))
        if dg:
            dg[0]["clean_frame_sums"] = digests.sums_digest(np.concatenate(sums))
    def core(d):
        return {k: v for k, v in d.items() if k != "clean_frame_sums"}

    parity = {"digests": dg[0] if dg else None, "all_slots_identical": all(core(d) == core(dg[0]) for d in dg),
              "counters": {k: k1[k] for k in ("n_cc", "n_unique", "tempo_count")}, "reference": None, "match": None}
    ref, gname = golden_digests(W, H, a.seed, F) if dg else (None, None)
    if ref:
        keys = [k for k in dg[0] if k in ref]
        parity["reference"] = ("%s[%d] (%s run on the same stream in the build container)"
                               % (gname, F, "the oracle, itself pinned to the reference on the first 1,000 frames of the 1080p stream," if ref.get("produced_by") == "oracle" else "the reference"))
        parity["match"] = bool(all(dg[0][k] == ref[k] for k in keys) and ref["tempo_count"] == k1["tempo_count"] and ref["n_cc"] == k1["n_cc"])
        parity["compared"] = keys + ["tempo_count", "n_cc"]
    if parity["match"] is False or not parity["all_slots_identical"]:
        sys.stderr.write("bench.py: digests differ (reference match: %r, slots identical: %r) -- the rate below is NOT a valid result\n"
                         % (parity["match"], parity["all_slots_identical"]))
        if os.environ.get("LM_BENCH_STRICT"):
            raise AssertionError("parity digests differ")

    # the same launch sequence once more with nothing else in flight (after the timed region): in the pipeline, kernels of the
    # matching stream share the GPU with it
    alone_ms = None
    if world == 1 and not os.environ.get("LM_BENCH_NO_ALONE"):
        close_groups()
        sl = slots[0]
        lib.check(lib.lm_ctx_set_profiling(sl["fs"].labeler.ctx, 1))
        front(sl, match=False)
        torch.cuda.synchronize()
        ms2, calls2, nfr2 = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_int64(0)
        lib.check(lib.lm_ctx_profile_read(sl["fs"].labeler.ctx, ctypes.addressof(ms2), ctypes.addressof(calls2), ctypes.addressof(nfr2)))
        lib.check(lib.lm_ctx_set_profiling(sl["fs"].labeler.ctx, 0))
        if calls2.value > 0:
            alone_ms = ms2.value / calls2.value         # the same mix of launches as one timed step
    close_groups()

    launch_ms = tot_ms / max(tot_calls, 1)
    frames_per_launch = tot_fr / max(tot_calls, 1)
    # The timed launch is lm_label_batch_logits: fp32 logits in (4 B/px), int32 labels out (4 B/px) -- the threshold is fused into the
    # row packing, so the byte frame SURVEY 8(d)'s 5 B/px figure counts (1 B in + 4 B out) is never written nor read.  Both accountings
    # are reported: `frac` on the launch's own algorithmic bytes, `frac_survey_5Bpx` on SURVEY's figure over the same (longer) launch.
    fused = bool(lib.lm_label_was_fused(slots[0]["fs"].labeler.ctx)) if slots else False
    bpp = (4 if fused else 1) + (4 if labels is not None else 0)
    algo_bytes = bpp * W * H * frames_per_launch
    achieved = algo_bytes / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
    kern = ("lm_label_batch_logits[lm_k_pack_rows_logits+lm_k_band+lm_k_seam_union+lm_k_flatten_flag+lm_k_apply_labels+lm_k_write_labels]" if fused else
            "lm_label_batch[lm_k_pack_rows+lm_k_band+lm_k_seam_union+lm_k_flatten_flag+lm_k_apply_labels+lm_k_write_labels]")
    roofline = {"bound": "hbm", "kernel": kern,
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "algorithmic_bytes_per_px": {"in": 4 if fused else 1, "out": 4 if labels is not None else 0,
                                             "note": ("fp32 logit read + int32 label written; threshold fused, no byte frame" if fused else
                                                      "uint8 frame read + int32 label written (SURVEY 8(d))")},
                "frac_survey_5Bpx": round(ALGO_BYTES_PER_PX * W * H * frames_per_launch / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if launch_ms > 0 else None,
                "traffic": None, "traffic_note": None,
                "launch_ms": round(launch_ms, 4), "launches": tot_calls, "frames_per_launch": round(frames_per_launch, 2),
                "algorithmic_bytes_per_launch": int(algo_bytes), "label_image_written": labels is not None,
                "timed": "HIP events around every labelling launch sequence inside the timed region"}
    roofline["traffic"], roofline["traffic_note"] = label_traffic(W, H, fused, frames_per_launch)
    if alone_ms:
        roofline["alone"] = {"launch_ms": round(alone_ms, 4), "achieved": round(algo_bytes / (alone_ms * 1e-3) / 1e9, 2),
                             "frac": round(algo_bytes / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "note": "same launches after the timed region, no matching kernels in flight"}
    # The operating point is chosen for `value`: in the timed region the labelling launches share the GPU with the matching kernels
    # and with step 03 of the previous step, which fill the gaps of the launch's latency-bound middle -- `frac` measures the launch
    # while it shares.  The schedule that keeps the launch to itself is a switch away and is measured in profiles/:
    roofline["operating_point"] = {
        "this_run": {"schedule": schedule, "label_parts": int(os.environ.get("LM_LABEL_PARTS", "1"))},
        "exclusive_alternative": {"how": "LM_BENCH_SCHEDULE=gated LM_LABEL_PARTS=2", "quoted_not_measured": True, "frac": 0.388, "value_frames_per_s": 58600,
                                  "source": "profiles/r03_s2_operating_points.txt (round 3's library on one lease; not re-measured by this run)"}}

    # ---- CPU baseline (the checker, timed after the timed region; rank 0 only)
    cpu = measure_cpu_baseline(logits, a.cpu_frames, 0, F, W, H) if a.cpu_frames > 0 else None

    out = {
        "metric": "frames/sec end-to-end binarize+CC+group @1080p", "value": round(F * a.steps / dt, 2), "unit": "frames/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "untimed_priming_steps": priming, "ms_per_step": round(dt / a.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak" if world == 1 else "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "configs[%d]: ONE synthetic %dx%d binary-board stream of %d frames per step%s: fp32 logits in HBM -> threshold+invert -> "
                               "CC label (int32 image) -> CC stats/records/crops -> temporal matching over the whole stream -> grouping (step 03) + all "
                               "%d frames reconstructed" % (2 if world == 1 else 3, W, H, F,
                                                            "" if world == 1 else ", frame-range sharded over %d ranks (records gathered to rank 0)" % world, F),
                   "frames_per_step": F, "frames_per_stream": F, "batch": B,
                   "stages_not_in_timed_region": ["fcn conv stack (logits are synthetic, SURVEY 8(d) config 3; measured separately in `fcn` / `e2e_rgb`)"],
                   "stream": dict({k: k1[k] for k in ("n_cc", "n_unique", "tempo_count")}, n_groups=int(ginfo[2]), n_split=int(ginfo[0])),
                   "parallelism": "one stream on one GPU" if world == 1 else "frame-range shards of one stream, gather to rank 0",
                   "pipeline_depth": depth, "matching_on_own_hip_stream": bool(split),
                   "schedule": {"gated": "lm_stream_run_logits schedule 1: labelling launches kept apart from the matching's wide kernels",
                                "free": "lm_stream_run_logits schedule 0: matching of a batch behind its records on its own queue, nothing kept apart",
                                "gated-py": "Python-driven gated loop"}.get(schedule, schedule)},
        "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "gen_seconds": round(gen_s, 2),
    }
    if world == 1 and a.fcn_frames > 0:
        del logits
        for sl in slots:
            sl["fs"].close()
        torch.cuda.empty_cache()
        res, eng, fsd = measure_fcn(a, lib, H, W, a.fcn_frames, not a.no_fcn_oracle)
        out["fcn"] = res
        if a.e2e_frames > 0:
            out["e2e_rgb"] = measure_e2e_rgb(a, lib, eng, H, W, a.e2e_frames, sd=fsd)
        eng.close()
    print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
