#!/usr/bin/env python3
"""bench.py -- frames/sec of the hot path (binarize-threshold + CC labelling + CC records + temporal matching
+ space-time grouping with frame reconstruction) on a synthetic 1080p stream, per BASELINE.json (configs[2], the configuration the
metric is quoted on: the synthetic 1080p stream on one MI355X; `--frames 10000` is its full length).

A "step" is one pass of the hot path over one synthetic stream of --frames frames whose fp32 logits are
already resident in HBM.  With N > 1 (launched by torch.distributed.run, one rank per GPU) every rank
processes an independent stream of the same shape (the reference's outer loop is over independent
lectures, console_ui_process.py:121-148), no data-path collective: weak scaling.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     the CC-labelling launch sequence (lm_label_batch: pack, rowscan, rowoff, union, resolve,
               write_labels), HBM bound: achieved = 5 B/px * W*H * frames_per_launch / mean launch duration,
               timed live with HIP events on the launching stream inside the timed region
  cpu_baseline the oracle (C port of the reference path, single thread) on a bounded prefix of the same stream
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
ALGO_BYTES_PER_PX = 5           # 1 B uint8 in + 4 B int32 label out


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=40,
                   help="timed steps; a step is one 256-frame stream, so the default run covers the 10k frames of BASELINE configs[2]")
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--frames", type=int, default=256, help="frames per stream (= per step)")
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--batch", type=int, default=64, help="frames per labelling launch")
    p.add_argument("--cpu-frames", type=int, default=200, help="prefix of the stream timed on the CPU oracle (0 = skip)")
    p.add_argument("--no-labels", action="store_true", help="do not materialise the int32 label image")
    p.add_argument("--no-pipeline", action="store_true", help="do not overlap step 03 of one stream with steps 01-02 of the next")
    p.add_argument("--depth", type=int, default=5, help="streams in flight (pipeline slots); depth-1 host workers run step 03")
    p.add_argument("--seed", type=int, default=20213)
    p.add_argument("--fcn-precision", default="f16x3", choices=["f16x3", "fp32"],
                   help="MFMA operand format of the FCN conv stack (fp32 accumulate in both)")
    p.add_argument("--workload", default="stream", choices=["stream", "fcn"],
                   help="stream = configs[2] (headline metric); fcn = configs[1], FCN-LectureNet inference on one 1080p frame")
    return p.parse_args()


MFMA_F32_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32 matrix peak
MFMA_F16_PEAK_TFLOPS = 2500.0   # dense f16/bf16 MFMA peak (spec); the f16x3 path executes 3 MFMA flops per algorithmic flop


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


def main_fcn(a):
    import torch
    from lecturemath_amd import _lib, fcn, synth
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    lib = _lib.load()
    H, W = a.height, a.width
    widths, pk = synth.FCN_SHIPPED_WIDTHS, 7
    sd = synth.fcn_random_state_dict(widths, pixel_kernel=pk, seed=0)
    eng = fcn.FcnEngine(widths, pk, 3, H, W, lib, precision=a.fcn_precision)
    eng.load_state_dict(sd)
    rgb, _ = synth.whiteboard_rgb(H, W, 1500, seed=20211)
    d_rgb = torch.from_numpy(rgb).cuda()
    lab = None
    for _ in range(a.warmup):
        eng.forward(d_rgb)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(a.steps):
        out, text, rec = eng.forward(d_rgb)
    e1.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gpu_ms = e0.elapsed_time(e1) / a.steps
    fl = fcn_flops(widths, pk, 3, H, W)
    tflops = fl / (gpu_ms * 1e-3) / 1e12
    cpu = None
    if a.cpu_frames > 0:
        from oracle import fcn as ofcn          # the checker / CPU baseline leg only
        torch.set_num_threads(os.cpu_count())
        t0 = time.perf_counter()
        with torch.no_grad():
            o, t, r = ofcn.forward(sd, ofcn.prepare_image(rgb))
        cdt = time.perf_counter() - t0
        err = float((out.cpu() - o[0, 0]).abs().max())
        cpu = {"value": round(1.0 / cdt, 4), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": "one 1080p frame, oracle/fcn.py (torch fp32 CPU functional restatement); max |logit diff| vs HIP = %.2e" % err}
    out = {
        "metric": "frames/sec FCN-LectureNet binarizer inference @1080p", "value": round(a.steps / dt, 3), "unit": "frames/s", "n_gpus": 1,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32" if a.fcn_precision == "fp32" else "f16x3 (fp16-split operands, fp32 accumulate)", "data": "synthetic",
        "config": {"workload": "configs[1]: FCN-LectureNet (shipped widths, 15.8 M params, random init + randomised BN) forward on one "
                               "%dx%d synthetic whiteboard frame" % (W, H), "gflop_per_frame": round(fl / 1e9, 1)},
        "roofline": ({"bound": "mfma", "kernel": "lm_fcn_forward[lm_k_conv_mfma + heads]", "achieved": round(tflops, 2),
                      "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tflops / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None,
                      "launch_ms": round(gpu_ms, 3)} if a.fcn_precision == "fp32" else
                     {"bound": "mfma", "kernel": "lm_fcn_forward[lm_k_conv_mfma_h + heads]", "achieved": round(3 * tflops, 2),
                      "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(3 * tflops / MFMA_F16_PEAK_TFLOPS, 4), "traffic": None,
                      "launch_ms": round(gpu_ms, 3), "algorithmic_tflops": round(tflops, 2),
                      "note": "achieved counts the executed f16 MFMA flops (3 per algorithmic flop: hi.hi + hi.lo + lo.hi)"}),
        "cpu_baseline": cpu}
    # MFMA-pipe utilisation from the PMC pass committed under profiles/ (not collectable live): busy cycles of the matrix pipe
    # over the conv-stack dispatches, as opposed to `frac` above, which prices useful flops against the dense peak
    ppath = os.path.join(ROOT, "profiles", "r01_fcn_mfma_pmc_%s.json" % a.fcn_precision)
    if os.path.exists(ppath) and (W, H) == (1920, 1080):
        out["roofline"]["mfma_util_pmc"] = json.load(open(ppath))["conv_stack_mfma_util"]
        out["roofline"]["mfma_util_pmc_source"] = "profiles/" + os.path.basename(ppath) + " (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE)"
    print(json.dumps(out))


def main():
    a = parse()
    if a.workload == "fcn":
        return main_fcn(a)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, "launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (a.gpus, world)
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
    from lecturemath_amd import _lib, device, synth
    lib = _lib.load()
    assert lib.is_device_build

    H, W, F = a.height, a.width, a.frames
    # ---- synthetic stream (host, numpy) -> logits resident in HBM
    t0 = time.time()
    frames_host = np.stack(list(synth.binary_stream(F, H, W, seed=a.seed + rank)))
    mask = torch.from_numpy(frames_host).cuda()
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1234 + rank)
    logits = torch.where(mask > 0, -4.0, 4.0).to(torch.float32)
    logits += (torch.rand(logits.shape, generator=gen, device="cuda", dtype=torch.float32) - 0.5)
    del mask
    gen_s = time.time() - t0

    # Pipeline slots (--depth): while some slots' streams are in step 03 (host list bookkeeping + group images + frame rendering,
    # on their own HIP streams, driven by worker threads -- ctypes releases the GIL), the next slots label and match the next
    # streams.  Every step still does all of its work inside the timed region; only consecutive, independent steps overlap.
    import concurrent.futures
    depth = 1 if a.no_pipeline else max(2, a.depth)
    prio = os.environ.get("LM_BENCH_PRIO", "front")       # which side's HIP streams get the higher priority: front | back | none
    # unless LM_BENCH_NO_SPLIT is set the slot's own stream only carries the matching kernels: normal priority (see `split` below)
    front_prio = -1 if (prio == "front" and (depth == 1 or os.environ.get("LM_BENCH_NO_SPLIT"))) else 0
    slots = []
    for _ in range(depth):
        fs = device.FrameStream(W, H, F, 0.85, 0.85, 85, 20, max_batch=a.batch, lib=lib)
        slots.append({"fs": fs, "binary": torch.empty((F, H, W), dtype=torch.uint8, device="cuda"),
                      "labels": None if a.no_labels else torch.empty((a.batch, H, W), dtype=torch.int32, device="cuda"),
                      "clean": torch.empty((a.batch, H, W), dtype=torch.uint8, device="cuda"),
                      # steps 01-02 are the bandwidth-bound part: their stream gets the higher priority, step 03's small kernels fill in
                      "s_front": torch.cuda.Stream(priority=front_prio), "s_back": torch.cuda.Stream(priority=-1 if prio == "back" else 0),
                      "done": torch.cuda.Event(),
                      "rdone": torch.cuda.Event(), "gr": None, "recorded": []})
    pool = concurrent.futures.ThreadPoolExecutor(max_workers=max(1, depth - 1))

    # The bandwidth-bound half of steps 01-02 (threshold, labelling, records: "wide") of ALL slots goes through ONE high-priority
    # HIP stream, so two labelling launches never share the GPU; the temporal matching of a slot (small latency-bound kernels,
    # among them a single-workgroup replay) runs on the slot's own normal-priority stream behind an event per batch, under the
    # wide kernels of the next batches / the next steps.  (The streams must differ in priority: with both high the runtime puts
    # them on one hardware queue and nothing overlaps -- 62-71 k frames/s; as below 73-76 k at depth 5.)  LM_BENCH_NO_SPLIT=1:
    # every slot's steps 01-02 on its own high-priority stream (67-71 k at depth 3, with dips to 50-55 k).
    split = depth > 1 and not os.environ.get("LM_BENCH_NO_SPLIT")
    s_wide = torch.cuda.Stream(priority=-1 if prio == "front" else 0) if split else None

    def front(sl, split=split):
        """steps 01 (threshold) + 02 (label, records, matching) of one stream"""
        fs, binary, labels = sl["fs"], sl["binary"], sl["labels"]
        s_rec = s_wide if split else sl["s_front"]
        with torch.cuda.stream(s_rec):
            stream = s_rec.cuda_stream
            if sl.get("gr") is not None:            # the previous step of this slot: its rendering must be done before
                sl["rdone"].synchronize()           # its tables go away and its buffers are reused
                sl["gr"].close()
                sl["gr"] = None
            fs.reset()
            lib.check(lib.lm_threshold_invert(logits.data_ptr(), binary.data_ptr(), F * H * W, 128, stream))
            for k, f0 in enumerate(range(0, F, a.batch)):
                n = min(a.batch, F - f0)
                # the label image of a batch is an output of the labelling kernel; the same buffer is reused per batch
                if not split:
                    lib.check(lib.lm_stream_push(fs.handle, binary[f0:f0 + n].data_ptr(), n,
                                                 labels.data_ptr() if labels is not None else None, stream))
                    continue
                lib.check(lib.lm_stream_push_records(fs.handle, binary[f0:f0 + n].data_ptr(), n,
                                                     labels.data_ptr() if labels is not None else None, stream))
                while len(sl["recorded"]) <= k:
                    sl["recorded"].append(torch.cuda.Event())
                sl["recorded"][k].record(s_rec)
                sl["s_front"].wait_event(sl["recorded"][k])
                lib.check(lib.lm_stream_match(fs.handle, n, sl["s_front"].cuda_stream))
            sl["done"].record(sl["s_front"])

    def back(sl):
        """step 03: grouping + reconstruction of every frame (frames_from_groups), rendered batch by batch"""
        torch.cuda.set_device(local_rank)
        tq = time.perf_counter()
        with torch.cuda.stream(sl["s_back"]):
            sl["s_back"].wait_event(sl["done"])
            gr = device.Grouping(sl["fs"], max_gap=85, min_times=3, t_window=5, min_recall=0.5, img_threshold=0.5, reconstruct=True)
            for f0 in range(0, F, a.batch):
                n = min(a.batch, F - f0)
                gr.render(f0, n, sl["clean"][:n])
            info = gr.array("scalars")
            # rendering is only enqueued here: the worker goes on to the next stream's step 03 while the GPU draws
            sl["rdone"].record(sl["s_back"])
            if os.environ.get("LM_BENCH_SYNC_RENDER"):
                sl["s_back"].synchronize()
                gr.close()
            else:
                sl["gr"] = gr
        if os.environ.get("LM_BENCH_VERBOSE"):
            sys.stderr.write("   step 03 worker: %.3f ms (from %.3f ms)\n" % ((time.perf_counter() - tq) * 1e3, tq * 1e3))
        return info

    max_fronts = int(os.environ.get("LM_BENCH_MAX_FRONTS", "0"))

    def run_steps(k):
        pending = [None] * depth
        info = None
        for i in range(k):
            ta = time.perf_counter()
            sl = slots[i % depth]
            if pending[i % depth] is not None:
                info = pending[i % depth].result()          # the slot's previous step must be finished before it is reused
            if max_fronts and i >= max_fronts:           # at most max_fronts steps' 01-02 halves queued on the GPU
                slots[(i - max_fronts) % depth]["done"].synchronize()
            tb = time.perf_counter()
            front(sl)
            pending[i % depth] = pool.submit(back, sl)
            if os.environ.get("LM_BENCH_VERBOSE"):
                sys.stderr.write("step %d: at %.3f ms, waited %.3f ms for its slot, enqueued steps 01-02 in %.3f ms\n"
                                 % (i, ta * 1e3, (tb - ta) * 1e3, (time.perf_counter() - tb) * 1e3))
        for p in pending:
            if p is not None:
                info = p.result()
        for sl in slots:
            if sl["gr"] is not None:
                sl["rdone"].synchronize()
                sl["gr"].close()
                sl["gr"] = None
        return info

    def stream_digest(fs):
        """sha1 over everything steps 01-02 leave in the stream: CC records with their unique assignments, frame offsets, bit
        crops, active list"""
        import hashlib
        r = fs.read()
        h = hashlib.sha1()
        for key in ("rec", "frame_off", "crop_off", "crop", "active"):
            h.update(np.ascontiguousarray(r[key]).tobytes())
        return h.hexdigest()

    # one step with nothing else in flight, on one stream: the reference result for the steps that overlap in the pipeline
    front(slots[0], split=False)
    torch.cuda.synchronize()
    digest_alone = stream_digest(slots[0]["fs"])
    run_steps(max(a.warmup, 0))
    torch.cuda.synchronize()
    fs = slots[0]["fs"]
    labels = slots[0]["labels"]
    k0 = fs.counters() if a.warmup > 0 else None      # also surfaces capacity errors before timing
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
    import ctypes
    tot_ms, tot_calls, tot_fr = 0.0, 0, 0
    for sl in slots:
        ms, calls, nfr = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_int64(0)
        lib.check(lib.lm_ctx_profile_read(sl["fs"].labeler.ctx, ctypes.addressof(ms), ctypes.addressof(calls), ctypes.addressof(nfr)))
        lib.check(lib.lm_ctx_set_profiling(sl["fs"].labeler.ctx, 0))
        tot_ms += ms.value
        tot_calls += calls.value
        tot_fr += nfr.value
    ms, calls, nfr = ctypes.c_double(tot_ms), ctypes.c_int64(tot_calls), ctypes.c_int64(tot_fr)
    k1 = fs.counters()
    assert k0 is None or k1 == k0, "steps are not reproducible: %r vs %r" % (k0, k1)
    # every slot's last step (they ran overlapped) left bit for bit what the step alone left
    identical = all(stream_digest(sl["fs"]) == digest_alone for sl in slots[:min(depth, a.steps)])
    if not identical:
        sys.stderr.write("bench.py: a pipelined step differs from the same step run alone -- the rate below is NOT a valid result\n")
        if os.environ.get("LM_BENCH_STRICT"):
            raise AssertionError("a pipelined step differs from the same step run alone")

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # the same launch sequence once more with nothing else in flight (after the timed region): in the pipeline, kernels of the
    # other streams' steps share the GPU with it, which stretches the live figure without saying anything about the kernels
    alone_ms = None
    if depth > 1:
        sl = slots[0]
        lib.check(lib.lm_ctx_set_profiling(sl["fs"].labeler.ctx, 1))
        front(sl, split=False)
        torch.cuda.synchronize()
        ms2, calls2, nfr2 = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_int64(0)
        lib.check(lib.lm_ctx_profile_read(sl["fs"].labeler.ctx, ctypes.addressof(ms2), ctypes.addressof(calls2), ctypes.addressof(nfr2)))
        lib.check(lib.lm_ctx_set_profiling(sl["fs"].labeler.ctx, 0))
        if calls2.value > 0 and nfr2.value == nfr.value * calls2.value // max(calls.value, 1):
            alone_ms = ms2.value / calls2.value

    launch_ms = ms.value / max(calls.value, 1)
    frames_per_launch = nfr.value / max(calls.value, 1)
    algo_bytes = ALGO_BYTES_PER_PX * W * H * frames_per_launch
    achieved = algo_bytes / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "r01_label_traffic_pmc.json")
    if os.path.exists(tpath) and (W, H) == (1920, 1080):
        tj = json.load(open(tpath))     # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same launch sequence (not collectable live)
        traffic = int(tj["traffic_bytes_per_frame"] * frames_per_launch)
        traffic_src = "profiles/r01_label_traffic_pmc.json (separate rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE passes, gfx950 FETCH x2 correction on the image read)"
    roofline = {"bound": "hbm", "kernel": "lm_label_batch[lm_k_band+lm_k_seam_union+lm_k_flatten_flag+lm_k_apply_labels+lm_k_write_labels]",
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic, "traffic_source": traffic_src, "launch_ms": round(launch_ms, 4), "frames_per_launch": frames_per_launch,
                "algorithmic_bytes_per_launch": int(algo_bytes), "label_image_written": labels is not None}
    if alone_ms:
        roofline["alone"] = {"launch_ms": round(alone_ms, 4), "achieved": round(algo_bytes / (alone_ms * 1e-3) / 1e9, 2),
                             "frac": round(algo_bytes / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "note": "same launches after the timed region, no other stream's step in flight"}

    # ---- CPU baseline: the oracle (C port, 1 thread) on a prefix of the same stream
    cpu = None
    if a.cpu_frames > 0 and world == 1:     # reported at N = 1 only
        from oracle import cc as occ
        n = min(a.cpu_frames, F)
        lg = logits[:n].cpu().numpy()
        t0 = time.perf_counter()
        st = occ.Stability(W, H, 0.85, 0.85, 85)
        for i in range(n):
            st.add_frame(occ.threshold_invert(lg[i]))
        cdt = time.perf_counter() - t0
        cpu = {"value": round(n / cdt, 3), "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": "first %d frames of the same stream: threshold+invert, label, stats, crops, temporal matching "
                         "(oracle/cc_oracle.c, single thread; host has %d cores)" % (n, os.cpu_count())}

    total_frames = F * a.steps * world
    out = {
        "metric": "frames/sec end-to-end binarize+CC+group @1080p", "value": round(total_frames / dt, 2), "unit": "frames/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "configs[2]: synthetic %dx%d binary-board stream, %d frames/stream/GPU: fp32 logits -> "
                               "threshold+invert -> CC label (int32 image) -> CC stats/records/crops -> temporal matching -> grouping (step 03) + "
                               "reconstructed frames"
                               % (W, H, F),
                   "frames_per_step": F, "batch": a.batch, "stages_not_in_timed_region": ["fcn conv stack (logits are synthetic, SURVEY 8(d) config 3)"],
                   "stream": dict({k: k1[k] for k in ("n_cc", "n_unique", "tempo_count")}, n_groups=int(ginfo[2]), n_split=int(ginfo[0])), "parallelism": "independent streams per GPU", "pipeline_depth": depth, "matching_on_own_hip_stream": bool(split),
                   "pipelined_steps_bit_identical_to_step_alone": bool(identical)},
        "roofline": roofline, "cpu_baseline": cpu, "gen_seconds": round(gen_s, 2),
    }
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
