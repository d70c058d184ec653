"""Parity checks shared by the GPU tests (real HIP library) and the CPU logic tests (emulated build).
Every check drives the product through its C ABI (lecturemath_amd.device -> ctypes) and compares with
the committed golden fixtures (reference outputs) and/or the oracle."""
import json
import os

import numpy as np

from lecturemath_amd import _lib, device, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STREAMS = ["accumulate_erase", "occluder_return", "short_gap_jitter"]


def unrag(arr, off):
    return [[tuple(int(v) for v in r) for r in arr[off[i]:off[i + 1]]] for i in range(len(off) - 1)]


def load_stream(name):
    g = np.load(os.path.join(GOLD, "g3_stream_%s.npz" % name))
    spec = json.loads(bytes(g["spec"]).decode())
    frames = (np.unpackbits(g["frames_packed"], axis=2)[:, :, :spec["w"]] * 255).astype(np.uint8)
    return g, spec, frames


def check_g1_frame(lib, i, g=None):
    """labels / counts / CC_AgeBoundaries arrays of golden frame i (and of its vertical flip vs the oracle)."""
    from oracle import cc as occ
    g = g if g is not None else np.load(os.path.join(GOLD, "g1_label.npz"))
    img = g["img%d" % i]
    h, w = img.shape
    lab = device.FrameLabeler(w, h, 2, lib)
    try:
        batch = np.stack([img, img[::-1].copy()])
        labels, counts = lab.label(lab.be.from_host(batch))
        labels = lab.be.to_host(labels)
        assert counts[0] == int(g["n%d" % i])
        assert labels.dtype == np.int32 and (labels[0] == g["labels%d" % i]).all()
        l2, n2 = occ.label4(batch[1])
        assert counts[1] == n2 and (labels[1] == l2).all()
        st = lab.stats(counts)
        if counts[0]:
            assert (st[0] == g["stats%d" % i]).all()
        if n2:
            assert (st[1] == np.stack(occ.age_boundaries(l2, None, n2)[:5])).all()
    finally:
        lab.close()


def state_equal_golden(r, g):
    assert r["tempo_count"] == int(g["tempo_count"])
    assert list(r["active"]) == list(g["active"])
    assert len(r["unique_recs"]) == len(g["unique_recs"]) and (r["unique_recs"] == g["unique_recs"]).all()
    assert r["unique_cc_frames"] == unrag(g["ucf"], g["ucf_off"])
    assert r["cc_idx_per_frame"] == unrag(g["cipf"], g["cipf_off"])
    got = np.concatenate([c.ravel() for c in r["unique_crops"]]) if r["unique_crops"] else np.zeros(0, np.uint8)
    assert (got == g["unique_crops"]).all()


def state_equal_oracle(r, o):
    assert r["tempo_count"] == o["tempo_count"]
    assert list(r["active"]) == list(o["active"])
    assert len(r["unique_recs"]) == len(o["unique_recs"]) and (r["unique_recs"] == o["unique_recs"]).all()
    assert r["unique_cc_frames"] == o["unique_cc_frames"]
    assert r["cc_idx_per_frame"] == o["cc_idx_per_frame"]
    for a, b in zip(r["unique_crops"], o["unique_crops"]):
        assert a.shape == b.shape and (a == b).all()


def run_stream(lib, frames, w, h, max_gap, max_batch=7, split=None, records_then_match=False, thresholds=(0.85, 0.85), **kw):
    fs = device.FrameStream(w, h, len(frames), thresholds[0], thresholds[1], max_gap, 20, max_batch=max_batch, lib=lib, **kw)
    try:
        dev = fs.be.from_host(frames)
        if records_then_match:      # all records first, then ONE matching call (chunks of 64 frames inside the library)
            fs.push_records(dev)
            fs.match(len(frames))
        elif split:
            fs.push(dev[:split])
            fs.push(dev[split:])
        else:
            fs.push(dev)
        return fs.result()
    finally:
        fs.close()


def check_stream_run_logits(lib, n_frames=23, h=96, w=160, batch=5, schedule=0, second_stream=False):
    """lm_stream_run_logits (the whole steps 01-02 loop in one call, logits in, matched stream out) == threshold_invert + push
    batch by batch, with and without a separate matching stream argument; and the oracle."""
    frames = np.stack(list(synth.binary_stream(n_frames, h, w, seed=8, glyphs_per_add=4, erase_every=8, jitter_p=0.4, occluder=True, max_ext=16)))
    logits = synth.logits_from_binary(frames, seed=2)
    ref = run_stream(lib, frames, w, h, 5, max_batch=batch)
    fs = device.FrameStream(w, h, n_frames, 0.85, 0.85, 5, 20, max_batch=batch, lib=lib)
    try:
        be = fs.be
        d_logits = be.from_host(logits)
        scratch = be.empty((batch, h, w), np.uint8)
        labels = be.empty((batch, h, w), np.int32)
        st = be.stream()
        st2 = st
        if second_stream and be.device:
            side = be.torch.cuda.Stream()
            st2 = side.cuda_stream
        # the stream object is first filled from a DIFFERENT input and reset: records left behind by an earlier stream must not be read
        # by the matching kernels of this one (they overlap the next batch's emission on the other queue)
        other = synth.logits_from_binary(np.stack(list(synth.binary_stream(n_frames, h, w, seed=99, glyphs_per_add=6, erase_every=5, max_ext=16))), seed=5)
        d_other = be.from_host(other)
        lib.check(lib.lm_stream_run_logits(fs.handle, _lib.ptr(d_other), n_frames, batch, None, _lib.ptr(labels), 128, 1, schedule, st, st2))
        if st2 is not st:
            side.synchronize()
        fs.result()
        fs.reset()
        lib.check(lib.lm_stream_run_logits(fs.handle, _lib.ptr(d_logits), n_frames, batch, _lib.ptr(scratch), _lib.ptr(labels), 128, 1, schedule, st, st2))
        if st2 is not st:
            side.synchronize()
        got = fs.result()
    finally:
        fs.close()
    for key in ("unique_cc_frames", "cc_idx_per_frame", "tempo_count"):
        assert got[key] == ref[key], key
    assert list(got["active"]) == list(ref["active"]) and (got["unique_recs"] == ref["unique_recs"]).all()
    from oracle import cc as occ
    o = occ.Stability(w, h, 0.85, 0.85, 5)
    for f in frames:
        o.add_frame(f)
    state_equal_oracle(got, o.result())


def check_stream_golden(lib, name, max_batch=7):
    g, spec, frames = load_stream(name)
    r = run_stream(lib, frames, spec["w"], spec["h"], spec["gap2"], max_batch=max_batch, split=len(frames) // 2)
    state_equal_golden(r, g)


def check_stream_oracle(lib, frames, max_gap, max_batch=5, thresholds=(0.85, 0.85), **kw):
    from oracle import cc as occ
    h, w = frames[0].shape
    st = occ.Stability(w, h, thresholds[0], thresholds[1], max_gap)
    for f in frames:
        st.add_frame(f)
    r = run_stream(lib, np.stack(frames), w, h, max_gap, max_batch=max_batch, thresholds=thresholds, **kw)
    state_equal_oracle(r, st.result())
    return r


def blob_stream(n_frames=14, h=160, w=800, seed=17):
    """What an hour of lecture looks like to step 02: one large component (a mesh of strokes over most of the frame, its crop
    several thousand words) that grows a little every other frame -- so consecutive versions are twins half of the time and
    near-identical large non-twins otherwise -- among glyph-sized CCs lying inside its box, some of them blinking.  Exercises
    the word-parallel paths: lm_k_emit's crop words, lm_k_mb_twin_cmp, the size prune and lm_k_mb_eval_big."""
    rng = np.random.default_rng(seed)
    mesh = np.zeros((h, w), np.uint8)
    mesh[10:h - 10:12, 8:w - 8] = 255              # long horizontal strokes ...
    mesh[10:h - 10, 8:w - 8:40] = 255              # ... tied together by vertical ones: ONE component
    dots = [(int(rng.integers(14, h - 18)), int(rng.integers(12, w - 20))) for _ in range(60)]
    frames = []
    for t in range(n_frames):
        img = mesh.copy()
        grow = 3 * (t // 2)                         # a spur that gets longer every other frame
        img[h // 2 + 3:h // 2 + 3 + 2, 20:20 + 4 + grow] = 255
        img[h // 2 + 1:h // 2 + 5, 20:22] = 255     # attached to a horizontal stroke
        for i, (y, x) in enumerate(dots):
            if (i + t) % 7 == 0:
                continue
            y0 = y - (y - 10) % 12 + 3              # between two strokes: a separate glyph-sized CC
            img[y0:y0 + 6, x - (x - 8) % 40 + 4:x - (x - 8) % 40 + 10] = 255
        frames.append(img)
    return frames


def check_stream_large_components(lib, **kw):
    frames = blob_stream(**kw)
    r = check_stream_oracle(lib, frames, max_gap=4, max_batch=5, max_ccs=1 << 14, max_crop_words=1 << 20)
    big = max(int(rec[4]) for rec in r["unique_recs"])
    assert big > 5000, big                          # the mesh: thousands of pixels, crop of thousands of words
    check_stream_oracle(lib, frames, max_gap=4, max_batch=5, records_then_match=True, max_ccs=1 << 14, max_crop_words=1 << 20)
    return r


def check_stream_threshold_edges(lib):
    """Thresholds the twin shortcut must respect: exactly 1.0 (only identical crops match: everything rides on twins) and
    above 1.0 (nothing ever matches, twin detection is off: every CC of every frame is a new unique)."""
    frames = list(synth.binary_stream(40, 64, 96, seed=5, glyphs_per_add=3, erase_every=6, jitter_p=0.3, occluder=True, max_ext=12))
    r = check_stream_oracle(lib, frames, 3, max_batch=16, thresholds=(1.0, 1.0))
    assert len(r["unique_recs"]) < sum(len(fr) for fr in r["cc_idx_per_frame"])
    r = check_stream_oracle(lib, frames, 3, max_batch=16, thresholds=(1.01, 0.5))
    assert len(r["unique_recs"]) == sum(len(fr) for fr in r["cc_idx_per_frame"])


def check_threshold_paths(lib):
    """lm_threshold_invert's comparison form (x >= x*, x* found on the device with the formula) vs the formula kernel
    (LM_THRESHOLD_FORMULA=1) for several thresholds: logits packed around each edge ulp by ulp, +-inf, NaN, a ragged tail."""
    rng = np.random.default_rng(9)
    lab = device.FrameLabeler(64, 64, 1, lib)
    try:
        for thr in (1, 2, 77, 127, 128, 129, 254, 255):
            centre = np.float32(np.log((thr / 255.0) / max(1.0 - thr / 255.0, 1e-9)))
            around = (np.full(20001, centre, np.float32).view(np.int32) + np.arange(-10000, 10001, dtype=np.int32)).view(np.float32)
            x = np.concatenate([around, rng.normal(0, 6, 30000).astype(np.float32),
                                np.array([np.inf, -np.inf, np.nan, 0.0, -0.0, 88.0, -88.0, 104.0, -104.0], np.float32)])
            x = x[:len(x) - (len(x) % 16) + 5]          # 5 pixels beyond the last 16-pixel group
            dev = lab.be.from_host(x)
            fast = lab.be.to_host(lab.threshold_invert(dev, thr))
            os.environ["LM_THRESHOLD_FORMULA"] = "1"
            try:
                formula = lab.be.to_host(lab.threshold_invert(dev, thr))
            finally:
                del os.environ["LM_THRESHOLD_FORMULA"]
            assert (fast == formula).all(), thr
            assert 0 < int((fast == 255).sum()) < len(x)
    finally:
        lab.close()


def check_stream_match_paths(lib, n_frames=80, max_gap=2, seed=11):
    """The batched matcher (default), the per-frame kernels (LM_MATCH_PER_FRAME=1) and the oracle agree on a stream whose
    uniques are created, retired (small max_gap) and re-created inside one matching batch, for several batch shapes."""
    frames = list(synth.binary_stream(n_frames, 64, 96, seed=seed, glyphs_per_add=3, erase_every=5, jitter_p=0.5, occluder=True,
                                      max_ext=12))
    r1 = check_stream_oracle(lib, frames, max_gap, max_batch=32)
    r2 = check_stream_oracle(lib, frames, max_gap, max_batch=1)
    r3 = check_stream_oracle(lib, frames, max_gap, max_batch=16, records_then_match=True)
    old = os.environ.get("LM_MATCH_PER_FRAME")
    os.environ["LM_MATCH_PER_FRAME"] = "1"
    try:
        r4 = check_stream_oracle(lib, frames, max_gap, max_batch=32)
    finally:
        if old is None:
            del os.environ["LM_MATCH_PER_FRAME"]
        else:
            os.environ["LM_MATCH_PER_FRAME"] = old
    for r in (r2, r3, r4):
        state_equal_oracle(r, r1)
    assert len(r1["unique_recs"]) > 30 and len(r1["active"]) < len(r1["unique_recs"])


def check_label_logits_fused(lib, shapes=((3, 37, 68), (2, 9, 1028), (1, 5, 4100))):
    """lm_label_batch_logits (threshold fused into the row packing: ballots of 1024-pixel blocks) == lm_threshold + lm_label_batch:
    labels, counts, statistics and the {0, 255} frames, for row widths that end inside a block / a 64-pixel word / beyond one 4096-pixel
    trip, both polarities, logits packed ulp by ulp around the threshold edge, +-inf and NaN; and the unfused fallback (width % 4 != 0)."""
    rng = np.random.default_rng(31)
    for (b, h, w) in tuple(shapes) + ((2, 21, 70),):
        lab = device.FrameLabeler(w, h, b, lib)
        try:
            for thr, invert in ((128, True), (128, False), (77, True)):
                centre = np.float32(np.log((thr / 255.0) / (1.0 - thr / 255.0)))
                x = rng.normal(0, 2.5, (b, h, w)).astype(np.float32)
                edge = (np.full(b * h * w, centre, np.float32).view(np.int32) + rng.integers(-40, 41, b * h * w).astype(np.int32)).view(np.float32)
                pick = rng.random((b, h, w)) < 0.3
                x[pick] = edge.reshape(b, h, w)[pick]
                x.reshape(-1)[:6] = np.array([np.inf, -np.inf, np.nan, 0.0, -0.0, 104.0], np.float32)
                dev = lab.be.from_host(x)
                labels, counts, binary = lab.label_logits(dev, thr, invert, want_binary=True)
                assert lib.lm_label_was_fused(lab.ctx) == (1 if w % 4 == 0 else 0)
                got_l, got_b, got_st = lab.be.to_host(labels), lab.be.to_host(binary), lab.stats(counts)
                ref_b = lab.be.empty((b, h, w), np.uint8)
                lib.check(lib.lm_threshold(_lib.ptr(dev), _lib.ptr(ref_b), b * h * w, thr, 1 if invert else 0, lab.be.stream()))
                ref_l, ref_counts = lab.label(ref_b)
                ref_st = lab.stats(ref_counts)
                assert (got_b == lab.be.to_host(ref_b)).all() and (got_l == lab.be.to_host(ref_l)).all() and (counts == ref_counts).all(), (b, h, w, thr, invert)
                assert all((a == r).all() for a, r in zip(got_st, ref_st))
                assert 0 < int((got_b == 255).sum()) < got_b.size
                # without the byte frames
                labels2, counts2, _ = lab.label_logits(dev, thr, invert, want_binary=False) if w % 4 == 0 else (labels, counts, None)
                assert (lab.be.to_host(labels2) == got_l).all() and (counts2 == counts).all()
        finally:
            lab.close()


def check_label_vs_oracle(lib, img):
    """labels, counts and CC_AgeBoundaries arrays of an arbitrary uint8 frame vs the oracle."""
    from oracle import cc as occ
    h, w = img.shape
    lab = device.FrameLabeler(w, h, 1, lib)
    try:
        labels, counts = lab.label(lab.be.from_host(img[None]))
        l, n = occ.label4(img)
        assert counts[0] == n and (lab.be.to_host(labels)[0] == l).all()
        st = lab.stats(counts)
        if n:
            assert (st[0] == np.stack(occ.age_boundaries(l, None, n)[:5])).all()
    finally:
        lab.close()


def check_label_batch_in_parts(lib, n_frames=19, h=45, w=150):
    """A batch large enough to be labelled in parts on two queues (lm_label_batch): labels, counts and statistics of every frame
    vs the oracle, twice in a row on the same context (the second call's parts follow the first call's join)."""
    from oracle import cc as occ
    rng = np.random.default_rng(77)
    frames = ((rng.random((n_frames, h, w)) < rng.uniform(0.1, 0.6, size=(n_frames, 1, 1))) * 255).astype(np.uint8)
    frames[3] = 0
    frames[n_frames - 1] = 255
    lab = device.FrameLabeler(w, h, n_frames, lib)
    try:
        for rep in range(2):
            dev = lab.be.from_host(frames if rep == 0 else frames[::-1].copy())
            labels, counts = lab.label(dev)
            got = lab.be.to_host(labels)
            st = lab.stats(counts)
            for k in range(n_frames):
                img = frames[k] if rep == 0 else frames[n_frames - 1 - k]
                l, n = occ.label4(img)
                assert counts[k] == n and (got[k] == l).all(), (rep, k)
                if n:
                    assert (st[k] == np.stack(occ.age_boundaries(l, None, n)[:5])).all(), (rep, k)
    finally:
        lab.close()


def grouping_equal_golden(r, g):
    """Every step-03 intermediate of the reference (G4 fixture) vs the product's Grouping.result()."""
    assert r["n_split"] == int(g["n_split"])
    assert r["unique_cc_frames"] == unrag(g["post_split_ucf"], g["post_split_ucf_off"])
    assert r["cc_idx_per_frame"] == unrag(g["post_split_cipf"], g["post_split_cipf_off"])
    assert r["stable_idxs"] == list(g["stable"])
    assert r["total_intersections"] == int(g["total_intersections"])
    tflat = [(a, b, np.float64(rc).view(np.int64), np.float64(p).view(np.int64))
             for a, lst in enumerate(r["time_overlapping_cc"]) for b, rc, p in lst]
    assert (np.asarray(tflat, np.int64).reshape(-1, 4) == g["time_ov"]).all()
    aflat = [(a, *t) for a, lst in enumerate(r["all_overlapping_cc"]) for t in lst]
    assert (np.asarray(aflat, np.int64).reshape(-1, 5) == g["all_ov"]).all()
    assert [[(m,) for m in grp] for grp in r["cc_groups"]] == unrag(g["groups"], g["groups_off"])
    assert sorted(r["group_idx_per_cc"].items()) == [tuple(x) for x in g["gid"]]
    ng = len(r["cc_groups"])
    assert [[(a,) for a in r["group_ages"][k]] for k in range(ng)] == unrag(g["ages"], g["ages_off"])
    assert [[(a,) for a in fr] for fr in r["groups_per_frame"]] == unrag(g["gpf"], g["gpf_off"])
    got = sorted((k, o, d["matched"], d["unmatched"], d["area_union"], d["area_intersection"])
                 for k in r["conflicts"] for o, d in r["conflicts"][k].items())
    exp = sorted(tuple(row) for row in g["conflicts"])
    assert len(got) == len(exp) and all(tuple(float(v) for v in a) == tuple(float(v) for v in b) for a, b in zip(got, exp))
    assert (np.asarray([r["group_boundaries"][k] for k in range(ng)], np.int64).reshape(-1, 4) == g["bounds"]).all()
    assert [len(r["group_images"][k]) for k in range(ng)] == list(g["gimg_count"])
    gi = np.concatenate([im.ravel() for k in range(ng) for im in r["group_images"][k]]) if ng else np.zeros(0, np.uint8)
    assert (gi == g["gimg"]).all()
    clean = np.stack(r["clean_binary"])
    assert (np.packbits(clean == 255, axis=2) == g["clean_packed"]).all()
    other = np.argwhere((clean != 0) & (clean != 255)).astype(np.int32)
    assert (other == g["clean_other"]).all() and (clean[(clean != 0) & (clean != 255)] == g["clean_other_val"]).all()


def check_grouping_golden(lib, name, max_batch=16):
    g, spec, frames = load_stream(name)
    fs = device.FrameStream(spec["w"], spec["h"], len(frames), 0.85, 0.85, spec["gap2"], 20, max_batch=max_batch, lib=lib)
    try:
        fs.push(fs.be.from_host(frames))
        gr = device.Grouping(fs, max_gap=spec["gap3"], min_times=3, t_window=5, min_recall=0.5, img_threshold=0.5)
        try:
            grouping_equal_golden(gr.result(), g)
        finally:
            gr.close()
    finally:
        fs.close()


def check_grouping_oracle(lib, frames, gap2=85, gap3=85, max_batch=8):
    """Step 03 of an arbitrary stream vs the oracle: groups, ages, group images and every reconstructed frame."""
    from oracle import cc as occ
    from oracle import grouping as og
    h, w = frames[0].shape
    st = occ.Stability(w, h, 0.85, 0.85, gap2)
    for f in frames:
        st.add_frame(f)
    o = og.run_step03(st.result(), max_gap=gap3)
    fs = device.FrameStream(w, h, len(frames), 0.85, 0.85, gap2, 20, max_batch=max_batch, lib=lib, max_ccs=len(frames) * h * w // 24,
                            max_crop_words=len(frames) * h * w // 4)
    try:
        fs.push(fs.be.from_host(np.stack(frames)))
        gr = device.Grouping(fs, max_gap=gap3, min_times=3, t_window=5, min_recall=0.5, img_threshold=0.5)
        try:
            r = gr.result()
        finally:
            gr.close()
    finally:
        fs.close()
    assert r["cc_groups"] == o["cc_groups"] and len(r["cc_groups"]) > 0
    ng = len(o["cc_groups"])
    assert [list(r["group_ages"][k]) for k in range(ng)] == [list(o["group_ages"][k]) for k in range(ng)]
    assert [list(fr) for fr in r["groups_per_frame"]] == [list(fr) for fr in o["groups_per_frame"]]
    for k in range(ng):
        assert len(r["group_images"][k]) == len(o["group_images"][k])
        for a, b in zip(r["group_images"][k], o["group_images"][k]):
            assert a.shape == b.shape and (a == b).all()
    assert (np.stack(r["clean_binary"]) == np.stack(o["clean_binary"])).all()
    return r


def churn_stream(n_frames=40, h=160, w=640, seed=21, empty_every=7):
    """Every frame re-draws its 5x4 dots at fresh random grid cells (almost everything is a new unique every frame: tens of
    thousands of in-batch sources, active positions far beyond the replay kernel's LDS tables), with completely empty
    frames in between (also as the very first frame)."""
    rng = np.random.default_rng(seed)
    ys, xs = np.arange(2, h - 5, 6), np.arange(2, w - 6, 7)
    frames = []
    for f in range(n_frames):
        img = np.zeros((h, w), np.uint8)
        if f % empty_every != 0:
            on = rng.random((len(ys), len(xs))) < 0.45
            for i, y in enumerate(ys):
                for j, x in enumerate(xs):
                    if on[i, j]:
                        dy, dx = rng.integers(0, 2, 2)
                        img[y + dy:y + dy + 4, x + dx:x + dx + 5] = 255
        frames.append(img)
    return frames


def dot_grid_stream(n_frames=6, h=72, w=520, seed=3):
    """Hundreds of 5x4-pixel dots per 64x256 tile (more stable groups in one render tile than its cooperative hit list
    holds), a few of them blinking."""
    rng = np.random.default_rng(seed)
    ys, xs = np.arange(2, h - 5, 6), np.arange(2, w - 6, 7)
    frames = []
    for f in range(n_frames):
        img = np.zeros((h, w), np.uint8)
        on = rng.random((len(ys), len(xs))) > (0.03 if f else 0.0)
        for i, y in enumerate(ys):
            for j, x in enumerate(xs):
                if on[i, j]:
                    img[y:y + 4, x:x + 5] = 255
        frames.append(img)
    return frames


def blink_overlap_stream(n_frames=30, h=80, w=300):
    """Shapes that alternate between two / three overlapping places: their groups are alive together, so their images are
    added on top of each other in the reconstructed frames (uint8 wrap-around: k images over a pixel leave -k mod 256), at
    columns that are not multiples of 32 and across render-tile borders (x = 256)."""
    frames = []
    for f in range(n_frames):
        img = np.zeros((h, w), np.uint8)
        img[5:9, 5:10] = 255
        for y, x in [(20, 27), (40, 61), (50, 250), (60, 120)]:
            if (f // 5) % 2 == 0:
                img[y:y + 10, x:x + 10] = 255
            else:
                img[y + 3:y + 13, x + 8:x + 18] = 255
        y3, x3 = [(30, 180), (38, 180), (34, 188)][(f // 4) % 3]
        img[y3:y3 + 12, x3:x3 + 12] = 255
        frames.append(img)
    return frames


def check_render_wraparound(lib):
    r = check_grouping_oracle(lib, blink_overlap_stream())
    values = set(np.unique(np.stack(r["clean_binary"])).tolist())
    assert {0, 253, 254, 255} <= values, values


def reference_c_library():
    """oracle/_ref/accessmath_lib.so: the reference's own C file compiled by oracle/Makefile (travels to the GPU box)."""
    import ctypes
    path = os.path.join(ROOT, "oracle", "_ref", "accessmath_lib.so")
    if not os.path.exists(path):
        import pytest
        pytest.skip("oracle/_ref/accessmath_lib.so not built (reference sources absent)")
    return ctypes.CDLL(path)


def check_legacy_exports(lib, big=False):
    """The classical-binarizer exports of accessmath_lib.c (speaker_detection_handle_frame, regionCumulativeDistribution,
    adapthisteq, combine_results) vs the reference C library itself, bit for bit (float64 outputs compared as bit patterns)."""
    import ctypes
    ref, mine = reference_c_library(), lib.cdll
    rng = np.random.default_rng(17)
    vp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    for L in (ref, mine):
        L.regionCumulativeDistribution.restype = None
        L.regionCumulativeDistribution.argtypes = [vp, ci, ci, ci, ci, ci, ci, cd, vp]
        L.adapthisteq.argtypes = [vp, ci, ci, cd, ci, ci, vp]
        L.combine_results.argtypes = [vp, vp, ci, ci, ctypes.c_ubyte, vp]
        L.speaker_detection_handle_frame.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp, vp, vp]
        L.speaker_detection_handle_frame.restype = ci
    sizes = [(97, 131), (64, 64)] + ([(1080, 1920)] if big else [])
    for h, w in sizes:
        # smooth image + noise so that histograms are neither flat nor degenerate
        yy, xx = np.mgrid[0:h, 0:w]
        gray = np.clip(120 + 60 * np.sin(xx / 17.0) * np.cos(yy / 23.0) + rng.normal(0, 25, (h, w)), 0, 255).astype(np.uint8)
        gray = np.ascontiguousarray(gray)
        # -- regionCumulativeDistribution
        for (x0, x1, y0, y1), slope in (((0, w - 1, 0, h - 1), 0.0), ((3, w // 2, 5, h - 2), 0.01), ((w // 3, w // 3, 0, 0), 0.02)):
            outs = []
            for L in (ref, mine):
                o = np.zeros(256, np.float64)
                L.regionCumulativeDistribution(gray.ctypes.data, w, h, x0, x1, y0, y1, slope, o.ctypes.data)
                outs.append(o)
            assert (outs[0].view(np.int64) == outs[1].view(np.int64)).all()
        # -- adapthisteq
        for gx, gy, slope in ((8, 8, 0.01), (3, 5, 0.0), (1, 1, 0.02), (2, 1, 0.005)):
            outs = []
            for L in (ref, mine):
                o = np.zeros((h, w), np.uint8)
                L.adapthisteq(gray.ctypes.data, w, h, slope, gx, gy, o.ctypes.data)
                outs.append(o)
            assert (outs[0] == outs[1]).all(), (h, w, gx, gy, slope, int((outs[0] != outs[1]).sum()))
        # -- combine_results
        board = rng.integers(0, 256, (h, w), dtype=np.uint8)
        for thr in (0, 97, 255):
            outs = []
            for L in (ref, mine):
                o = np.zeros((h, w), np.uint8)
                L.combine_results(board.ctypes.data, gray.ctypes.data, w, h, thr, o.ctypes.data)
                outs.append(o)
            assert (outs[0] == outs[1]).all()
        # -- speaker_detection_handle_frame
        f0 = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        f1 = f0.copy()
        f1[h // 4:h // 2, w // 3:w // 3 + 20] = rng.integers(0, 256, (h // 2 - h // 4, 20, 3), dtype=np.uint8)
        for a, b, thr, jump in ((f1, f0, 20, 1), (f1, f0, 20, 4), (f0, f0, 20, 2), (f1, f0, 300, 1)):
            outs = []
            for L in (ref, mine):
                bd, av, dv = np.zeros(4), np.zeros(2), np.zeros(2)
                n = L.speaker_detection_handle_frame(a.ctypes.data, b.ctypes.data, w, h, 3, thr, jump, bd.ctypes.data, av.ctypes.data, dv.ctypes.data)
                outs.append((n, bd.view(np.int64).tolist(), av.view(np.int64).tolist(), dv.view(np.int64).tolist()))
            assert outs[0] == outs[1], (h, w, thr, jump, outs)


def check_fcn_golden(lib, name, tol=1e-3, precision="f16x3", require_planar=False, formats=None):
    """HIP FCN forward vs the reference module's outputs (G5 fixture); tolerance 1e-3 on logits (BASELINE.json)."""
    from lecturemath_amd import fcn
    g = np.load(os.path.join(GOLD, "g5_fcn_%s.npz" % name))
    sd = {k[3:]: g[k] for k in g.files if k.startswith("sd.")}
    rgb = g["rgb"]
    h, w = rgb.shape[:2]
    eng = fcn.FcnEngine(g["widths"], int(g["pk"]), 3, h, w, lib, precision=precision, formats=formats)
    try:
        assert eng.planar or not require_planar, "the planar engine refused this network"
        eng.load_state_dict(sd)
        out, text, rec = (eng.be.to_host(t) for t in eng.forward(rgb))
        assert np.abs(out - g["out"][0, 0]).max() <= tol
        assert np.abs(text - g["text"][0, 0]).max() <= tol
        assert np.abs(rec - g["rec"][0]).max() <= tol
        return float(np.abs(out - g["out"][0, 0]).max())
    finally:
        eng.close()
