"""CPU: the oracle reproduces every golden vector the reference produced (tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest

import lm_checks
from lm_checks import GOLD


def test_g1_label_stats_crops(oracle_built):
    occ = oracle_built
    g = np.load(os.path.join(GOLD, "g1_label.npz"))
    for i in range(int(g["n"])):
        img = g["img%d" % i]
        labels, n = occ.label4(img)
        assert n == int(g["n%d" % i]) and (labels == g["labels%d" % i]).all()
        if n:
            st = occ.age_boundaries(labels, np.zeros(img.shape, np.float32), n)
            assert (np.stack(st[:5]) == g["stats%d" % i]).all()
            assert (st[5] == g["ages%d" % i]).all()
        rec, crops = occ.extract(labels, n)
        assert (rec == g["rec%d" % i]).all()
        flat = np.concatenate([c.ravel() for c in crops]) if crops else np.zeros(0, np.uint8)
        assert (flat == g["crops%d" % i]).all()


def test_g2_overlap(oracle_built):
    occ = oracle_built
    g = np.load(os.path.join(GOLD, "g2_overlap.npz"))
    oa = ob = 0
    for k, row in enumerate(g["boxes"]):
        a, b = row[:5], row[5:]
        na = (a[1] - a[0] + 1) * (a[3] - a[2] + 1)
        nb = (b[1] - b[0] + 1) * (b[3] - b[2] + 1)
        ca = g["crops_a"][oa:oa + na].reshape(a[3] - a[2] + 1, -1)
        cb = g["crops_b"][ob:ob + nb].reshape(b[3] - b[2] + 1, -1)
        oa += na
        ob += nb
        m = occ.overlap(a[:4], ca, b[:4], cb)
        assert m / float(np.int32(a[4])) == g["recall"][k]
        assert m / float(np.int32(b[4])) == g["precision"][k]


@pytest.mark.parametrize("name", lm_checks.STREAMS)
def test_g3_g4_stream(oracle_built, name):
    from oracle import grouping as og
    occ = oracle_built
    g, spec, frames = lm_checks.load_stream(name)
    st = occ.Stability(spec["w"], spec["h"], 0.85, 0.85, spec["gap2"])
    for f in frames:
        st.add_frame(f)
    state = st.result()
    lm_checks.state_equal_golden(state, g)
    r = og.run_step03(state, max_gap=spec["gap3"])
    assert r["n_split"] == int(g["n_split"])
    assert state["unique_cc_frames"] == lm_checks.unrag(g["post_split_ucf"], g["post_split_ucf_off"])
    assert state["cc_idx_per_frame"] == lm_checks.unrag(g["post_split_cipf"], g["post_split_cipf_off"])
    assert r["stable_idxs"] == list(g["stable"])
    assert r["total_intersections"] == int(g["total_intersections"])
    tflat = [(a, b, np.float64(rc).view(np.int64), np.float64(p).view(np.int64))
             for a, lst in enumerate(r["time_overlapping_cc"]) for b, rc, p in lst]
    assert (np.asarray(tflat, np.int64).reshape(-1, 4) == g["time_ov"]).all()
    aflat = [(a, *t) for a, lst in enumerate(r["all_overlapping_cc"]) for t in lst]
    assert (np.asarray(aflat, np.int64).reshape(-1, 5) == g["all_ov"]).all()
    assert [[(m,) for m in grp] for grp in r["cc_groups"]] == lm_checks.unrag(g["groups"], g["groups_off"])
    assert sorted(r["group_idx_per_cc"].items()) == [tuple(x) for x in g["gid"]]
    ng = len(r["cc_groups"])
    assert [[(a,) for a in r["group_ages"][k]] for k in range(ng)] == lm_checks.unrag(g["ages"], g["ages_off"])
    assert [[(a,) for a in fr] for fr in r["groups_per_frame"]] == lm_checks.unrag(g["gpf"], g["gpf_off"])
    cflat = [(k, o, d["matched"], d["unmatched"], d["area_union"], d["area_intersection"])
             for k in sorted(r["conflicts"]) for o, d in r["conflicts"][k].items()]
    assert (np.asarray(cflat, np.float64).reshape(-1, 6) == g["conflicts"]).all()
    assert (np.asarray([r["group_boundaries"][k] for k in range(ng)], np.int64).reshape(-1, 4) == g["bounds"]).all()
    assert [len(r["group_images"][k]) for k in range(ng)] == list(g["gimg_count"])
    gi = np.concatenate([im.ravel() for k in range(ng) for im in r["group_images"][k]]) if ng else np.zeros(0, np.uint8)
    assert (gi == g["gimg"]).all()
    clean = np.stack(r["clean_binary"])
    assert (np.packbits(clean == 255, axis=2) == g["clean_packed"]).all()
    other = np.argwhere((clean != 0) & (clean != 255)).astype(np.int32)
    assert (other == g["clean_other"]).all() and (clean[(clean != 0) & (clean != 255)] == g["clean_other_val"]).all()


def test_g6_threshold(oracle_built):
    g = np.load(os.path.join(GOLD, "g6_threshold.npz"))
    out = oracle_built.threshold_invert(g["logits"])
    diff = int((out != g["expected"]).sum())
    # libm expf vs torch's vectorised sigmoid: only pixels within 1 ulp of the 128/255 edge may differ
    assert diff <= 2, diff
    far = np.abs(g["logits"] - 0.0078433) > 1e-4
    assert (out[far] == g["expected"][far]).all()


@pytest.mark.parametrize("name", lm_checks.STREAMS)
def test_g7_step04(name):
    """oracle/segmentation.py (step 04, deletion events) vs the reference's intervals and binary sums (G7)."""
    import json
    from oracle import segmentation as seg
    g = np.load(os.path.join(GOLD, "g7_step04_%s.npz" % name))
    ages = {k: [int(v) for v in g["ages"][g["ages_off"][k]:g["ages_off"][k + 1]]] for k in range(len(g["ages_off"]) - 1)}
    bounds = {k: tuple(int(v) for v in g["bounds"][k]) for k in range(len(g["bounds"]))}
    for i, ps in enumerate(json.loads(bytes(g["params"]).decode())):
        iv = seg.run_step04(int(g["n_frames"]), int(g["w"]), int(g["h"]), ages, bounds, ps["VIDEO_SEGMENTATION_DEL_EVENT_ADD_THRESHOLD"],
                            ps["VIDEO_SEGMENTATION_DEL_EVENT_MIN_LENGTH"], ps["VIDEO_SEGMENTATION_DEL_EVENT_THRESHOLD"])
        assert [tuple(int(v) for v in x) for x in iv] == [tuple(int(v) for v in x) for x in g["intervals_%d" % i]]
    # binary sums: the G4 fixture holds the reference's reconstructed frames
    g4, spec, _ = lm_checks.load_stream(name)
    clean = np.unpackbits(g4["clean_packed"], axis=2)[:, :, :spec["w"]].astype(np.uint8) * 255
    for (f, y, x), v in zip(g4["clean_other"], g4["clean_other_val"]):
        clean[f, y, x] = v
    assert [float(v) for v in seg.binary_sums(list(clean))] == [float(v) for v in g["sums"]]


def g4_space_time(name):
    """group ages / boundaries / images of the reference's step 03 (G4 fixture) as the dicts step 04 / 05 consume."""
    g, spec, _ = lm_checks.load_stream(name)
    ng = len(g["gimg_count"])
    ages = {k: [int(v[0]) for v in lm_checks.unrag(g["ages"], g["ages_off"])[k]] for k in range(ng)}
    bounds = {k: tuple(int(v) for v in g["bounds"][k]) for k in range(ng)}
    images, off = {}, 0
    for k in range(ng):
        w, h = bounds[k][1] - bounds[k][0] + 1, bounds[k][3] - bounds[k][2] + 1
        images[k] = []
        for _ in range(int(g["gimg_count"][k])):
            images[k].append(g["gimg"][off:off + w * h].reshape(h, w).copy())
            off += w * h
    return spec, ages, bounds, images


@pytest.mark.parametrize("name", lm_checks.STREAMS)
def test_g8_step05(name):
    """oracle/keyframes.py (step 05 core) vs the reference's keyframes and CC times (G8)."""
    from oracle import keyframes as okf
    spec, ages, bounds, images = g4_space_time(name)
    g = np.load(os.path.join(GOLD, "g8_step05_%s.npz" % name))
    n = int(g["n_frames"])
    times = [1000.0 * i for i in range(n)]
    for k, segs in enumerate(json.loads(bytes(g["segments"]).decode())):
        frames, cc_times = okf.keyframes(n, spec["w"], spec["h"], times, ages, bounds, images, [tuple(s) for s in segs])
        kf = np.stack(frames)
        assert (kf[..., 0] == kf[..., 1]).all() and (kf[..., 0] == kf[..., 2]).all()
        assert (np.packbits(kf[..., 0] == 255, axis=2) == g["keyframes_%d" % k]).all()
        flat = [(s, *t) for s, lst in enumerate(cc_times) for t in lst]
        assert (np.asarray(flat, np.float64).reshape(-1, 6) == g["times_%d" % k]).all()


def test_g6b_lanczos():
    """oracle/resize.py (numpy restatement of Pillow's LANCZOS for 8-bit images) against images resized by Pillow itself"""
    from oracle import resize
    g = np.load(os.path.join(GOLD, "g6b_lanczos.npz"))
    for i, (h, w, oh, ow) in enumerate(g["cases"]):
        assert (resize.resize_lanczos(g["in%d" % i], int(ow), int(oh)) == g["out%d" % i]).all(), i


@pytest.mark.parametrize("name", ["k7_70x94", "k3_135x240", "k7_66x130_wide"])
def test_g5_fcn(name):
    """oracle/fcn.py (the torch fp32 restatement every HIP FCN test is measured against) vs the REFERENCE module's outputs (G5,
    tests/golden/make_golden_fcn.py): forward() logits / text logits / reconstruction, the x_up1 intermediate, and binarize()'s three
    byte images.  Same torch build, same operators in the same order: the bar is exact equality; 1e-6 is allowed for a torch
    whose conv kernels reassociate differently."""
    import torch
    from oracle import fcn as ofcn
    g = np.load(os.path.join(GOLD, "g5_fcn_%s.npz" % name))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    rgb = g["rgb"]
    with torch.no_grad():
        out, text, rec, inter = ofcn.forward(sd, ofcn.prepare_image(rgb), return_intermediates=True)
    assert np.abs(out.numpy() - g["out"]).max() <= 1e-6
    assert np.abs(text.numpy() - g["text"]).max() <= 1e-6
    assert np.abs(rec.numpy() - g["rec"]).max() <= 1e-6
    assert np.abs(inter["up1"].numpy() - g["x_up1"]).max() <= 1e-6
    b, t, r = ofcn.binarize(sd, rgb)
    assert (b == g["binary"]).all() and (t == g["text_mask"]).all() and (r == g["rec_img"]).all()
