"""Checks of the drop-in layer (lecturemath_amd/dropin: the reference's module paths, classes and script entry points)."""
import importlib.util
import os
import pickle
import sys
import types

import numpy as np

import lm_checks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "lecturemath_amd", "dropin")


def use_library(lib):
    """Make the drop-in modules use `lib` (tests hand in the emulated build; the GPU tests the real one)."""
    from lecturemath_amd import _lib
    _lib._default = lib
    if DROPIN not in sys.path:
        sys.path.insert(0, DROPIN)
    from AccessMath.preprocessing.content.labeler import Labeler
    for fs in Labeler._streams.values():
        fs.close()
    Labeler._streams = {}


def load_script(name):
    spec = importlib.util.spec_from_file_location("lm_script_" + name.replace(".", "_"), os.path.join(DROPIN, name))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def fake_process(conf_overrides=None, params=None):
    from AM_CommonTools.configuration.configuration import Configuration
    data = {"CC_STABILITY_MIN_RECALL": "0.850", "CC_STABILITY_MIN_PRECISION": "0.850", "CC_STABILITY_MAX_GAP": "85",
            "CC_STABILITY_MIN_TIMES": "3", "CC_GROUPING_MIN_IMAGE_THRESHOLD": "0.5", "CC_GROUPING_TEMPORAL_WINDOW": "5",
            "CC_GROUPING_MIN_RECALL": "0.5", "CC_GROUPING_MIN_TIME_F_MEASURE": "None", "CC_GROUPING_MIN_TIME_IOU": "None"}
    data.update(conf_overrides or {})
    return types.SimpleNamespace(configuration=Configuration(data), params=params or {})


def check_steps_02_03(lib, name):
    """pre_ST3D 02 -> pickle (reference schema) -> unpickle -> pre_ST3D 03, against the reference's G3/G4 golden."""
    use_library(lib)
    from lecturemath_amd import png
    g, spec, frames = lm_checks.load_stream(name)
    compressed = [png.encode_gray8(f) for f in frames]
    times = [1000.0 * i for i in range(len(frames))]
    idxs = list(range(len(frames)))
    s02 = load_script("pre_ST3D_v3.0_02_cc_analaysis.py")
    s03 = load_script("pre_ST3D_v3.0_03_cc_grouping.py")
    proc = fake_process({"CC_STABILITY_MAX_GAP": str(spec["gap2"])})
    t, i, est = s02.process_input(proc, (times, idxs, compressed))
    # step-02 state as the reference exposes it
    assert est.tempo_count == int(g["tempo_count"])
    assert [[(int(a), int(b)) for a, b in fl] for fl in est.unique_cc_frames] == lm_checks.unrag(g["ucf"], g["ucf_off"])
    assert [[(int(u), int(c.cc_id)) for u, c in fr] for fr in est.cc_idx_per_frame] == lm_checks.unrag(g["cipf"], g["cipf_off"])
    assert list(est.cc_active) == list(g["active"])
    u0 = est.unique_cc_objects[0]
    assert isinstance(u0.min_x, np.int32) and u0.img.dtype == np.uint8
    blob = pickle.dumps((t, i, est), protocol=pickle.HIGHEST_PROTOCOL)
    t2, i2, est2 = pickle.loads(blob)
    proc3 = fake_process({"CC_STABILITY_MAX_GAP": str(spec["gap3"])})
    rec, conf, st3d = s03.process_input(proc3, (t2, i2, est2))
    group_ages, conflicts = conf
    ng = len(group_ages)
    assert [[(a,) for a in group_ages[k]] for k in range(ng)] == lm_checks.unrag(g["ages"], g["ages_off"])
    got = sorted((k, o, d["matched"], d["unmatched"], d["area_union"], float(d["area_intersection"]))
                 for k in conflicts for o, d in conflicts[k].items())
    assert [tuple(float(v) for v in r) for r in got] == sorted(tuple(float(v) for v in row) for row in g["conflicts"])
    assert (np.asarray([st3d.cc_group_boundaries[k] for k in range(ng)], np.int64).reshape(-1, 4) == g["bounds"]).all()
    gi = np.concatenate([im.ravel() for k in range(ng) for im in st3d.cc_group_images[k]]) if ng else np.zeros(0, np.uint8)
    assert (gi == g["gimg"]).all()
    clean = np.stack([png.decode_gray8(c) for c in rec[2]])
    assert (np.packbits(clean == 255, axis=2) == g["clean_packed"]).all()
    assert (clean[(clean != 0) & (clean != 255)] == g["clean_other_val"]).all()
    assert st3d.width == spec["w"] and st3d.height == spec["h"] and rec[0] == times
    check_step_04(name, [rec, conf, st3d], clean)
    return blob


def g7(name):
    import json
    g = np.load(os.path.join(lm_checks.GOLD, "g7_step04_%s.npz" % name))
    return g, json.loads(bytes(g["params"]).decode())


def check_step_04(name, step03_outputs, clean=None):
    """pre_ST3D 04 (deletion-event segmentation) on step-03 outputs vs the reference's intervals (G7), three parameter sets;
    binary sums of the reconstructed frames vs VideoSegmenter.compute_binary_sums of the reference."""
    if DROPIN not in sys.path:
        sys.path.insert(0, DROPIN)
    from AccessMath.preprocessing.content.video_segmenter import VideoSegmenter
    g, param_sets = g7(name)
    s04 = load_script("pre_ST3D_v3.0_04_vid_segmentation.py")
    for k, values in enumerate(param_sets):
        proc = fake_process(dict({key: str(v) for key, v in values.items()}, VIDEO_SEGMENTATION_METHOD="3"))
        intervals = s04.process_input(proc, step03_outputs)
        assert [tuple(int(v) for v in iv) for iv in intervals] == [tuple(int(v) for v in iv) for iv in g["intervals_%d" % k]]
    if clean is not None:
        sums = VideoSegmenter.compute_binary_sums(list(clean))
        assert [float(v) for v in sums] == [float(v) for v in g["sums"]]


def check_step_05_ties(lib, cases=None):
    """Step 05 on the tie-heavy synthetic structures of G8b (hundreds of groups per segment, many starting at the same frame and
    overlapping in chains): the groups the reference draws -- decided by its tie-break inside conflict sets -- keyframes and CC
    time lists, for two segmentations per case."""
    import json
    use_library(lib)
    from AccessMath.data.space_time_struct import SpaceTimeStruct
    from AccessMath.preprocessing.content.keyframe_extractor import KeyframeExtractor
    g = np.load(os.path.join(lm_checks.GOLD, "g8b_step05_ties.npz"))
    for c in (range(int(g["n_cases"])) if cases is None else cases):
        meta = json.loads(bytes(g["meta_%d" % c]).decode())
        ng, off = meta["n_groups"], g["ages_off_%d" % c]
        ages = {k: [int(v) for v in g["ages_%d" % c][off[k]:off[k + 1]]] for k in range(ng)}
        bounds = {k: tuple(int(v) for v in g["bounds_%d" % c][k]) for k in range(ng)}
        bits = np.unpackbits(g["images_%d" % c])
        images, pos = {}, 0
        for k in range(ng):
            w, h = bounds[k][1] - bounds[k][0] + 1, bounds[k][3] - bounds[k][2] + 1
            images[k] = []
            for _ in range(len(ages[k]) - 1):
                images[k].append((bits[pos:pos + w * h].reshape(h, w) * 255).astype(np.uint8))
                pos += w * h
        n = meta["n"]
        st3d = SpaceTimeStruct([1000.0 * i for i in range(n)], list(range(n)), meta["h"], meta["w"], ages, images, bounds)
        for k, segs in enumerate(meta["segs"]):
            keyframes, cc_times = KeyframeExtractor.GenerateFromST3DForIntervals(st3d, [tuple(sg) for sg in segs], False)
            kf = np.stack(keyframes)
            assert (np.packbits(kf[..., 0] == 255, axis=2) == g["keyframes_%d_%d" % (c, k)]).all(), (c, k)
            flat = np.asarray([(sidx, *t) for sidx, lst in enumerate(cc_times) for t in lst], np.float64).reshape(-1, 6)
            assert flat.shape == g["times_%d_%d" % (c, k)].shape and (flat == g["times_%d_%d" % (c, k)]).all(), (c, k)


def check_step_05(lib, name):
    """Step 05 core (KeyframeExtractor.GenerateFromST3DForIntervals) on the reference's own step-03 outputs (G4 fixture) vs
    the reference's keyframes and CC times (G8), three segmentations; plus the entry point's interval arithmetic."""
    import json
    use_library(lib)
    from AccessMath.data.space_time_struct import SpaceTimeStruct
    from AccessMath.preprocessing.content.keyframe_extractor import KeyframeExtractor
    g4, spec, _ = lm_checks.load_stream(name)
    ng = len(g4["gimg_count"])
    ages = {k: [int(v[0]) for v in lm_checks.unrag(g4["ages"], g4["ages_off"])[k]] for k in range(ng)}
    bounds = {k: tuple(int(v) for v in g4["bounds"][k]) for k in range(ng)}
    images, off = {}, 0
    for k in range(ng):
        w, h = bounds[k][1] - bounds[k][0] + 1, bounds[k][3] - bounds[k][2] + 1
        images[k] = []
        for _ in range(int(g4["gimg_count"][k])):
            images[k].append(g4["gimg"][off:off + w * h].reshape(h, w).copy())
            off += w * h
    g = np.load(os.path.join(lm_checks.GOLD, "g8_step05_%s.npz" % name))
    n = int(g["n_frames"])
    times = [1000.0 * i for i in range(n)]
    st3d = SpaceTimeStruct(times, list(range(n)), spec["h"], spec["w"], ages, images, bounds)
    for k, segs in enumerate(json.loads(bytes(g["segments"]).decode())):
        segs = [tuple(sg) for sg in segs]
        keyframes, cc_times = KeyframeExtractor.GenerateFromST3DForIntervals(st3d, segs, False)
        kf = np.stack(keyframes)
        assert kf.dtype == np.uint8 and (kf[..., 0] == kf[..., 1]).all() and (kf[..., 0] == kf[..., 2]).all()
        assert (np.packbits(kf[..., 0] == 255, axis=2) == g["keyframes_%d" % k]).all()
        flat = [(sidx, *t) for sidx, lst in enumerate(cc_times) for t in lst]
        assert (np.asarray(flat, np.float64).reshape(-1, 6) == g["times_%d" % k]).all()
    s05 = load_script("pre_ST3D_v3.0_05_generate_summary.py")
    (indices, stimes, kfs), = s05.process_input(types.SimpleNamespace(database=None), [st3d, segs])
    assert indices == [sg[1] for sg in segs] and stimes == [times[sg[1]] for sg in segs] and len(kfs) == len(segs)
    # the exported intervals: boundaries in the middle of the gaps (written out the long way here)
    idx_iv, time_iv, _, _ = s05.tiling_intervals(st3d, segs)
    want_idx, want_time, prev_i, prev_t = [], [], 0, 0
    for k, (_, last) in enumerate(segs):
        if k + 1 < len(segs):
            nxt = segs[k + 1][0]
            end_i, end_t = int((st3d.frame_indices[last] + st3d.frame_indices[nxt]) / 2), (st3d.frame_times[last] + st3d.frame_times[nxt]) / 2.0
        else:
            end_i, end_t = st3d.frame_indices[last], st3d.frame_times[last]
        want_idx.append((prev_i, end_i))
        want_time.append((prev_t, end_t))
        prev_i, prev_t = end_i, end_t
    assert idx_iv == want_idx and time_iv == want_time


def check_pipeline(lib, name):
    """lecturemath_amd.pipeline.LecturePipeline (steps 02-05 in one process, device-resident hand-off) on a golden stream:
    intervals == the reference's step 04 (G7, parameter set 2) and keyframes == the reference's step 05 on those intervals (G8)."""
    import json
    use_library(lib)
    from lecturemath_amd.pipeline import LecturePipeline
    g3, spec, frames = lm_checks.load_stream(name)
    g7_, params = g7(name)
    g8 = np.load(os.path.join(lm_checks.GOLD, "g8_step05_%s.npz" % name))
    conf = dict(params[2], CC_STABILITY_MAX_GAP=spec["gap2"])
    pipe = LecturePipeline(spec["w"], spec["h"], conf=conf, lib=lib)
    n = len(frames)
    half = n // 2
    pipe.add_binary_frames(np.stack(frames[:half]), [1000.0 * i for i in range(half)], list(range(half)))
    pipe.add_binary_frames(np.stack(frames[half:]), [1000.0 * i for i in range(half, n)], list(range(half, n)))
    # the fixtures ran step 02 and step 03 with different CC_STABILITY_MAX_GAP values: switch the key between the two
    pipe.configuration.data["CC_STABILITY_MAX_GAP"] = str(spec["gap3"])
    out = pipe.finish(reconstructed_png=True)
    assert [tuple(int(v) for v in iv) for iv in out["intervals"]] == [tuple(int(v) for v in iv) for iv in g7_["intervals_2"]]
    assert json.loads(bytes(g8["segments"]).decode())[0] == [list(iv) for iv in out["intervals"]]
    kf = np.stack(out["keyframes"])
    assert (np.packbits(kf[..., 0] == 255, axis=2) == g8["keyframes_0"]).all()
    from lecturemath_amd import png
    clean = np.stack([png.decode_gray8(c) for c in out["reconstructed_png"]])
    assert (np.packbits(clean == 255, axis=2) == g3["clean_packed"]).all()
    dev = lib and out["reconstructed_device"](0, min(4, n))
    assert (pipe.be.to_host(dev) == clean[:min(4, n)]).all()


def check_image_pairs(lib, seed=3, n=60, side=96):
    """device.image_pairs_overlap vs a numpy all-pairs test: boxes at arbitrary (unaligned) positions, widths around the
    32-bit word boundaries, disjoint ink inside overlapping boxes."""
    from lecturemath_amd import device
    rng = np.random.default_rng(seed)
    boxes, images = [], []
    for k in range(n):
        w, h = int(rng.choice([1, 5, 31, 32, 33, 40, 64, 65])), int(rng.integers(1, 20))
        x0, y0 = int(rng.integers(0, side - 1)), int(rng.integers(0, side - 1))
        img = (rng.random((h, w)) < (0.15 if k % 3 else 0.6)).astype(np.uint8) * 255
        boxes.append((x0, x0 + w - 1, y0, y0 + h - 1))
        images.append(img)
    want = []
    canvas = []
    for (x0, x1, y0, y1), img in zip(boxes, images):
        c = np.zeros((side + 80, side + 80), bool)
        c[y0:y1 + 1, x0:x1 + 1] = img > 0
        canvas.append(c)
    for i in range(n):
        for j in range(i + 1, n):
            if (canvas[i] & canvas[j]).any():
                want.append((i, j))
    assert device.image_pairs_overlap(boxes, images, lib) == want and len(want) > 10


def check_step_04_from_golden(name):
    """Host-only: the step-04 drop-in fed with the reference's own step-03 outputs (ages, boundaries) from the G7 fixture."""
    if DROPIN not in sys.path:
        sys.path.insert(0, DROPIN)
    from AccessMath.data.space_time_struct import SpaceTimeStruct
    g, _ = g7(name)
    n = int(g["n_frames"])
    ages = {k: [int(v) for v in g["ages"][g["ages_off"][k]:g["ages_off"][k + 1]]] for k in range(len(g["ages_off"]) - 1)}
    bounds = {k: tuple(int(v) for v in g["bounds"][k]) for k in range(len(g["bounds"]))}
    st3d = SpaceTimeStruct([float(i) for i in range(n)], list(range(n)), int(g["h"]), int(g["w"]), ages, {}, bounds)
    check_step_04(name, [([float(i) for i in range(n)], list(range(n)), []), (ages, {}), st3d])


def check_labeler(lib):
    use_library(lib)
    from AccessMath.preprocessing.content.labeler import Labeler
    g = np.load(os.path.join(lm_checks.GOLD, "g1_label.npz"))
    for i in (3, 6, 8, 17):
        ccs = Labeler.extractConnectedComponents(g["img%d" % i])
        rec = np.asarray([(c.cc_id, c.min_x, c.max_x, c.min_y, c.max_y, c.size) for c in ccs], np.int32).reshape(-1, 6)
        assert (rec == g["rec%d" % i]).all()
        flat = np.concatenate([c.img.ravel() for c in ccs]) if ccs else np.zeros(0, np.uint8)
        assert (flat == g["crops%d" % i]).all()
        if ccs:
            assert ccs[0].start_time == 0.0 and ccs[0].img.dtype == np.uint8
    # filter_small=False keeps every label; is_labeled=True goes through the CC_AgeBoundaries export
    from oracle import cc as occ
    img = g["img3"]
    labels, n = occ.label4(img)
    every = Labeler.extractConnectedComponents(img, filter_small=False)
    assert len(every) == n and [c.cc_id for c in every] == list(range(n))
    pre = Labeler.extractSpatioTemporalContent(labels, np.zeros(img.shape, np.float32), True, True)
    ref = Labeler.extractConnectedComponents(img)
    assert [(c.cc_id, c.size) for c in pre] == [(c.cc_id, c.size) for c in ref]


FCN_EDGE = 0.0078433            # trunc(sigmoid(x) * 255) >= 128 flips at x = ln(128 / 127)
FCN_EDGE_TOL_F16X3 = 1e-4       # the first engine / planar-f16x3 (measured 2e-6) -- the tiny G5 networks run on it
FCN_EDGE_TOL_MIXED = 2.5e-4     # the shipped per-layer assignment of the planar engine (profiles/r04_fcn_formats.*: <= 2.1e-4 at 1080p)


def assert_binarization(got, exp, logits, tol, what="binary"):
    """A thresholded output may differ from the reference's ONLY where the reference's logit lies within `tol` -- the engine's own error
    bound, not the 1e-3 bar -- of the decision edge.  Returns the number of such flips (exempted AND different)."""
    diff = np.asarray(got) != np.asarray(exp)
    band = np.abs(np.asarray(logits, np.float64) - FCN_EDGE) < tol
    away = diff & ~band
    if away.any():
        ys, xs = np.nonzero(away)
        raise AssertionError("%s: %d of %d pixels differ outside the %.1e band around the threshold edge; first at (%d, %d), logit %r" % (
            what, int(away.sum()), diff.size, tol, ys[0], xs[0], float(np.asarray(logits)[ys[0], xs[0]])))
    return int(diff.sum())


def check_fcn_class(lib, name="k7_70x94", worker=True):
    """FCN_LectureNet.CreateFromConfig / load_state_dict / binarize and the step-01 worker vs the reference's outputs."""
    use_library(lib)
    import PIL.Image
    from AM_CommonTools.configuration.configuration import Configuration
    from AccessMath.lecturenet_v1.FCN_lecturenet import FCN_LectureNet
    from AccessMath.preprocessing.video_worker.FCN_lecturenet_binarizer import FCN_LectureNet_Binarizer
    from lecturemath_amd import fcn, png
    g = np.load(os.path.join(lm_checks.GOLD, "g5_fcn_%s.npz" % name))
    conf = Configuration({key: str(int(v)) for (key, _), v in zip(fcn.WIDTH_KEYS, g["widths"])})
    conf.set("FCN_BINARIZER_NET_PIXEL_KERNEL_SIZE", str(int(g["pk"])))
    net = FCN_LectureNet.CreateFromConfig(conf, 3, False)
    net.load_state_dict({k[3:]: g[k] for k in g.files if k.startswith("sd.")})
    net = net.eval().cuda()
    binary, text_mask, rec_img = net.binarize(PIL.Image.fromarray(g["rgb"]), return_others=True, force_binary=True)
    # thresholded outputs may differ only where the logit sits within the fp32 tolerance of the decision edge
    tol = FCN_EDGE_TOL_MIXED if net._get_engine(*g["rgb"].shape[:2]).planar else FCN_EDGE_TOL_F16X3
    flips = assert_binarization(binary, g["binary"], g["out"][0, 0], tol) + assert_binarization(text_mask, g["text_mask"], g["text"][0, 0], tol, "text mask")
    assert flips <= 4, flips          # of ~6,600 pixels
    assert np.abs(rec_img.astype(np.int32) - g["rec_img"].astype(np.int32)).max() <= 1
    if not worker:
        return
    worker = FCN_LectureNet_Binarizer(net)
    worker.initialize(g["rgb"].shape[1], g["rgb"].shape[0])
    worker.handleFrame(np.ascontiguousarray(g["rgb"][:, :, ::-1]), None, 0, 1000.0, 1000.0, 30)
    assert worker.frame_times == [1000.0] and worker.frame_indices == [30] and worker.getWorkName()
    dec = png.decode_gray8(worker.compressed_frames[0])
    assert_binarization(dec, 255 - g["binary"], g["out"][0, 0], tol, "worker frame")


def check_step01_entry_points(lib, tmp_dir, name="k7_70x94", n_frames=3, tool=True):
    """The step-01 script's callbacks as the harness calls them (pre_ST3D_v3.0_01_binarize.py:20-55: get_worker builds the
    network from the configuration and a state_dict written with torch.save, the sampler feeds worker.handleFrame, get_results
    hands back (frame_times, frame_indices, compressed_frames)) and the one-image tool test_FCN_binarizer.py (:13-59) through
    its main() with the reference's argv -- both against the reference's binarization of the G5 fixture."""
    use_library(lib)
    import importlib.util
    import types
    import PIL.Image
    import torch
    from AM_CommonTools.configuration.configuration import Configuration
    from lecturemath_amd import fcn, png
    g = np.load(os.path.join(lm_checks.GOLD, "g5_fcn_%s.npz" % name))
    tmp_dir = str(tmp_dir)
    os.makedirs(os.path.join(tmp_dir, "models"), exist_ok=True)
    torch.save({k[3:]: torch.from_numpy(np.asarray(g[k])) for k in g.files if k.startswith("sd.")}, os.path.join(tmp_dir, "models", "net.dat"))
    conf_path = os.path.join(tmp_dir, "lecture.conf")
    with open(conf_path, "w") as f:
        f.write("# written by tests/dropin_checks.py\nOUTPUT_PATH = %s\nBINARIZATION_FCN_LECTURENET_DIR = models\n"
                "BINARIZATION_FCN_LECTURENET_FILENAME = net.dat\nFCN_BINARIZER_NET_PIXEL_KERNEL_SIZE = %d\n" % (tmp_dir, int(g["pk"])))
        for (key, _), v in zip(fcn.WIDTH_KEYS, g["widths"]):
            f.write("%s = %d\n" % (key, int(v)))

    def script(fname):
        spec = importlib.util.spec_from_file_location("lm_entry_" + fname.replace(".", "_"), os.path.join(DROPIN, fname))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    tol = FCN_EDGE_TOL_F16X3 if any(int(v) % 16 for v in g["widths"]) else FCN_EDGE_TOL_MIXED     # which engine the network gets
    # ---- step 01
    s01 = script("pre_ST3D_v3.0_01_binarize.py")
    process = types.SimpleNamespace(configuration=Configuration.from_file(conf_path), params={})
    worker = s01.get_worker(process)
    # the frames come from an exported-lecture folder through the image-list frame source, as
    # ConsoleUIProcess.start_image_list_preprocessing (console_ui_process.py:188-221) wires it
    import json
    from AccessMath.preprocessing.video_processor.image_list_processor import ImageListProcessor
    img_dir = os.path.join(tmp_dir, "export", "JPEGImages")
    os.makedirs(img_dir, exist_ok=True)
    for k in range(n_frames):
        PIL.Image.fromarray(g["rgb"]).save(os.path.join(img_dir, "%d.png" % (30 * (k + 1))))
    with open(os.path.join(img_dir, "index.json"), "w") as f:
        json.dump({str(30 * (k + 1)): {"video_time": 0.0, "frame_idx": 30 * (k + 1), "abs_time": 1000.0 * (k + 1), "video_idx": 0}
                   for k in range(n_frames)}, f)
    ImageListProcessor(os.path.join(tmp_dir, "export"), img_extension=".png").doProcessing(worker)
    times, indices, compressed = s01.get_results(worker)
    assert times == [1000.0 * (k + 1) for k in range(n_frames)] and indices == [30 * (k + 1) for k in range(n_frames)]
    assert len(compressed) == n_frames and not hasattr(worker, "lecture_net")
    for c in compressed:
        assert assert_binarization(png.decode_gray8(c), 255 - g["binary"], g["out"][0, 0], tol, "step-01 frame") <= 4
    if not tool:
        return
    # ---- test_FCN_binarizer.py
    PIL.Image.fromarray(g["rgb"]).save(os.path.join(tmp_dir, "in.png"))
    tool = script("test_FCN_binarizer.py")
    argv = sys.argv
    try:
        sys.argv = ["test_FCN_binarizer.py", conf_path, os.path.join(tmp_dir, "models", "net.dat"), os.path.join(tmp_dir, "in.png"),
                    os.path.join(tmp_dir, "out")]
        tool.main()
    finally:
        sys.argv = argv
    binary = np.asarray(PIL.Image.open(os.path.join(tmp_dir, "out_BIN.png")))
    text = np.asarray(PIL.Image.open(os.path.join(tmp_dir, "out_text.png")))
    bg = np.asarray(PIL.Image.open(os.path.join(tmp_dir, "out_bg.png")).convert("RGB"))
    assert assert_binarization(binary, g["binary"], g["out"][0, 0], tol, "tool binary") <= 4
    assert assert_binarization(text, g["text_mask"], g["text"][0, 0], tol, "tool text mask") <= 4
    assert np.abs(bg[:, :, ::-1].astype(np.int32) - g["rec_img"].astype(np.int32)).max() <= 1       # the tool writes RGB, rec_img is BGR


def check_rebuilt_binary_images(lib, name="short_gap_jitter"):
    """CCStabilityEstimator.rebuilt_binary_images (cc_stability_estimator.py:166-179; dead work in the reference's step 03, kept
    for callers): every kept CC painted back gives the input frame minus the CCs below MIN_CC_PIXELS."""
    use_library(lib)
    from AccessMath.preprocessing.content.cc_stability_estimator import CCStabilityEstimator
    from oracle import cc as occ
    g, spec, frames = lm_checks.load_stream(name)
    est = CCStabilityEstimator(spec["w"], spec["h"], 0.85, 0.85, spec["gap2"], False)
    for f in frames[:12]:
        est.add_frame(f, True)
    est.finish_processing()
    rebuilt = est.rebuilt_binary_images()
    assert len(rebuilt) == 12
    for f, r in zip(frames, rebuilt):
        labels, n = occ.label4(f)
        rec, crops = occ.extract(labels, n)
        exp = np.zeros_like(f)
        for (cc_id, x0, x1, y0, y1, size), crop in zip(rec, crops):
            exp[y0:y1 + 1, x0:x1 + 1] |= crop
        assert r.dtype == np.uint8 and (r == exp).all()


def check_overlap_golden(lib):
    """G2: ConnectedComponent.getOverlapFMeasure of the drop-in class (AND + count on the crops) vs the reference's recall /
    precision, bit for bit as float64, for 200 CC pairs incl. disjoint ones (connected_component.py:202-250)."""
    use_library(lib)
    from AM_CommonTools.data.connected_component import ConnectedComponent
    g = np.load(os.path.join(lm_checks.GOLD, "g2_overlap.npz"))
    oa = ob = 0
    for k, bx in enumerate(g["boxes"]):
        ax0, ax1, ay0, ay1, asz, bx0, bx1, by0, by1, bsz = (int(v) for v in bx)
        ah, aw, bh, bw = ay1 - ay0 + 1, ax1 - ax0 + 1, by1 - by0 + 1, bx1 - bx0 + 1
        a = ConnectedComponent(0, ax0, ax1, ay0, ay1, asz, g["crops_a"][oa:oa + ah * aw].reshape(ah, aw))
        b = ConnectedComponent(1, bx0, bx1, by0, by1, bsz, g["crops_b"][ob:ob + bh * bw].reshape(bh, bw))
        oa += ah * aw
        ob += bh * bw
        r, p = a.getOverlapFMeasure(b, False, False)
        assert np.float64(r).view(np.int64) == g["recall"][k].view(np.int64) and np.float64(p).view(np.int64) == g["precision"][k].view(np.int64), k


def check_resize_golden(lib):
    """lm_resample_rgb8 (Pillow's LANCZOS, FCN_lecturenet.py:434-437) against images Pillow itself resized (G6b, recorded in the build
    container: halvings of even / odd sizes, a 3:1 reduction, an enlargement, a one-axis change; RGB and single channel), byte for
    byte; lm_upsample_nearest_u8 (:481-486) against the index formula."""
    from lecturemath_amd import resize
    g = np.load(os.path.join(lm_checks.GOLD, "g6b_lanczos.npz"))
    rs = resize.DeviceResizer(lib)
    for i, (h, w, oh, ow) in enumerate(g["cases"]):
        got = rs.be.to_host(rs.lanczos(g["in%d" % i], int(ow), int(oh)))
        assert got.shape == (oh, ow, 3) and (got == g["out%d" % i]).all(), i
        one = rs.be.to_host(rs.lanczos(np.ascontiguousarray(g["in%d" % i][:, :, 2]), int(ow), int(oh)))
        assert (one == g["out%d" % i][:, :, 2]).all(), i
    a = g["in1"]
    for (oh, ow) in ((2 * a.shape[0], 2 * a.shape[1]), (3 * a.shape[0], 4 * a.shape[1])):
        ys, xs = (np.arange(oh) * a.shape[0]) // oh, (np.arange(ow) * a.shape[1]) // ow
        assert (rs.be.to_host(rs.nearest(a, ow, oh)) == a[ys][:, xs]).all()
        assert (rs.be.to_host(rs.nearest(np.ascontiguousarray(a[:, :, 0]), ow, oh)) == a[:, :, 0][ys][:, xs]).all()


def check_fcn_4k_resize_branch(lib, shipped=False):
    """binarize() on a 3840x2160 frame (> 2.5 MP, FCN_lecturenet.py:435-437,481-494): PIL LANCZOS halving, FCN at 1080p,
    NEAREST x2 back -- against the oracle's torch forward on the same halved image.  Tiny network by default (the branch is
    about the resize plumbing), the shipped widths with shipped=True; a pixel may differ only where the oracle's logit lies within
    the ENGINE's error bound of the threshold edge, and the flips are counted."""
    use_library(lib)
    import PIL.Image
    import torch
    from AM_CommonTools.configuration.configuration import Configuration
    from AccessMath.lecturenet_v1.FCN_lecturenet import FCN_LectureNet
    from lecturemath_amd import fcn, synth
    from oracle import fcn as ofcn
    widths = ofcn.SHIPPED_WIDTHS if shipped else (8,) * 18
    sd = ofcn.random_state_dict(widths, pixel_kernel=7 if shipped else 3, seed=5)
    conf = Configuration({key: str(v) for (key, _), v in zip(fcn.WIDTH_KEYS, widths)})
    if shipped:
        conf.set("FCN_BINARIZER_NET_PIXEL_KERNEL_SIZE", "7")
    net = FCN_LectureNet.CreateFromConfig(conf, 3, False)
    net.load_state_dict(sd)
    rgb, _ = synth.whiteboard_rgb(2160, 3840, n_glyphs=3000, seed=9)
    pil = PIL.Image.fromarray(rgb)
    binary, text_mask, rec_img = net.cuda().binarize(pil, return_others=True, force_binary=True)
    assert binary.shape == (2160, 3840) and rec_img.shape == (2160, 3840, 3)
    half = np.asarray(pil.resize((1920, 1080), PIL.Image.LANCZOS))
    from lecturemath_amd import resize
    rs = resize.DeviceResizer(lib)
    assert (rs.be.to_host(rs.lanczos(rgb, 1920, 1080)) == half).all()      # the device's LANCZOS halving of the 4K frame == Pillow's
    with torch.no_grad():
        o, t, r = ofcn.forward(sd, ofcn.prepare_image(half))
    exp = ((torch.sigmoid(o)[0, 0].numpy() * 255).astype(np.uint8) >= 128).astype(np.uint8) * 255
    assert (binary[::2, ::2] == binary[1::2, 1::2]).all() and (binary[::2, ::2] == binary[::2, 1::2]).all()          # NEAREST x2 structure
    flips = assert_binarization(binary[::2, ::2], exp, o[0, 0].numpy(), FCN_EDGE_TOL_MIXED if shipped else FCN_EDGE_TOL_F16X3, "4K binary")
    assert flips <= (1000 if shipped else 8), flips        # random-init logits crowd the edge (std ~0.1): 5e-4 of the 1080p frame at most
