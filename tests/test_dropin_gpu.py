"""GPU: the drop-in layer (reference module paths, classes, pre_ST3D_* process_input functions) on the gfx950 library."""
import pytest

import dropin_checks
import lm_checks

pytestmark = pytest.mark.gpu


def test_labeler(hip_lib, oracle_built):
    dropin_checks.check_labeler(hip_lib)


@pytest.mark.parametrize("name", lm_checks.STREAMS)
def test_steps_02_03(hip_lib, name):
    dropin_checks.check_steps_02_03(hip_lib, name)


@pytest.mark.parametrize("name", ["k7_70x94", "k3_135x240"])
def test_fcn_class_and_worker(hip_lib, name):
    dropin_checks.check_fcn_class(hip_lib, name)


def test_fcn_4k_resize_branch(hip_lib):
    dropin_checks.check_fcn_4k_resize_branch(hip_lib)


def test_frame_sums_device(hip_lib):
    """lm_frame_sums (step 04, compute_binary_sums) on 1080p frames incl. an unaligned view: exact integer sums."""
    import torch
    from lecturemath_amd import device
    dropin_checks.use_library(hip_lib)
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    frames = torch.randint(0, 256, (5, 1080, 1920), dtype=torch.uint8, device="cuda", generator=g)
    frames[2] = 255
    want = [int(v) for v in frames.to(torch.int64).sum(dim=(1, 2)).cpu()]
    assert list(device.frame_sums(frames, hip_lib)) == want
    odd = frames[:, :333, :1919].contiguous()       # frame bases and sizes that are not multiples of 16
    assert list(device.frame_sums(odd, hip_lib)) == [int(v) for v in odd.to(torch.int64).sum(dim=(1, 2)).cpu()]
    from AccessMath.preprocessing.content.video_segmenter import VideoSegmenter
    assert VideoSegmenter.compute_binary_sums(frames) == [v / 255 for v in want]


def test_image_pairs_overlap(hip_lib):
    dropin_checks.check_image_pairs(hip_lib)
    dropin_checks.check_image_pairs(hip_lib, seed=9, n=400, side=300)


@pytest.mark.parametrize("name", lm_checks.STREAMS)
def test_step_05(hip_lib, name):
    """Step 05 core: keyframes per video segment and their CC time lists vs the reference (G8)."""
    dropin_checks.check_step_05(hip_lib, name)


@pytest.mark.parametrize("name", lm_checks.STREAMS)
def test_pipeline(hip_lib, name):
    """Steps 02-05 in one process with device-resident hand-off: step-04 intervals and step-05 keyframes vs the reference."""
    dropin_checks.check_pipeline(hip_lib, name)


def test_pipeline_rgb_input(hip_lib):
    """FCN -> threshold -> invert -> step 02 inside the pipeline == the same stages run one by one."""
    import numpy as np
    import torch
    from lecturemath_amd import fcn, synth, _lib
    from lecturemath_amd.pipeline import LecturePipeline
    from oracle import fcn as ofcn
    dropin_checks.use_library(hip_lib)
    from AccessMath.lecturenet_v1.FCN_lecturenet import FCN_LectureNet
    h, w = 96, 160
    widths = [8, 8, 16, 16, 16, 16, 16, 16, 16, 16, 8, 8, 8, 8, 8, 8, 8, 8]
    sd = ofcn.random_state_dict(widths, pixel_kernel=3, seed=1)
    eng = fcn.FcnEngine(widths, 3, 3, h, w, hip_lib)
    eng.load_state_dict(sd)
    net = type("Net", (), {"forward_logits": staticmethod(lambda rgb: eng.forward(rgb))})()
    frames = [synth.whiteboard_rgb(h, w, n_glyphs=25, seed=s)[0] for s in range(4)]
    pipe = LecturePipeline(w, h, network=net, lib=hip_lib)
    pipe.add_rgb_frames(np.stack(frames))
    fs = pipe.estimator._stream
    assert fs.counters()["n_frames"] == 4
    # the same frames through the stages one by one: the stream must hold exactly the CCs of those binaries
    from oracle import cc as occ
    try:
        raw = fs.read(with_crops=False)
        for i, f in enumerate(frames):
            logits = eng.forward(f)[0]
            got = torch.empty((h, w), dtype=torch.uint8, device="cuda")
            hip_lib.check(hip_lib.lm_threshold_invert(_lib.ptr(logits), _lib.ptr(got), h * w, 128, torch.cuda.current_stream().cuda_stream))
            binary = got.cpu().numpy()
            assert set(np.unique(binary)) <= {0, 255}
            labels, n = occ.label4(binary)
            sizes = np.bincount(labels.ravel(), minlength=n + 1)[1:]
            kept = int((sizes >= 20).sum())
            assert int(raw["frame_off"][i + 1] - raw["frame_off"][i]) == kept
    finally:
        pipe.estimator._stream.close()
    eng.close()
