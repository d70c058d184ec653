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
