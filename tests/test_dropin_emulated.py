"""CPU: the drop-in layer (reference module paths / script entry points) on the emulated library."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

import dropin_checks
import lm_checks


def test_png_roundtrip_and_filters():
    from lecturemath_amd import png
    rng = np.random.default_rng(0)
    img = (rng.random((37, 53)) < 0.3).astype(np.uint8) * 255
    assert (png.decode_gray8(png.encode_gray8(img)) == img).all()
    import io
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray((rng.random((40, 61)) * 255).astype(np.uint8)).save(buf, format="PNG", optimize=True)   # adaptive filters
    a = png.decode_gray8(np.frombuffer(buf.getvalue(), np.uint8))
    assert (a == np.array(Image.open(io.BytesIO(buf.getvalue())))).all()


def test_labeler(emu_lib, oracle_built):
    dropin_checks.check_labeler(emu_lib)


def test_steps_02_03_short_gap(emu_lib):
    dropin_checks.check_steps_02_03(emu_lib, "short_gap_jitter")


def test_resize_golden(emu_lib):
    dropin_checks.check_resize_golden(emu_lib)


def test_fcn_class(emu_lib):
    dropin_checks.check_fcn_class(emu_lib, worker=False)       # the worker: test_step01_entry_points below


REF = "/root/reference/ACCESS2021_release"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference only exists in the build container")
def test_pickle_interop_with_reference(emu_lib, tmp_path):
    """Our step-02 pickle is consumed by the REFERENCE's step 03 (its own classes, in a separate interpreter), and the
    reference's step-02 pickle by our step 03: both must reproduce the golden group ages."""
    blob = dropin_checks.check_steps_02_03(emu_lib, "short_gap_jitter")
    (tmp_path / "ours.dat").write_bytes(blob)
    code = r'''
import sys, pickle, io, contextlib, json
sys.path.insert(0, "%s"); sys.path.insert(0, "%s")
import ref_env; ref_env.enter()
from AccessMath.preprocessing.content.cc_stability_estimator import CCStabilityEstimator
t, i, est = pickle.load(open("%s", "rb"))
assert type(est).__module__ == "AccessMath.preprocessing.content.cc_stability_estimator"
with contextlib.redirect_stdout(io.StringIO()):
    est.split_stable_cc_by_gaps(4, 3)
    stable = est.get_stable_cc_idxs(3)
    tov, tot, aov = est.compute_overlapping_stable_cc(stable, 5)
    groups, gid = est.compute_groups(stable, tov, 0.5, None, None)
    ages, gpf = est.compute_groups_temporal_information(groups)
print(json.dumps([ages[k] for k in range(len(groups))]))
# and the other direction: a reference-made step-02 pickle
import numpy as np
sys.path.insert(0, "%s")
import lm_checks
g, spec, frames = lm_checks.load_stream("short_gap_jitter")
e2 = CCStabilityEstimator(spec["w"], spec["h"], 0.85, 0.85, spec["gap2"], False)
for f in frames: e2.add_frame(f, True)
e2.finish_processing = None
pickle.dump((t, i, e2), open("%s", "wb"), protocol=pickle.HIGHEST_PROTOCOL)
''' % (os.path.join(dropin_checks.ROOT, "tests", "golden"), dropin_checks.ROOT, tmp_path / "ours.dat",
       os.path.join(dropin_checks.ROOT, "tests"), tmp_path / "theirs.dat")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    g, spec, frames = lm_checks.load_stream("short_gap_jitter")
    ages_ref = json.loads(out.stdout.strip().splitlines()[-1])
    assert [[(a,) for a in lst] for lst in ages_ref] == lm_checks.unrag(g["ages"], g["ages_off"])
    # reference's pickle -> our step 03
    dropin_checks.use_library(emu_lib)
    t, i, est = pickle.loads((tmp_path / "theirs.dat").read_bytes())
    assert type(est).__module__ == "AccessMath.preprocessing.content.cc_stability_estimator" and hasattr(est, "_imported")
    s03 = dropin_checks.load_script("pre_ST3D_v3.0_03_cc_grouping.py")
    rec, conf, st3d = s03.process_input(dropin_checks.fake_process({"CC_STABILITY_MAX_GAP": str(spec["gap3"])}), (t, i, est))
    ng = len(conf[0])
    assert [[(a,) for a in conf[0][k]] for k in range(ng)] == lm_checks.unrag(g["ages"], g["ages_off"])
    gi = np.concatenate([im.ravel() for k in range(ng) for im in st3d.cc_group_images[k]])
    assert (gi == g["gimg"]).all()


@pytest.mark.parametrize("name", lm_checks.STREAMS)
def test_step_04_on_reference_step03_outputs(name):
    """Step 04 drop-in (host list logic) on the reference's own step-03 outputs: intervals of three parameter sets."""
    dropin_checks.check_step_04_from_golden(name)


def test_frame_sums_kernel(emu_lib):
    from lecturemath_amd import device
    rng = np.random.default_rng(5)
    frames = rng.integers(0, 256, (3, 37, 53), dtype=np.uint8)
    frames[1] = 255
    assert list(device.frame_sums(frames, emu_lib)) == [int(f.astype(np.int64).sum()) for f in frames]


def test_image_pairs_overlap(emu_lib):
    dropin_checks.check_image_pairs(emu_lib)


def test_step_05_short_gap(emu_lib):
    dropin_checks.check_step_05(emu_lib, "short_gap_jitter")


def test_step_05_tie_heavy_structures(emu_lib):
    dropin_checks.check_step_05_ties(emu_lib, cases=(0, 1))


def test_pipeline_short_gap(emu_lib):
    dropin_checks.check_pipeline(emu_lib, "short_gap_jitter")


def test_step01_entry_points(emu_lib, tmp_path):
    """pre_ST3D_v3.0_01_binarize.py get_worker / get_results with a torch.save'd state_dict, and test_FCN_binarizer.py main()."""
    dropin_checks.check_step01_entry_points(emu_lib, tmp_path, n_frames=1, tool=False)     # the one-image tool: GPU suite


def test_rebuilt_binary_images(emu_lib, oracle_built):
    dropin_checks.check_rebuilt_binary_images(emu_lib)


def test_overlap_golden_g2(emu_lib):
    dropin_checks.check_overlap_golden(emu_lib)
