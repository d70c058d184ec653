"""CPU: kernel LOGIC of the product's HIP sources, compiled for the CPU by the test-only fiber emulator
(tests/hipemu).  This is a development aid for a GPU-less container -- the parity tests proper are the
gpu-marked ones, which run the same checks through the real gfx950 library."""
import os

import numpy as np
import pytest

import lm_checks
from lecturemath_amd import device, synth


@pytest.mark.parametrize("i", range(18))
def test_label_stats_golden(emu_lib, oracle_built, i):
    lm_checks.check_g1_frame(emu_lib, i)


def test_threshold_golden(emu_lib):
    g = np.load(os.path.join(lm_checks.GOLD, "g6_threshold.npz"))
    lab = device.FrameLabeler(96, 64, 1, emu_lib)
    out = lab.threshold_invert(g["logits"])
    far = np.abs(g["logits"] - 0.0078433) > 1e-4
    assert (out[far] == g["expected"][far]).all()
    assert int((out != g["expected"]).sum()) <= 2
    lab.close()


def test_threshold_comparison_form(emu_lib):
    lm_checks.check_threshold_paths(emu_lib)
    lm_checks.check_label_logits_fused(emu_lib)


def test_stream_vs_oracle_small(emu_lib, oracle_built):
    frames = list(synth.binary_stream(24, 96, 160, seed=5, glyphs_per_add=4, erase_every=9, jitter_p=0.4, occluder=True,
                                      max_ext=16))
    r = lm_checks.check_stream_oracle(emu_lib, frames, max_gap=5, max_batch=5)
    assert len(r["unique_recs"]) > 20


def test_stream_run_logits(emu_lib, oracle_built):
    lm_checks.check_stream_run_logits(emu_lib)


def test_stream_match_paths_agree(emu_lib, oracle_built):
    lm_checks.check_stream_match_paths(emu_lib, n_frames=48)


def test_stream_golden_short_gap(emu_lib):
    lm_checks.check_stream_golden(emu_lib, "short_gap_jitter", max_batch=16)


def test_drop_in_cc_age_boundaries(emu_lib, oracle_built):
    rng = np.random.default_rng(3)
    img = ((rng.random((40, 75)) < 0.45) * 255).astype(np.uint8)
    labels, n = oracle_built.label4(img)
    ages = rng.random(img.shape).astype(np.float32)
    outs = [np.zeros(n, np.int32) for _ in range(5)]
    oa = np.zeros(n, np.float32)
    rc = emu_lib.CC_AgeBoundaries(labels.ctypes.data, ages.ctypes.data, 75, 40, n, *[o.ctypes.data for o in outs], oa.ctypes.data)
    assert rc == 0
    exp = oracle_built.age_boundaries(labels, ages, n)
    for a, b in zip(outs + [oa], exp):
        assert (a == b).all()
    # signed ages: the reference's sequential "-1 = nothing yet" rule (accessmath_lib.c:405-407) lets a negative age be
    # overwritten by whatever pixel follows; labels ending on a negative age keep it
    for seed in (4, 5):
        ages = np.random.default_rng(seed).normal(0.2, 1.0, img.shape).astype(np.float32)
        oa = np.zeros(n, np.float32)
        emu_lib.CC_AgeBoundaries(labels.ctypes.data, ages.ctypes.data, img.shape[1], img.shape[0], n, *[o.ctypes.data for o in outs], oa.ctypes.data)
        exp_age = oracle_built.age_boundaries(labels, ages, n)[5]
        assert (exp_age < 0).any() and (exp_age >= 0).any() and (oa == exp_age).all()


def test_label_host(emu_lib, oracle_built):
    img = synth.glyph_mask(70, 130, 30, seed=9)
    out = np.zeros(img.shape, np.int32)
    n = emu_lib.lm_label_host(img.ctypes.data, 130, 70, out.ctypes.data)
    l, m = oracle_built.label4(img)
    assert n == m and (out == l).all()


def test_capacity_error_is_reported(emu_lib):
    frames = np.stack(list(synth.binary_stream(6, 64, 96, seed=1, glyphs_per_add=6, erase_every=0, max_ext=12)))
    fs = device.FrameStream(96, 64, 6, max_batch=3, max_ccs=4, max_crop_words=64, lib=emu_lib)
    fs.push(frames)
    from lecturemath_amd import _lib
    with pytest.raises(_lib.LecturemathError) as e:
        fs.counters()
    assert e.value.code == _lib.LM_ERR_CAPACITY
    fs.close()


def test_dense_wide_noise_band_fallback(emu_lib, oracle_built):
    """> 8192 runs inside one 64-row band (LDS forest falls back to L2 atomics) and > 512 labels per stats tile."""
    rng = np.random.default_rng(21)
    img = ((rng.random((70, 1100)) < 0.5) * 255).astype(np.uint8)
    lm_checks.check_label_vs_oracle(emu_lib, img)


def test_label_batch_in_parts(emu_lib, oracle_built):
    lm_checks.check_label_batch_in_parts(emu_lib)


def test_label_wide_frame_16_row_bands(emu_lib, oracle_built):
    """Frames wider than 2048 px (4K: WW = 60 words) are labelled in 16-row bands (the LDS tuning for wide rows)."""
    rng = np.random.default_rng(23)
    img = ((rng.random((37, 2200)) < 0.35) * 255).astype(np.uint8)
    lm_checks.check_label_vs_oracle(emu_lib, img)


def test_grouping_crowded_tiles_vs_oracle(emu_lib, oracle_built):
    r = lm_checks.check_grouping_oracle(emu_lib, lm_checks.dot_grid_stream(n_frames=4, h=40, w=520))
    assert len(r["cc_groups"]) > 200


def test_stream_large_components(emu_lib, oracle_built):
    """A component whose crop has thousands of words next to glyph-sized ones: crop emission, twin comparison and pair
    evaluation distribute their work over crop words, not CCs."""
    lm_checks.check_stream_large_components(emu_lib, n_frames=8)


def test_stream_many_small_ccs(emu_lib, oracle_built):
    """~1,000 kept CCs and ~1,300 crop words per frame: lm_k_emit's workgroups start in the middle of the frame's crop words
    (two-round 256-way search for their first CC) and walk through several windows of staged CC descriptors; lm_k_select
    passes over more than one wave of labels per thread group; lm_k_mb_twin_cmp's workgroups likewise."""
    frames = lm_checks.churn_stream(n_frames=5, empty_every=3)
    r = lm_checks.check_stream_oracle(emu_lib, frames, max_gap=3, max_batch=5, max_ccs=1 << 14, max_crop_words=1 << 16)
    assert max(len(x) for x in r["cc_idx_per_frame"]) > 600


def test_stream_thousands_of_ccs_per_frame(emu_lib, oracle_built):
    """~7,000 kept CCs per frame: lm_k_select carries its scan over more than one pass of 4,096 labels, lm_k_emit's first-CC search
    runs over more than 256 x 16 entries, the joins see more than one chunk of sources."""
    frames = lm_checks.dot_grid_stream(n_frames=3, h=300, w=1000, seed=5)
    r = lm_checks.check_stream_oracle(emu_lib, frames, max_gap=3, max_batch=3, max_ccs=1 << 16, max_crop_words=1 << 18)
    assert max(len(x) for x in r["cc_idx_per_frame"]) > 6000


def test_stream_threshold_edges(emu_lib, oracle_built):
    lm_checks.check_stream_threshold_edges(emu_lib)


def test_render_overlapping_group_images(emu_lib, oracle_built):
    lm_checks.check_render_wraparound(emu_lib)


def test_legacy_exports_vs_reference_c(emu_lib, oracle_built):
    lm_checks.check_legacy_exports(emu_lib)


def test_grouping_golden_short_gap(emu_lib):
    lm_checks.check_grouping_golden(emu_lib, "short_gap_jitter")


def test_grouping_golden_host_threads(emu_lib):
    """lm_group_run splits its two largest host loops (entry lists, conflict rows) over threads on long lectures; here the split
    is forced (LM_GROUP_THREADS, read once per process: a child) on a short stream, more parts than some ranges have items."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
             "import lm_checks\nfrom lecturemath_amd import _lib\n"
             "lib = _lib.load(%r)\n"
             "lm_checks.check_grouping_golden(lib, 'short_gap_jitter')\nprint('threads ok')\n") % (
                 root, os.path.join(root, "tests"), emu_lib.path)
    env = dict(os.environ, LM_GROUP_THREADS="7")
    r = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "threads ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
def test_fcn_golden_tiny(emu_lib, precision):
    assert lm_checks.check_fcn_golden(emu_lib, "k7_70x94", precision=precision) < 1e-4


@pytest.mark.parametrize("precision,tol", [("mixed", 5e-4)])       # (planar-f16x3: on the GPU, test_fcn_golden_planar_engine)
def test_fcn_golden_planar_engine_emulated(emu_lib, precision, tol):
    """the planar engine's kernels (CPU emulation of the same sources) against the REFERENCE module's outputs (G5 wide case)"""
    assert lm_checks.check_fcn_golden(emu_lib, "k7_66x130_wide", tol=tol, precision=precision, require_planar=True) <= tol


PLANAR_TINY_WIDTHS = (16, 32, 16, 16, 16, 32, 32, 16, 16, 16, 16, 16, 16, 48, 16, 32, 32, 16)       # channel blocks of 1, 2 and 3 tiles; one merged-dx transposed conv


@pytest.mark.parametrize("precision,tol", [("mixed", 5e-4)])        # (planar-f16x3 at 1e-5: on the GPU, test_fcn_shipped_config_vs_oracle / _1080p_)
def test_fcn_planar_tiny_vs_oracle(emu_lib, precision, tol):
    """The planar FCN engine (csrc/lm_fcn2.hip: gather-GEMM on 16x16x32 MFMA tiles, planar f16 activations, LDS-DMA staging, pair
    planes) on the CPU emulator against the torch oracle: an odd-sized frame (every output_size border, floor pooling), then a
    smaller frame through the SAME engine (halos / tile overhang re-zeroed)."""
    import torch
    from lecturemath_amd import fcn, synth
    from oracle import fcn as ofcn
    sd = ofcn.random_state_dict(PLANAR_TINY_WIDTHS, pixel_kernel=7, seed=1)
    eng = fcn.FcnEngine(PLANAR_TINY_WIDTHS, 7, 3, 45, 61, emu_lib, precision=precision)
    assert eng.planar
    eng.load_state_dict(sd)
    for h, w in ((45, 61), (33, 40)):
        rgb, _ = synth.whiteboard_rgb(h, w, n_glyphs=20, seed=4)
        out, text, rec = (np.asarray(x) for x in eng.forward(rgb))
        with torch.no_grad():
            o, t, r = ofcn.forward(sd, ofcn.prepare_image(rgb))
        assert np.abs(out - o[0, 0].numpy()).max() <= tol and np.abs(text - t[0, 0].numpy()).max() <= tol and np.abs(rec - r[0].numpy()).max() <= tol, (h, w)
    eng.close()


def test_emulated_library_under_asan_ubsan(oracle_built, tmp_path):
    """The product's HIP sources built for the CPU with -fsanitize=address,undefined (tests/hipemu `make asan`) and run in a
    child interpreter with the ASan runtime preloaded: the FCN forward pass (f16x3; every dynamic-LDS kernel of lm_fcn.hip, on
    an odd-sized frame), a stream with large components and step 03 (group images, reconstruction).  The emulator allocates
    the dynamic LDS of every launch at exactly the size the launch asked for and fills it with NaN patterns before every block,
    so an LDS index past a kernel's allocation aborts the child and LDS that is consumed without having been written poisons
    the compared results.  (The child stays clear of torch -- importing it under the ASan runtime takes a minute -- so the
    oracle's FCN outputs are computed here and handed over.)"""
    import subprocess
    import sys
    import torch
    from oracle import fcn as ofcn
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu")
    subprocess.check_call(["make", "-s", "-C", d, "asan"])
    rt = subprocess.check_output(["make", "-s", "-C", d, "asan-runtime"]).decode().strip()
    widths = (8, 16, 16, 8, 8, 8, 8, 8, 8, 8, 16, 16, 8, 8, 8, 8, 16, 8)
    sd = ofcn.random_state_dict(widths, pixel_kernel=7, seed=2)
    rgb = np.random.default_rng(1).integers(0, 256, (35, 53, 3), dtype=np.uint8)        # odd sizes: every output_size padding
    with torch.no_grad():
        o, t, r = ofcn.forward(sd, ofcn.prepare_image(rgb))
    fx = str(tmp_path / "fcn_case.npz")
    np.savez(fx, rgb=rgb, out=o[0, 0].numpy(), text=t[0, 0].numpy(), rec=r[0].numpy(), **{"sd." + k: v.numpy() for k, v in sd.items()})
    # the planar engine (lm_fcn2.hip) on a network it accepts (widths in multiples of 16)
    pwidths = (16,) * 13 + (48, 16, 32, 32, 16)
    psd = ofcn.random_state_dict(pwidths, pixel_kernel=7, seed=5)
    with torch.no_grad():
        po, pt, pr = ofcn.forward(psd, ofcn.prepare_image(rgb))
    pfx = str(tmp_path / "fcn_planar_case.npz")
    np.savez(pfx, out=po[0, 0].numpy(), text=pt[0, 0].numpy(), rec=pr[0].numpy(), **{"sd." + k: v.numpy() for k, v in psd.items()})
    child = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from lecturemath_amd import _lib, fcn\nimport lm_checks\n"
        "lib = _lib.load(%r)\n"
        "g = np.load(%r)\n"
        "eng = fcn.FcnEngine(%r, 7, 3, 35, 53, lib, precision='f16x3')\n"
        "eng.load_state_dict({k[3:]: g[k] for k in g.files if k.startswith('sd.')})\n"
        "out, text, rec = eng.forward(g['rgb'])\n"
        "assert np.abs(out - g['out']).max() < 1e-4 and np.abs(text - g['text']).max() < 1e-4 and np.abs(rec - g['rec']).max() < 1e-4\n"
        "p = np.load(%r)\n"
        "eng2 = fcn.FcnEngine(%r, 7, 3, 35, 53, lib, precision='mixed')\n"
        "assert eng2.planar\n"
        "eng2.load_state_dict({k[3:]: p[k] for k in p.files if k.startswith('sd.')})\n"
        "out, text, rec = eng2.forward(g['rgb'])\n"
        "assert np.abs(out - p['out']).max() < 5e-4 and np.abs(text - p['text']).max() < 5e-4 and np.abs(rec - p['rec']).max() < 5e-4\n"
        "lm_checks.check_stream_large_components(lib, n_frames=4)\n"
        "lm_checks.check_grouping_oracle(lib, lm_checks.dot_grid_stream(n_frames=4, h=40, w=520))\n"
        "print('sanitized run ok')\n" % (os.path.dirname(os.path.dirname(d)), os.path.dirname(d), os.path.join(d, "liblecturemath_emu_asan.so"), fx,
                                            widths, pfx, pwidths))
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:detect_stack_use_after_return=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "sanitized run ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
