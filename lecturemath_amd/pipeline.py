"""Steps 01-05 of the ACCESS2021 pipeline in ONE process, frames staying in HBM between the steps (the
reference hands PNG-compressed frames from script to script through pickles; here nothing but the final products leaves
the device unless asked for).

    pipe = LecturePipeline(1920, 1080, conf={...reference config keys...}, network=fcn_or_None)
    pipe.add_rgb_frames(rgb_u8, times, indices)        # FCN -> threshold -> invert -> label/records/matching
    pipe.add_binary_frames(binary_u8, times, indices)  # already binarized input (ink = 255)
    out = pipe.finish()                                # grouping, video segmentation, keyframes

The stages are the drop-in classes and scripts themselves (lecturemath_amd/dropin: CCStabilityEstimator,
pre_ST3D_v3.0_03/04/05 process_input), so every product is the one the step-by-step scripts give; only the hand-offs differ
(add_frames_device / no PNG).  `out` holds: group_ages, conflicts, st3d (SpaceTimeStruct), intervals, keyframes, cc_times,
and -- lazily, on request -- the reconstructed frames as PNG byte strings like the reference's CC_RECONSTRUCTED_OUTPUT."""
import importlib.util
import os
import sys
import types

import numpy as np

_DROPIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dropin")

DEFAULT_CONF = {   # the shipped values (configs/FCN_LectureNet.conf), as the scripts read them
    "CC_STABILITY_MIN_RECALL": "0.850", "CC_STABILITY_MIN_PRECISION": "0.850", "CC_STABILITY_MAX_GAP": "85",
    "CC_STABILITY_MIN_TIMES": "3", "CC_GROUPING_MIN_IMAGE_THRESHOLD": "0.5", "CC_GROUPING_TEMPORAL_WINDOW": "5",
    "CC_GROUPING_MIN_RECALL": "0.5", "CC_GROUPING_MIN_TIME_F_MEASURE": "None", "CC_GROUPING_MIN_TIME_IOU": "None",
    "VIDEO_SEGMENTATION_METHOD": "3", "VIDEO_SEGMENTATION_DEL_EVENT_ADD_THRESHOLD": "10",
    "VIDEO_SEGMENTATION_DEL_EVENT_MIN_LENGTH": "15", "VIDEO_SEGMENTATION_DEL_EVENT_THRESHOLD": "0.25",
    "FCN_BINARIZER_BINARY_THRESHOLD": "128",
}


def _script(name):
    spec = importlib.util.spec_from_file_location("lm_pipeline_" + name.replace(".", "_"), os.path.join(_DROPIN, name))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class LecturePipeline:
    def __init__(self, width, height, conf=None, network=None, lib=None, verbose=False):
        if _DROPIN not in sys.path:
            sys.path.insert(0, _DROPIN)
        from AM_CommonTools.configuration.configuration import Configuration
        from AccessMath.preprocessing.content.cc_stability_estimator import CCStabilityEstimator
        from lecturemath_amd import _lib, device
        if lib is not None:
            _lib._default = lib
        self.lib = lib or _lib.load()
        self.be = device.Backend(self.lib)
        values = dict(DEFAULT_CONF)
        values.update({k: str(v) for k, v in (conf or {}).items()})
        self.configuration = Configuration(values)
        self.process = types.SimpleNamespace(configuration=self.configuration, params={}, database=None)
        self.width, self.height, self.network, self.verbose = int(width), int(height), network, verbose
        c = self.configuration
        self.estimator = CCStabilityEstimator(self.width, self.height, c.get_float("CC_STABILITY_MIN_RECALL"),
                                              c.get_float("CC_STABILITY_MIN_PRECISION"), c.get_int("CC_STABILITY_MAX_GAP"), verbose)
        self.frame_times, self.frame_indices = [], []

    # ---- step 01 + 02 ------------------------------------------------------------------------------------------------
    def _stamp(self, n, times, indices):
        base = len(self.frame_indices)
        self.frame_indices.extend(list(indices) if indices is not None else range(base, base + n))
        self.frame_times.extend(list(times) if times is not None else [float(i) for i in range(base, base + n)])

    def add_binary_frames(self, binary, times=None, indices=None):
        """uint8 [n, H, W], ink = 255 (the worker's inverted binary), numpy or device tensor."""
        frames = binary if not isinstance(binary, np.ndarray) else self.be.from_host(np.ascontiguousarray(binary, np.uint8))
        self.estimator.add_frames_device(frames)
        self._stamp(int(frames.shape[0]), times, indices)

    def add_rgb_frames(self, rgb, times=None, indices=None):
        """uint8 [n, H, W, 3]: FCN-LectureNet logits -> sigmoid * 255 >= threshold -> inverted binary (ink = 255), all on the device
        (FCN_lecturenet.py:452-467 + FCN_lecturenet_binarizer.py:54), then step 02."""
        if self.network is None:
            raise ValueError("LecturePipeline(network=...) is needed for RGB input")
        thr = self.configuration.get_int("FCN_BINARIZER_BINARY_THRESHOLD", 128)
        n = int(rgb.shape[0])
        if hasattr(self.network, "binarize_frames_device"):
            out = self.network.binarize_frames_device(rgb, thr)
        else:                                   # any object with forward_logits(rgb_u8 [H,W,3]) -> (logits, text, rec) on the device
            from lecturemath_amd import _lib
            out = self.be.empty((n, self.height, self.width), np.uint8)
            for i in range(n):
                logits, _, _ = self.network.forward_logits(rgb[i])
                dst = out[i] if not isinstance(out, np.ndarray) else out[i:i + 1]
                self.lib.check(self.lib.lm_threshold_invert(_lib.ptr(logits), _lib.ptr(dst), self.height * self.width, thr, self.be.stream()))
        self.estimator.add_frames_device(out)
        self._stamp(n, times, indices)

    # ---- steps 03, 04, 05 ----------------------------------------------------------------------------------------------
    def finish(self, reconstructed_png=False):
        from AccessMath.data.space_time_struct import SpaceTimeStruct
        est, cfg = self.estimator, self.configuration
        est.finish_processing()
        max_gap, min_times = cfg.get_int("CC_STABILITY_MAX_GAP", 85), cfg.get_int("CC_STABILITY_MIN_TIMES", 3)
        est.split_stable_cc_by_gaps(max_gap, min_times)
        stable = est.get_stable_cc_idxs(min_times)
        time_ov, _, all_ov = est.compute_overlapping_stable_cc(stable, cfg.get_int("CC_GROUPING_TEMPORAL_WINDOW", 5))
        groups, gid = est.compute_groups(stable, time_ov, cfg.get("CC_GROUPING_MIN_RECALL", 0.0), cfg.get("CC_GROUPING_MIN_TIME_F_MEASURE", 0.5),
                                         cfg.get("CC_GROUPING_MIN_TIME_IOU", 0.25))
        group_ages, groups_per_frame = est.compute_groups_temporal_information(groups)
        conflicts = est.compute_conflicting_groups(stable, all_ov, len(groups), gid)
        group_images, group_boundaries = est.compute_group_images(groups, group_ages, cfg.get_float("CC_GROUPING_MIN_IMAGE_THRESHOLD", 0.5))
        st3d = SpaceTimeStruct(self.frame_times, self.frame_indices, est.height, est.width, group_ages, group_images, group_boundaries)
        compressed = (est.frames_from_groups(groups, group_boundaries, groups_per_frame, group_ages, group_images, None, min_times, True)
                      if reconstructed_png else [])
        step03 = [(self.frame_times, self.frame_indices, compressed), (group_ages, conflicts), st3d]
        import contextlib
        import io
        sink = contextlib.nullcontext() if self.verbose else contextlib.redirect_stdout(io.StringIO())
        with sink:
            intervals = _script("pre_ST3D_v3.0_04_vid_segmentation.py").process_input(self.process, step03)
            (summary_indices, summary_times, keyframes), = _script("pre_ST3D_v3.0_05_generate_summary.py").process_input(self.process, [st3d, intervals])
        from AccessMath.preprocessing.content.keyframe_extractor import KeyframeExtractor   # noqa: F401  (cc_times below)
        return {"group_ages": group_ages, "conflicts": conflicts, "st3d": st3d, "intervals": intervals, "keyframes": keyframes,
                "summary_indices": summary_indices, "summary_times": summary_times, "reconstructed_png": compressed,
                "reconstructed_device": lambda first, count: est.frames_from_groups_device(first, count)}
