"""ctypes binding of the C ABI declared in include/lecturemath_amd.h.

The product path is HIP only: `load()` opens lecturemath_amd/liblecturemath_hip.so (built by
`__graft_entry__.build()` / `make -C lecturemath_amd/csrc`) and raises LecturemathLibraryError when it
is missing -- there is no CPU fallback.  Tests may hand an explicit `path` (the fiber-emulated build
of the same sources under tests/hipemu) to exercise kernel logic in the GPU-less container.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_PATH = os.path.join(_HERE, "liblecturemath_hip.so")

LM_OK, LM_ERR_ARG, LM_ERR_HIP, LM_ERR_CAPACITY, LM_ERR_STATE = 0, 1, 2, 3, 4


class LecturemathLibraryError(RuntimeError):
    pass


class LecturemathError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("lecturemath_amd error %d: %s" % (code, msg))
        self.code = code


_vp = ctypes.c_void_p
_i32 = ctypes.c_int32
_i64 = ctypes.c_int64
_f64 = ctypes.c_double

# name -> (restype, argtypes); must list every symbol include/lecturemath_amd.h declares
SIGNATURES = {
    "lm_abi_version": (ctypes.c_int, []),
    "lm_last_error": (ctypes.c_char_p, []),
    "lm_is_device_build": (ctypes.c_int, []),
    "CC_AgeBoundaries": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "lm_ctx_create": (_vp, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "lm_ctx_destroy": (None, [_vp]),
    "lm_threshold_invert": (ctypes.c_int, [_vp, _vp, _i64, ctypes.c_int, _vp]),
    "lm_threshold": (ctypes.c_int, [_vp, _vp, _i64, ctypes.c_int, ctypes.c_int, _vp]),
    "lm_label_batch": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp]),
    "lm_label_counts": (ctypes.c_int, [_vp, _vp, _vp]),
    "lm_cc_stats_batch": (ctypes.c_int, [_vp, _vp]),
    "lm_cc_stats_read": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "lm_ctx_set_profiling": (ctypes.c_int, [_vp, ctypes.c_int]),
    "lm_ctx_profile_read": (ctypes.c_int, [_vp, _vp, _vp, _vp]),
    "lm_label_host": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp]),
    "lm_stream_create": (_vp, [_vp, ctypes.c_int, _i64, _i64, ctypes.c_int, _f64, _f64, ctypes.c_int, ctypes.c_int]),
    "lm_stream_destroy": (None, [_vp]),
    "lm_stream_reset": (ctypes.c_int, [_vp, _vp]),
    "lm_stream_set_min_pixels": (ctypes.c_int, [_vp, ctypes.c_int]),
    "lm_stream_push": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp]),
    "lm_stream_push_records": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp]),
    "lm_stream_match": (ctypes.c_int, [_vp, ctypes.c_int, _vp]),
    "lm_stream_run_logits": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp]),
    "lm_stream_push_labelled": (ctypes.c_int, [_vp, ctypes.c_int, _vp]),
    "lm_stream_pack_size": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, _vp]),
    "lm_stream_pack": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, _i64, _vp]),
    "lm_stream_append_packed": (ctypes.c_int, [_vp, _vp, _i64, _vp]),
    "lm_stream_match_stats": (ctypes.c_int, [_vp, _vp, _vp]),
    "lm_stream_assign_bytes": (ctypes.c_int, [_vp, _vp, _vp]),
    "lm_stream_export_assign": (ctypes.c_int, [_vp, _vp, _i64, _vp]),
    "lm_stream_import_assign": (ctypes.c_int, [_vp, _vp, _i64, _vp]),
    "lm_frame_sums": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int64, _vp, _vp]),
    "speaker_detection_handle_frame": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp]),
    "regionCumulativeDistribution": (None, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, _vp]),
    "adapthisteq": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, _vp]),
    "combine_results": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_ubyte, _vp]),
    "lm_image_pairs_overlap": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, _vp, ctypes.c_int64, _vp, _vp]),
    "lm_stream_counters": (ctypes.c_int, [_vp, _vp, _vp]),
    "lm_stream_import": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int, _i64, _i64, ctypes.c_int, _i64, _vp, _vp, _vp,
                                        ctypes.c_int, ctypes.c_int, _vp]),
    "lm_stream_read": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "lm_group_run": (_vp, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f64, _f64, ctypes.c_int, _vp]),
    "lm_group_destroy": (None, [_vp]),
    "lm_group_render": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, _vp]),
    "lm_group_array": (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp]),
    "lm_label_batch_logits": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp]),
    "lm_label_was_fused": (ctypes.c_int, [_vp]),
    "lm_fcn_create": (_vp, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "lm_fcn_destroy": (None, [_vp]),
    "lm_fcn_set_layer": (ctypes.c_int, [_vp, ctypes.c_int, _vp, _i64, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int]),
    "lm_fcn_set_layer_terms": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    "lm_fcn_forward": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp]),
    "lm_resample_rgb8": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_int, _vp, _vp,
                                        ctypes.c_int, _vp]),
    "lm_upsample_nearest_u8": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int, _vp]),
    "lm_fcn2_create": (_vp, [_vp, _vp, ctypes.c_int, ctypes.c_int]),
    "lm_fcn2_destroy": (None, [_vp]),
    "lm_fcn2_set_layer": (ctypes.c_int, [_vp, ctypes.c_int, _vp, ctypes.c_int, _vp, _i64, ctypes.c_int, _vp, ctypes.c_int]),
    "lm_fcn2_forward": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp]),
}


def ptr(x):
    """Address of a torch tensor / numpy array / raw int (None -> NULL)."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    if hasattr(x, "ctypes"):
        return x.ctypes.data
    raise TypeError("cannot take the address of %r" % type(x))


class Library:
    def __init__(self, path=None):
        self.path = path or DEFAULT_PATH
        if not os.path.exists(self.path):
            raise LecturemathLibraryError(
                "HIP library %s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % self.path)
        try:
            self.cdll = ctypes.CDLL(self.path)
        except OSError as e:   # e.g. libamdhip64 missing
            raise LecturemathLibraryError("cannot load %s: %s" % (self.path, e)) from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(self.cdll, name)
            except AttributeError as e:
                raise LecturemathLibraryError("%s does not export %s" % (self.path, name)) from e
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)
        self.is_device_build = bool(self.lm_is_device_build())

    def last_error(self):
        m = self.lm_last_error()
        return m.decode() if m else ""

    def check(self, rc):
        if rc != LM_OK:
            raise LecturemathError(rc, self.last_error())


_default = None


def load(path=None):
    """The process-wide library (HIP build unless an explicit path is given)."""
    global _default
    if path is not None:
        return Library(path)
    if _default is None:
        _default = Library()
    return _default
