"""CPU: the C-ABI library loads and exports every symbol include/lecturemath_amd.h declares (no compute calls)."""
import ctypes
import os
import re

from lecturemath_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "lecturemath_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", src)
    reference_symbols = {"CC_AgeBoundaries", "speaker_detection_handle_frame", "regionCumulativeDistribution", "adapthisteq", "combine_results"}
    return sorted(set(n for n in names if n.startswith("lm_") or n in reference_symbols))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_lib.SIGNATURES)


def test_hip_library_exports_all():
    import __graft_entry__ as ge
    ge.build()
    assert os.path.exists(_lib.DEFAULT_PATH)
    # dlopen needs libamdhip64 (present in this image); no device is touched by loading
    lib = ctypes.CDLL(_lib.DEFAULT_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert lib.lm_is_device_build() == 1
    assert lib.lm_abi_version() == 1


def test_missing_library_fails_loudly(tmp_path):
    import pytest
    with pytest.raises(_lib.LecturemathLibraryError):
        _lib.Library(str(tmp_path / "nope.so"))
