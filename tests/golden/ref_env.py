"""Imports the reference hot path in THIS container only (golden generation / oracle pinning).

/root/reference never travels to the GPU box; nothing under tests/ imports this module except
make_golden.py and the container-only oracle-vs-reference tests, all of which skip when
/root/reference is absent.
"""
import os
import subprocess
import sys
import tempfile

REF_ROOT = "/root/reference/ACCESS2021_release"
HERE = os.path.dirname(os.path.abspath(__file__))


def available():
    return os.path.isdir(REF_ROOT)


_scratch = None


def enter():
    """chdir into a scratch dir holding a gcc build of the reference C lib, and put the reference +
    shims on sys.path.  Returns the scratch dir."""
    global _scratch
    if _scratch is not None:
        return _scratch
    assert available()
    _scratch = tempfile.mkdtemp(prefix="lm_ref_")
    subprocess.check_call(["gcc", "-m64", "-shared", "-fPIC", "-O2",
                           os.path.join(REF_ROOT, "accessmath_lib.c"),
                           "-o", os.path.join(_scratch, "accessmath_lib.so"), "-lm"])
    sys.dont_write_bytecode = True
    sys.path.insert(0, os.path.join(HERE, "_ref_shims"))
    sys.path.insert(0, REF_ROOT)
    os.chdir(_scratch)   # the reference does CDLL('./accessmath_lib.so') at import time
    import warnings
    warnings.filterwarnings("ignore", category=DeprecationWarning)
    return _scratch
