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
