"""Last file of the GPU suite (alphabetical order): the stack-invariant counters of a -DDRT_CHECK=1 build of the library must be zero
after everything that ran before it in this process (see test_gpu_parity.py::test_lds_stack_invariants_hold_in_a_checked_build, which
runs a selection of the suite against such a build in a subprocess).  Against a normal build the counters read -1 and the test is
skipped."""
import ctypes
import os

import pytest

pytestmark = pytest.mark.gpu


def test_zz_check_counters_are_zero():
    from drt_amd import _lib
    out = (ctypes.c_int64 * 4)()
    _lib.check(_lib.lib().drt_check_violations(out))
    vals = list(out)
    if vals[0] < 0:
        assert not os.environ.get("DRT_EXPECT_CHECKED"), "DRT_EXPECT_CHECKED is set but the loaded library was built without -DDRT_CHECK=1"
        pytest.skip("library built without -DDRT_CHECK=1")
    names = ("stores above the lane's rows", "pops of an empty stack", "overwritten guard rows", "visits that started with an illegal stack")
    assert vals == [0, 0, 0, 0], dict(zip(names, vals))
