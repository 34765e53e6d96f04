"""The fuzz scenes of tests/test_gpu_fuzz.py without a GPU: the traversal headers compiled for the host (tests/hostsim) and the oracle's own
BVH tracer (oracle/bvh_tracer.c, the CPU baseline of bench.py) against the oracle's exhaustive test -- T and ID bit for bit.  What a GPU run
checks on the kernels is checked here on the same source lines (drt_tri.h, drt_traverse.h, drt_lbvh.h) as g++ compiles them."""
import numpy as np
import pytest

from oracle import diffrender_oracle as orc
from test_gpu_fuzz import scene
from test_hostsim import HostScene


@pytest.mark.parametrize("seed", range(16))
def test_host_traversal_and_oracle_bvh_equal_the_exhaustive_test(hostsim, seed):
    V, F, rays, _ = scene(seed, n_rays=300)
    To, IDo = orc.trace_closest(F, V, rays, bvh=False)
    if len(F) >= 2:                                     # (the host scene builder wants a tree)
        s = HostScene(hostsim, F, V.astype(np.float64))
        T, ID, _ = s.intersect(rays)
        assert np.array_equal(ID, IDo) and np.array_equal(T, To)
        Ta, IDa, _ = s.intersect(rays, any_hit=True)
        assert np.array_equal(IDa >= 0, IDo >= 0)
    Tb, IDb = orc.trace_closest(F, V, rays, bvh=True)
    assert np.array_equal(IDb, IDo) and np.array_equal(Tb, To)
