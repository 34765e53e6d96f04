"""The C-ABI library: builds, loads, exports every symbol include/*.h declares, and fails
cleanly (no crash, no fallback) when there is no GPU.  No compute calls here."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT
from drt_amd import _lib, build


def declared_symbols():
    names = set()
    inc = os.path.join(ROOT, "include")
    for f in os.listdir(inc):
        text = open(os.path.join(inc, f)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(drt_[a-z0-9_]+)\s*\(", text))
    return names


def test_library_builds_and_exports_every_declared_symbol():
    build.build()
    h = ctypes.CDLL(_lib.LIB_PATH)
    decl = declared_symbols()
    assert len(decl) >= 15
    for name in decl:
        assert hasattr(h, name), f"{name} declared in include/ but not exported"
    assert decl == set(_lib.SIGNATURES), "drt_amd/_lib.py must bind exactly the declared C ABI"
    assert _lib.lib().drt_version() >= 1


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_gpu_means_loud_failure_not_fallback():
    h = ctypes.c_void_p()
    rc = _lib.lib().drt_create(0, ctypes.byref(h))
    assert rc != 0 and not h.value
    assert _lib.lib().drt_last_error()
    from drt_amd.optix_mesh import optix_mesh
    with pytest.raises(RuntimeError):
        optix_mesh(0)
    from drt_amd import diffrender
    with pytest.raises(RuntimeError):
        diffrender.Scene(os.path.join(ROOT, "data", "hand_vh.ply"))


def test_null_arguments_are_rejected():
    lib = _lib.lib()
    assert lib.drt_create(0, None) != 0
    assert lib.drt_update_vert(None, None, 0, None) != 0
    assert lib.drt_intersect(None, None, 0, None, None, None) != 0
    assert lib.drt_ray_loss(None, None, None, None, None, 5, None, None, None, None, None) != 0
    lib.drt_destroy(None)   # no-op
