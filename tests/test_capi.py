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


def test_optix_extension_loads_with_the_reference_call(tmp_path):
    """Boundary B1 as the reference binds it (DiffRender.py:3-6): torch.utils.cpp_extension.load(name="optix", sources=[...])
    of drt_amd/csrc/optix_hip.cpp gives a module with the pybind11 class of optix_extend.cpp:77-83; and the in-tree build
    (drt_amd/optix.so) is the same class.  Host-only C++, so it builds and imports without a GPU; constructing a tracer
    without one raises."""
    from torch.utils.cpp_extension import load
    build.build()
    inc, cflags, ldflags = build.optix_ext_flags()
    optix = load(name="optix", sources=[build.OPTIX_SRC], extra_include_paths=inc, extra_cflags=cflags, extra_ldflags=ldflags,
                 build_directory=str(tmp_path))
    members = {m for m in dir(optix.optix_mesh) if not m.startswith("_")}
    assert members == {"update_mesh", "update_vert", "intersect"}
    build.build_optix_ext()
    import drt_amd.optix as intree
    assert {m for m in dir(intree.optix_mesh) if not m.startswith("_")} == members
    if not torch.cuda.is_available():
        for mod in (optix, intree):
            with pytest.raises(RuntimeError, match="libdrt_hip error"):
                mod.optix_mesh(0)
