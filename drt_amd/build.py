"""Build libdrt_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# gfx950 kernels + the C ABI, one translation unit per concern; drt_remesh.cpp is the host-only part (remeshing)
UNITS = ["drt_api.hip", "drt_build.hip", "drt_trace.hip", "drt_pipeline.hip", "drt_raster.hip", "drt_edges.hip", "drt_topology.hip", "drt_remesh_gpu.hip", "drt_remesh.cpp"]
OUT = os.path.join(HERE, "libdrt_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]


def sources():
    inc = os.path.join(os.path.dirname(HERE), "include")
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + [os.path.join(inc, f) for f in sorted(os.listdir(inc))]


def source_hash():
    """sha256 over the kernel sources and the C-ABI header (names + contents, sorted): what profiles/pmc.json is stamped with
    (tools/make_pmc_json.py) and what bench.py recomputes before it prices live launch times against those counters."""
    import hashlib
    h = hashlib.sha256()
    for f in sources():
        if f.endswith((".hip", ".h", ".cpp", ".map")):
            h.update(os.path.basename(f).encode() + b"\0")
            h.update(open(f, "rb").read())
    return h.hexdigest()


def up_to_date(out=OUT):
    if not os.path.exists(out):
        return False
    t = os.path.getmtime(out)
    return all(os.path.getmtime(f) <= t for f in sources())


def build(force=False, verbose=False, out=OUT, extra_flags=()):
    """Compile every unit (in parallel) and link them into ``out``; ``extra_flags`` e.g. ("-DDRT_STACK_FAST=3",)."""
    if not force and out == OUT and not extra_flags and up_to_date():
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory(prefix="drt_build_") as tmp:
        def compile_one(unit):
            obj = os.path.join(tmp, unit.rsplit(".", 1)[0] + ".o")
            cmd = [hipcc] + FLAGS + list(extra_flags) + ["-c", "-o", obj, os.path.join(CSRC, unit)]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            return obj
        with ThreadPoolExecutor(max_workers=len(UNITS)) as pool:
            objs = list(pool.map(compile_one, UNITS))
        link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"), "-o", out] + objs
        if verbose:
            print(" ".join(link), flush=True)
        subprocess.check_call(link)
    return out


OPTIX_SRC = os.path.join(CSRC, "optix_hip.cpp")
OPTIX_OUT = os.path.join(HERE, "optix.so")


def optix_ext_flags():
    """(include dirs, compile flags, link flags) of the pybind11 tracer class -- the same ones the reference-style
    ``torch.utils.cpp_extension.load(name="optix", sources=[optix_hip.cpp], extra_include_paths=..., extra_cflags=...,
    extra_ldflags=...)`` call takes."""
    inc = [os.path.join(os.path.dirname(HERE), "include"), "/opt/rocm/include"]
    cflags = ["-D__HIP_PLATFORM_AMD__=1"]
    ldflags = ["-L" + HERE, "-ldrt_hip", "-Wl,-rpath," + HERE, "-lc10_hip"]
    return inc, cflags, ldflags


def build_optix_ext(force=False, verbose=False):
    """g++ the torch extension in-tree (drt_amd/optix.so: ``import drt_amd.optix``); host-only C++ over the C ABI."""
    deps = [OPTIX_SRC, OUT, os.path.join(os.path.dirname(HERE), "include", "drt_hip.h")]
    if not force and os.path.exists(OPTIX_OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OPTIX_OUT) for d in deps):
        return OPTIX_OUT
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    inc, cflags, ldflags = optix_ext_flags()
    inc = inc + ce.include_paths() + [sysconfig.get_paths()["include"]]
    tlib = ce.library_paths()[0]
    cmd = (["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DTORCH_EXTENSION_NAME=optix", "-DTORCH_API_INCLUDE_EXTENSION_H",
            f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"] + cflags + ["-isystem" + i for i in inc] +
           ["-o", OPTIX_OUT, OPTIX_SRC, "-L" + tlib, "-Wl,-rpath," + tlib, "-lc10", "-ltorch_cpu", "-ltorch", "-ltorch_python"] +
           [f for f in ldflags if not f.startswith("-Wl,-rpath")] + ["-Wl,-rpath,$ORIGIN"])      # in-tree: libdrt_hip.so sits next to it
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OPTIX_OUT


if __name__ == "__main__":
    build(force=True, verbose=True)
    build_optix_ext(force=True, verbose=True)
