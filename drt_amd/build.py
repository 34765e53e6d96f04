"""Build libdrt_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# gfx950 kernels + the C ABI, one translation unit per concern; drt_remesh.cpp is the host-only part (remeshing)
UNITS = ["drt_api.hip", "drt_build.hip", "drt_trace.hip", "drt_pipeline.hip", "drt_edges.hip", "drt_remesh.cpp"]
OUT = os.path.join(HERE, "libdrt_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]


def sources():
    inc = os.path.join(os.path.dirname(HERE), "include")
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + [os.path.join(inc, f) for f in sorted(os.listdir(inc))]


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


if __name__ == "__main__":
    build(force=True, verbose=True)
