"""Build libdrt_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "drt_kernels.hip")      # gfx950 kernels + device-side C ABI
SRC_HOST = os.path.join(HERE, "csrc", "drt_remesh.cpp")  # host-only part of the C ABI (remeshing)
OUT = os.path.join(HERE, "libdrt_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def sources():
    d = os.path.join(HERE, "csrc")
    inc = os.path.join(os.path.dirname(HERE), "include")
    return [os.path.join(d, f) for f in sorted(os.listdir(d))] + [os.path.join(inc, f) for f in sorted(os.listdir(inc))]


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(f) <= t for f in sources())


def build(force=False, verbose=False):
    if not force and up_to_date():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-o", OUT, SRC, SRC_HOST]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force=True, verbose=True)
