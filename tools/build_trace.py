#!/usr/bin/env python3
"""Build-only loop (update_verticex = full LBVH rebuild) for rocprofv3 --kernel-trace: per-kernel times of the build alone.
usage (via gpurun): rocprofv3 --kernel-trace --stats -d gpurun_out/bt -o bt -- python tools/build_trace.py [mesh] [subdiv] [iters]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from drt_amd import diffrender as Render, mesh_io  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "horse"
sub = int(sys.argv[2]) if len(sys.argv) > 2 else 1
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 30
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
m = mesh_io.read_ply(os.path.join(root, "data", f"{name}_vh.ply"))
for _ in range(sub):
    m = mesh_io.subdivide_midpoint(m)
scene = Render.Scene(m)
v = scene.vertices.detach().clone()
for _ in range(5):
    scene.update_verticex(v)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    scene.update_verticex(v)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f"{name} x{4 ** sub}: {scene.faces.shape[0]} triangles, update_verticex + sync {dt * 1e3:.3f} ms")
