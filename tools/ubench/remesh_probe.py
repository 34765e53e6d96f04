"""Where a device remesh call spends its time: per step (split / collapse / flip / smooth / project), rounds, host syncs.
usage (via gpurun): python tools/ubench/remesh_probe.py [target_len_factor]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from drt_amd import diffrender as Render, mesh_io, remesh_gpu as RG

factor = float(sys.argv[1]) if len(sys.argv) > 1 else 0.9
mesh = mesh_io.subdivide_midpoint(mesh_io.read_ply("data/horse_vh.ply"))
scene = Render.Scene(mesh, 0)
L = scene.mean_len * factor
V, F = scene.vertices.detach(), scene.faces
import hashlib
def run():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    V2, F2, stats = RG.isotropic_remesh_gpu(V, F, L, surface=scene.optix_mesh, return_stats=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    digest = hashlib.sha256(V2.cpu().numpy().tobytes() + F2.cpu().numpy().tobytes()).hexdigest()[:12]
    return dt, len(F2), stats, digest
# 1. the call as the optimisation loop makes it (no instrumentation: the rounds of a batch overlap their enqueueing)
for rep in range(5):
    dt, nf2, stats, digest = run()
    print(f"remesh {len(F)} -> {nf2} faces, target {L:.3f}: {1e3 * dt:.1f} ms  {stats}  mesh sha256 {digest}")
# 2. where the time goes: every step bracketed by synchronize (which serialises host and device: the total of this run is larger)
T = {}
def timed(name, fn):
    def wrap(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); T[name] = T.get(name, 0.0) + time.perf_counter() - t0
        return r
    return wrap
for name in ("split_long_edges", "collapse_short_edges", "flip_edges", "smooth_tangential", "project_to_surface", "compact"):
    setattr(RG._Work, name, timed(name, getattr(RG._Work, name)))
dt, nf2, stats, digest = run()
print(f"with every step bracketed by synchronize: {1e3 * dt:.1f} ms   " + "  ".join(f"{k} {1e3 * v:.1f}" for k, v in sorted(T.items(), key=lambda kv: -kv[1])) + " (ms)")
