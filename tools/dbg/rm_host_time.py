"""Where the HOST spends a device remesh call: time inside each drt_rm_* call (enqueue only, no synchronize), inside .item() / .tolist()
(the round trips), and the rest (torch allocations, fills, Python).
usage (via gpurun): python tools/dbg/rm_host_time.py"""
import os, sys, time, collections
sys.path.insert(0, os.getcwd())
import torch
from drt_amd import diffrender as Render, mesh_io, remesh_gpu as RG, _lib
lib = _lib.lib()
T = collections.defaultdict(float); N = collections.defaultdict(int)
class Wrap:
    def __init__(self, lib): self._lib = lib
    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith("drt_rm_") and name != "drt_closest_point": return fn
        def call(*a):
            t0 = time.perf_counter(); rc = fn(*a); T[name] += time.perf_counter() - t0; N[name] += 1
            return rc
        return call
_lib._lib = Wrap(lib)
for meth in ("item", "tolist"):
    orig = getattr(torch.Tensor, meth)
    def make(orig, meth):
        def f(self, *a, **k):
            t0 = time.perf_counter(); r = orig(self, *a, **k); T["sync:" + meth] += time.perf_counter() - t0; N["sync:" + meth] += 1
            return r
        return f
    setattr(torch.Tensor, meth, make(orig, meth))
mesh = mesh_io.subdivide_midpoint(mesh_io.read_ply("data/horse_vh.ply"))
scene = Render.Scene(mesh, 0)
L = scene.mean_len * 0.9
V, F = scene.vertices.detach(), scene.faces
for rep in range(3):
    T.clear(); N.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    RG.isotropic_remesh_gpu(V, F, L, surface=scene.optix_mesh)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    tot = sum(T.values())
    print(f"call {1e3 * dt:.1f} ms; inside wrapped calls {1e3 * tot:.1f} ms, elsewhere on the host {1e3 * (dt - tot):.1f} ms")
    for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
        print(f"   {k:32s} {N[k]:4d} calls {1e3 * v:7.2f} ms  ({1e6 * v / N[k]:6.1f} us each)")
