#!/usr/bin/env python3
"""Soak run of the tracer fuzz of tests/test_gpu_fuzz.py over many more seeds than the test suite holds (the suite runs seeds 0-15).
usage (on the GPU box): python tools/fuzz_soak.py [first_seed] [n_seeds] [offset]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as fz
from oracle import diffrender_oracle as orc
from drt_amd.optix_mesh import optix_mesh

first = int(sys.argv[1]) if len(sys.argv) > 1 else 16
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
offset = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0          # the scene moved away from the origin by this many of its own sizes
bad = 0; rays_total = 0; t0 = time.time()
for seed in range(first, first + count):
    V, F, rays, scale = fz.scene(seed)
    if offset:
        sh = (np.array([1.0, -0.7, 0.4]) * offset * scale).astype(np.float32)
        V = (V + sh).astype(np.float32); rays = rays.copy(); rays[:, :3] += sh
    t = optix_mesh(0)
    t.update_mesh(torch.tensor(F, device="cuda"), torch.tensor(V, device="cuda"))
    R = torch.tensor(rays, device="cuda")
    T, ID = t.intersect(R); Tb, IDb = t.intersect_bruteforce(R); hit = t.intersect_any(R)
    To, IDo = orc.trace_closest(F, V, rays, bvh=False)
    ok = torch.equal(ID, IDb) and torch.equal(T, Tb) and np.array_equal(ID.cpu().numpy(), IDo) and np.array_equal(T.cpu().numpy(), To) and torch.equal(hit.bool(), ID >= 0) and t.check()[0] == 0
    rays_total += len(rays)
    if not ok:
        bad += 1
        d = int(((ID.cpu().numpy() != IDo) | (T.cpu().numpy() != To)).sum()); d2 = int((~torch.eq(ID, IDb)).sum())
        print(f"seed {seed}: MISMATCH tree vs oracle {d}, tree vs exhaustive {d2} (tris {len(F)}, scale {scale:.3g})")
print(f"{count} scenes, {rays_total} rays, {bad} scenes with a mismatch, {time.time() - t0:.0f} s")
import ctypes
from drt_amd import _lib
out = (ctypes.c_int64 * 4)()
_lib.check(_lib.lib().drt_check_violations(out))
if out[0] >= 0:                                # a -DDRT_CHECK=1 build (DRT_HIP_LIB): its LDS-stack violation counters
    print("checked build: stores above the rows / pops of an empty stack / guard rows overwritten / illegal stack at a visit =", list(out))
    bad += sum(out)
sys.exit(1 if bad else 0)
