#!/usr/bin/env python3
"""Soak run of tests/test_gpu_remesh.py::test_device_remesh_on_random_shapes over seeds the suite does not hold (it runs 0-5): random closed
shapes coarsened and refined by the device remesher, every invariant of the test checked, plus determinism (two runs, same bits).
usage (on the GPU box): python tools/fuzz_soak_remesh.py [first_seed] [n_seeds]"""
import os, sys, time, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_remesh as tr
from test_gpu_fuzz import _shape

first = int(sys.argv[1]) if len(sys.argv) > 1 else 6
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0; t0 = time.time()
fn = getattr(tr.test_device_remesh_on_random_shapes, "__wrapped__", tr.test_device_remesh_on_random_shapes)
for seed in range(first, first + count):
    try:
        fn(seed)
        rng = np.random.default_rng(100 + seed)
        mesh = _shape(rng, noises=(0.0, 0.03, 0.06))
        L = 0.8 * float(tr._edge_len(mesh).mean())
        a, _ = tr._gpu_remesh(mesh, L); b, _ = tr._gpu_remesh(mesh, L)
        assert np.array_equal(a.vertices, b.vertices) and np.array_equal(a.faces, b.faces), "not deterministic"
    except Exception:
        bad += 1
        print(f"seed {seed}: FAILED\n{traceback.format_exc(limit=3)}", flush=True)
print(f"{count} seeds x 2 target lengths + determinism, {bad} failures, {time.time() - t0:.0f} s")
