#!/usr/bin/env python3
"""Soak run of the projection-pass fuzz tests of tests/test_gpu_raster.py (random intrinsics / poses, random image sizes / view counts) over many seeds.
usage (on the GPU box): python tools/fuzz_soak_raster.py <first_seed> <n_seeds>      (round 3: seeds 8-407, no failure)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_raster as tr
from conftest import IOR
from drt_amd import diffrender
diffrender.intIOR = IOR
diffrender.DENSE_FACE_IDS = True      # (what tests/conftest.py's autouse fixture sets under pytest: the tests compare last_face1 for EVERY ray)
first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0; t0 = time.time()
for seed in range(first, first + count):
    for fn in (tr.test_random_intrinsics_and_poses, tr.test_random_image_sizes_and_view_counts):
        try:
            fn(diffrender, seed)
        except Exception as e:
            bad += 1
            print("seed", seed, fn.__name__, "FAILED:", type(e).__name__, str(e).splitlines()[0][:200] if str(e) else "")
print(count, "seeds x 2,", bad, "failures,", int(time.time() - t0), "s")
