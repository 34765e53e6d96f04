#!/usr/bin/env python3
"""Soak run of the whole-path, silhouette / smoothness and closest-point fuzz tests of tests/test_gpu_fuzz.py over more seeds than the suite
holds.   usage (on the GPU box): python tools/fuzz_soak_path.py <first_seed> <n_seeds>      (round 3: seeds 20-139, no failure)"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as fz
from drt_amd import diffrender
diffrender.DENSE_FACE_IDS = True      # (what tests/conftest.py's autouse fixture sets under pytest)
first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0; t0 = time.time()
for seed in range(first, first + count):
    for fn in (fz.test_random_shapes_cameras_and_ior_through_the_whole_path, fz.test_random_shapes_through_the_silhouette_and_smoothness_terms, fz.test_random_soups_closest_point):
        try:
            fn(seed)
        except Exception as e:
            bad += 1
            print("seed", seed, fn.__name__, "FAILED:", str(e).splitlines()[0][:300])
print(count, "seeds x 3 tests,", bad, "failures,", int(time.time() - t0), "s")
