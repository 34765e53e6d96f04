"""pytest configuration: `gpu` marker, repo on sys.path, shared helpers."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "data")
IOR = 1.4723


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests are skipped (not failed) on a box without a GPU; `-m gpu` on the GPU box runs them."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs an MI355X (no GPU visible)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _dense_face_ids():
    """The parity tests compare Scene.last_face1 / last_face2 for EVERY ray: ask the library to define them everywhere."""
    from drt_amd import diffrender
    old = diffrender.DENSE_FACE_IDS
    diffrender.DENSE_FACE_IDS = True
    yield
    diffrender.DENSE_FACE_IDS = old


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def data_path(name):
    return os.path.join(DATA, name)


_P = ctypes.c_void_p
_I64 = ctypes.c_int64
_D = ctypes.c_double


@pytest.fixture(scope="session")
def hostsim():
    """tests/hostsim: the device math headers compiled for the host with g++ (test-only)."""
    src = os.path.join(ROOT, "tests", "hostsim", "hostsim.cpp")
    out_dir = os.path.join(ROOT, "tests", "hostsim", "_build")
    so = os.path.join(out_dir, "libhostsim.so")
    os.makedirs(out_dir, exist_ok=True)
    deps = [src] + [os.path.join(ROOT, "drt_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "drt_amd", "csrc")) if f.endswith(".h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", so, src])
    hs = ctypes.CDLL(so)
    hs.hs_create.restype = _P
    hs.hs_create.argtypes = [_P, _I64, _P, _I64]
    hs.hs_destroy.argtypes = [_P]
    hs.hs_height.argtypes = [_P]
    hs.hs_sorted_faces.argtypes = [_P, _P]
    hs.hs_check.restype = _I64
    hs.hs_check.argtypes = [_P]
    hs.hs_intersect.argtypes = [_P, _P, _I64, _P, _P, ctypes.c_int, _P]
    hs.hs_closest_point.argtypes = [_P, _P, _I64, _P, _P, _P]
    hs.hs_within_distance.argtypes = [_P, _P, _P, _I64, _P]
    hs.hs_closest_near.argtypes = [_P, _P, _P, _I64, _P, _P, _P]
    hs.hs_render_forward.argtypes = [_P, _P, _P, _P, _I64, _D, _D, _P, _P, _P, _P, _P]
    hs.hs_render_backward.argtypes = [_P, _P, _P, _P, _I64, _D, _D, _P, _P, _P, _P, _P]
    hs.hs_ray_loss.argtypes = [_P, _P, _P, _P, _P, _I64, _P, _P]
    hs.hs_fused.argtypes = [_P, _P, _P, _P, _P, _P, _I64, _D, _D, _P, _P, _P]
    hs.hs_bounce.argtypes = [_P, _P, _P, _I64, _D, _D, _P, _P, _P, _P, _P, _P, _P, _P, _P]
    hs.hs_dihedral.argtypes = [_P, _P, _I64, ctypes.c_int, _P, _P, _P, _P]
    hs.hs_silhouette_flags.argtypes = [_P, _P, _I64, _P, _P]
    hs.hs_edge_sample_forward.argtypes = [_P, _P, _P, _I64, _P, _P, _P, _P]
    hs.hs_edge_sample_backward.argtypes = [_P, _P, _I64, _P, _P, _P, ctypes.c_int, _P]
    hs.hs_fit_view.argtypes = [_P, _P, ctypes.c_int, ctypes.c_int, _P]
    hs.hs_verify_rays.argtypes = [_P, _P, _P, ctypes.c_int, ctypes.c_int, _P]
    hs.hs_morton_plan.argtypes = [_P, _P, _P, _P]
    hs.hs_morton_key.restype = ctypes.c_uint32
    hs.hs_morton_key.argtypes = [_P, _P]
    hs.hs_raster.restype = _I64
    hs.hs_raster.argtypes = [_P, _P, _P, _P, ctypes.c_int, ctypes.c_int, _P, _P]
    hs.hs_fx_from.argtypes = [_D, _P, _P, _P]
    hs.hs_fx_to.restype = _D
    hs.hs_fx_to.argtypes = [ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint32]
    hs.hs_fx_sum.restype = _D
    hs.hs_fx_sum.argtypes = [_P, _I64]
    return hs


def fixture_view(g):
    """(origin, ray_dir, screen_pixel, valid, camera_M numpy tuple) of a render fixture, on the CPU."""
    import torch
    from drt_amd import views
    res = int(g["res"])
    o, d = views.generate_ray(res, res, g["Kinv"], g["Rinv"])
    rng = np.random.default_rng(int(g["target_seed"]))
    P = res * res
    center = fixture_frame(g)[0]
    sp = rng.standard_normal((P, 3)) * 40.0 + np.asarray(center) + np.array([0.0, 0.0, 150.0])
    valid = rng.random(P) > 0.1
    return o, d, torch.tensor(sp), torch.tensor(valid)


_frame = {}
_mesh = {}
HEADLINE_FIXTURE = "horse50k_r256_v11"      # the reference's own Python on BASELINE.json's ~50k-triangle mesh (make_golden.py horse)
BIG_FIXTURES = [HEADLINE_FIXTURE, "mouse37k_r256_v29"]     # + BASELINE.json configs[2]'s hull, mouse_vh.ply x4 (make_golden.py mouse)


def _fixture_key(g):
    if "mesh_sha256" not in g.files:
        return "hand"
    return str(g["hull"]) if "hull" in g.files else "horse"


def fixture_mesh(g):
    """The mesh a render fixture was made on: hand_vh.ply, or -- fixtures that carry `mesh_sha256` -- <hull>_vh.ply after one midpoint
    subdivision (horse: 50 248 triangles, mouse: 36 984), rebuilt here exactly as make_golden.py built it and checked against the
    fixture's hash."""
    from drt_amd import mesh_io
    key = _fixture_key(g)
    if key not in _mesh:
        if key == "hand":
            _mesh[key] = mesh_io.read_ply(data_path("hand_vh.ply"))
        else:
            import hashlib
            import tempfile
            hull = mesh_io.subdivide_midpoint(mesh_io.read_ply(data_path(f"{key}_vh.ply")))
            with tempfile.TemporaryDirectory() as tmp:
                f = os.path.join(tmp, f"{key}_x4.ply")
                mesh_io.write_ply(f, hull.vertices, hull.faces)
                m = mesh_io.read_ply(f)
            sha = hashlib.sha256(np.ascontiguousarray(m.vertices, np.float64).tobytes() + np.ascontiguousarray(m.faces, np.int64).tobytes()).hexdigest()
            assert sha == str(g["mesh_sha256"]) and len(m.faces) == int(g["n_faces"]) > 30000
            _mesh[key] = m
    return _mesh[key]


def fixture_frame(g):
    from drt_amd import views
    key = _fixture_key(g)
    if key not in _frame:
        _frame[key] = views.mesh_frame(fixture_mesh(g).vertices)
    return _frame[key]


def mesh_frame_hand():
    if "hand" not in _frame:
        from drt_amd import mesh_io, views
        _frame["hand"] = views.mesh_frame(mesh_io.read_ply(data_path("hand_vh.ply")).vertices)
    return _frame["hand"]
