#!/usr/bin/env python3
"""EXPERIMENT (round-2 review item 4c): node visits per REFRACTED ray on the product's Morton tree collapsed to 4-wide nodes (what
k_trace walks) and to 8-wide nodes (a BVH8), counted on the host by tools/bvhq/bvhq.cpp::bvhq_visits_wide.  The rays are the ones
the second traversal of a step sees: camera rays of four turntable views refracted into the object at their first hit (flat
normals, IOR 1.4723, the 1e-5 offset).  Prints inner / leaf visits per ray, the longest ray and the box tests per ray; with the
measured instruction counts of a visit (103 per 4-wide inner visit of which 24 per child, 72 per leaf visit)
that prices the 8-wide variant without building it.   usage: python tools/bvhq/bvh_width.py [mesh] [subdiv] [res]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from drt_amd import mesh_io, views

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libbvhq.so"))
P = ctypes.c_void_p
lib.bvhq_visits_wide.restype = None
lib.bvhq_visits_wide.argtypes = [P, ctypes.c_int64, P, ctypes.c_int64, ctypes.c_int, P, ctypes.c_int64, P, P]
IOR = 1.4723


def wide(F, V, W, rays):
    ids = np.empty(len(rays), np.int32); st = np.zeros(6)
    lib.bvhq_visits_wide(F.ctypes.data, len(F), V.ctypes.data, len(V), W, rays.ctypes.data, len(rays), ids.ctypes.data, st.ctypes.data)
    return ids, st


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "horse"
    sub = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    res = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    m = mesh_io.read_ply(os.path.join(ROOT, "data", f"{name}_vh.ply"))
    for _ in range(sub):
        m = mesh_io.subdivide_midpoint(m)
    F = np.ascontiguousarray(m.faces, np.int32); V = np.ascontiguousarray(m.vertices, np.float32)
    c, e = views.mesh_frame(m.vertices)
    rays = []
    for k in (3, 20, 41, 60):
        R, K, Ri, Ki = views.turntable_cameras(c, e, 72, res, res)[k]
        o, d = views.generate_ray(res, res, Ki, Ri)
        rays.append(np.concatenate([o.numpy(), d.numpy()], 1))
    rays = np.concatenate(rays)                                  # float64 camera rays
    r32 = np.ascontiguousarray(rays.astype(np.float32))
    ids, st4 = wide(F, V, 4, r32)
    hit = ids >= 0
    # refract at the first hit (float64, flat normal)
    o, d = rays[hit, :3], rays[hit, 3:]
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    tri = m.vertices[m.faces[ids[hit]]].astype(np.float64)
    e1, e2 = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
    n = np.cross(e1, e2); n /= np.linalg.norm(n, axis=1, keepdims=True)
    t = np.einsum("ij,ij->i", tri[:, 0] - o, n) / np.einsum("ij,ij->i", d, n)
    p = o + t[:, None] * d
    cos_i = -np.einsum("ij,ij->i", d, n)
    flip = cos_i < 0
    n[flip] *= -1; cos_i = np.abs(cos_i)
    eta = 1.0 / IOR
    k2 = 1.0 - eta * eta * (1.0 - cos_i * cos_i)
    dr = eta * d + (eta * cos_i - np.sqrt(np.maximum(k2, 0)))[:, None] * n
    dr /= np.linalg.norm(dr, axis=1, keepdims=True)
    inner = np.ascontiguousarray(np.concatenate([p + 1e-5 * dr, dr], 1).astype(np.float32))
    print(f"{name} x{4 ** sub}: {len(F)} triangles, {hit.sum()} refracted rays of 4 views at {res}^2")
    base = None
    for W in (2, 4, 6, 8, 16):
        idw, st = wide(F, V, W, inner)
        if base is None:
            base = idw
        assert np.array_equal(idw, base), "every width must give the same hits"
        est = st[0] * (103 - 4 * 24 + W * 24) + st[1] * 72
        print(f"  W={W:2d}: inner {st[0]:6.2f} + leaf {st[1]:5.2f} = {st[0] + st[1]:6.2f} visits per ray; longest ray {int(st[3]):4d} visits ({int(st[2])} inner); "
              f"children per node {st[4]:.2f}; box tests per ray {st[5]:6.1f}; ~{est:6.0f} vector instructions per ray (7 + 24 W per inner visit, 72 per leaf)")


if __name__ == "__main__":
    main()
