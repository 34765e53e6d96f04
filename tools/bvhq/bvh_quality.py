#!/usr/bin/env python3
"""EXPERIMENT: node visits per ray of the product's traversal on (a) the Morton LBVH it builds and (b) a binned-SAH binary tree
built offline here, both collapsed/quantised/traversed by the same headers (tools/bvhq/bvhq.cpp).  Answers: how much would a
better builder buy?   usage: python tools/bvhq/bvh_quality.py [mesh] [subdiv]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from drt_amd import mesh_io, views

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libbvhq.so"))
lib.bvhq_visits.restype = ctypes.c_double
P = ctypes.c_void_p
lib.bvhq_visits.argtypes = [P, ctypes.c_int64, P, ctypes.c_int64, P, P, P, P, ctypes.c_int64, P, P]


def sah_tree(tri_lo, tri_hi, bins=16, leaf=1):
    """Top-down binned SAH over triangle boxes -> (order, child0, child1) with slots in depth-first leaf order."""
    n = len(tri_lo)
    cen = 0.5 * (tri_lo + tri_hi)
    order, c0, c1 = [], [], []

    def area(lo, hi):
        d = np.maximum(hi - lo, 0)
        return d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 2] * d[..., 0]

    sys.setrecursionlimit(100000)

    def build(ids):
        if len(ids) == 1:
            order.append(int(ids[0]))
            return ~(len(order) - 1)
        me = len(c0); c0.append(0); c1.append(0)
        clo, chi = cen[ids].min(0), cen[ids].max(0)
        best = (np.inf, None)
        for ax in range(3):
            ext = chi[ax] - clo[ax]
            if ext <= 0:
                continue
            b = np.minimum(((cen[ids, ax] - clo[ax]) / ext * bins).astype(int), bins - 1)
            lo_b = np.full((bins, 3), np.inf); hi_b = np.full((bins, 3), -np.inf); cnt = np.bincount(b, minlength=bins)
            np.minimum.at(lo_b, b, tri_lo[ids]); np.maximum.at(hi_b, b, tri_hi[ids])
            l_lo = np.minimum.accumulate(lo_b, 0); l_hi = np.maximum.accumulate(hi_b, 0); l_n = np.cumsum(cnt)
            r_lo = np.minimum.accumulate(lo_b[::-1], 0)[::-1]; r_hi = np.maximum.accumulate(hi_b[::-1], 0)[::-1]; r_n = np.cumsum(cnt[::-1])[::-1]
            for s in range(bins - 1):
                if l_n[s] == 0 or r_n[s + 1] == 0:
                    continue
                cost = area(l_lo[s], l_hi[s]) * l_n[s] + area(r_lo[s + 1], r_hi[s + 1]) * r_n[s + 1]
                if cost < best[0]:
                    best = (cost, (ax, b <= s))
        if best[1] is None:
            m = np.zeros(len(ids), bool); m[:len(ids) // 2] = True
        else:
            m = best[1][1]
        l = build(ids[m]); r = build(ids[~m])
        c0[me], c1[me] = l, r
        return me

    build(np.arange(n))
    return np.array(order, np.int32), np.array(c0, np.int32), np.array(c1, np.int32)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "horse"
    sub = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    m = mesh_io.read_ply(os.path.join(ROOT, "data", f"{name}_vh.ply"))
    for _ in range(sub):
        m = mesh_io.subdivide_midpoint(m)
    F = np.ascontiguousarray(m.faces, np.int32); V = np.ascontiguousarray(m.vertices, np.float32)
    c, e = views.mesh_frame(m.vertices)
    res = 256
    rays = []
    for k in (3, 20, 41, 60):
        R, K, Ri, Ki = views.turntable_cameras(c, e, 72, res, res)[k]
        o, d = views.generate_ray(res, res, Ki, Ri)
        rays.append(np.concatenate([o.numpy(), d.numpy()], 1).astype(np.float32))
    rays = np.ascontiguousarray(np.concatenate(rays))
    ID = np.empty(len(rays), np.int32)

    def run(order, c0, c1, sel):
        r = np.ascontiguousarray(rays[sel]); out = np.empty(len(r), np.int32); lv = ctypes.c_double()
        v = lib.bvhq_visits(F.ctypes.data, len(F), V.ctypes.data, len(V), order.ctypes.data, None if c0 is None else c0.ctypes.data,
                            None if c1 is None else c1.ctypes.data, r.ctypes.data, len(r), out.ctypes.data, ctypes.byref(lv))
        return v, lv.value, out

    order = np.empty(len(F), np.int32)
    v_all, _, ids = run(order, None, None, slice(None))
    hit = ids >= 0
    # candidates = rays that enter the root box at all would be closer to what k_trace sees; report hits and all
    v_l, lv_l, id_l = run(order, None, None, hit)
    tri = V[F]
    o2, a, b = sah_tree(tri.min(1), tri.max(1))
    v_s, lv_s, id_s = run(o2, a, b, hit)
    assert np.array_equal(id_l, id_s), "the two trees must give the same hits"
    print(f"{name} x{4 ** sub}: {len(F)} triangles, {hit.sum()} hitting primary rays")
    print(f"  Morton LBVH : {v_l:6.2f} visits per hitting ray ({lv_l:.2f} leaf visits)   all rays {v_all:.2f}")
    print(f"  binned SAH  : {v_s:6.2f} visits per hitting ray ({lv_s:.2f} leaf visits)   -> {100 * (1 - v_s / v_l):.0f} % fewer")


if __name__ == "__main__":
    main()
