"""TEST INFRASTRUCTURE ONLY -- an independent acceptance check for the remesh step between optimisation passes.

The reference does not remesh itself: `Meshlabserver.remesh` (reference optim.py:12-52) shells out to MeshLab 2020.04's filter
"Remeshing: Isotropic Explicit Remeshing" with the parameter set of optim.py:17-32

    Iterations 3, Adaptive false, TargetLen L, FeatureDeg 180 (no crease is a feature), CheckSurfDist true, MaxSurfDist 1 (absolute),
    SplitFlag / CollapseFlag / SwapFlag / SmoothFlag / ReprojectFlag all true,

and reloads the result (`scene.update_mesh`, optim.py:52).  MeshLab is an external program that is not in this image (nor is its library,
vcglib), so there are no outputs of it to compare against: PARITY UNPINNED for this step, by necessity.  What CAN be checked
independently of the product's two remeshers (drt_amd/csrc/drt_remesh.cpp on the host, drt_amd/csrc/drt_remesh_gpu.hip on the device)
is the CONTRACT that parameter set defines -- the published algorithm (Botsch & Kobbelt 2004, "A remeshing approach to multiresolution
modeling", section 4; vcglib's IsotropicRemeshing follows it) run with those switches:

  * refine splits every edge longer than 4/3 L, collapse removes every edge shorter than 4/5 L unless the operation is vetoed
    (link condition, a normal flip, a new edge above 4/3 L, MaxSurfDist), so after the last iteration the edge lengths concentrate in
    [4/5 L, 4/3 L]; smoothing and re-projection run AFTER them and move lengths a little, which is why the band is a share, not a bound;
  * swap equalises valences towards 6 (interior vertices of a closed surface);
  * re-projection puts every vertex ON the input surface; with CheckSurfDist no local operation may move the surface further than
    MaxSurfDist from the input, checked at the new faces' sample points -- one-sided, output -> input;
  * every operation preserves the topology: a closed oriented 2-manifold stays one, with the same number of components and genus.

This module measures exactly that from raw arrays with numpy (no product code, no torch): `check(...)` returns the measurements and a list
of violated clauses.  tests/test_remesh.py and tests/test_gpu_remesh.py hold BOTH remeshers against it.  Thresholds that are shares
(not hard bounds of the algorithm) are stated with the reason next to them.
"""
from __future__ import annotations

import numpy as np


def _directed_edges(F):
    return np.stack([F[:, [0, 1, 2]].reshape(-1), F[:, [1, 2, 0]].reshape(-1)], axis=1)


def topology(V, F):
    """Closed oriented 2-manifold?  Returns a dict: ok flags, Euler characteristic, components, genus (per component sum)."""
    V, F = np.asarray(V), np.asarray(F, dtype=np.int64)
    nv = len(V)
    out = {"n_vertices": nv, "n_faces": len(F)}
    out["indices_in_range"] = bool(len(F) > 0 and F.min() >= 0 and F.max() < nv)
    if not out["indices_in_range"]:
        out["ok"] = False
        return out
    out["no_unused_vertices"] = bool(len(np.unique(F)) == nv)
    out["no_degenerate_index_faces"] = bool(((F[:, 0] != F[:, 1]) & (F[:, 1] != F[:, 2]) & (F[:, 0] != F[:, 2])).all())
    de = _directed_edges(F)
    key = de[:, 0] * nv + de[:, 1]
    rev = de[:, 1] * nv + de[:, 0]
    # every directed edge exactly once (consistent orientation, no edge shared by three faces) and its reverse present (closed)
    out["oriented"] = bool(len(np.unique(key)) == len(key))
    out["closed"] = bool(np.array_equal(np.sort(key), np.sort(rev)))
    canon = np.sort(F, axis=1)
    out["no_duplicate_faces"] = bool(len(np.unique(canon, axis=0)) == len(F))
    n_edges = len(key) // 2
    chi = nv - n_edges + len(F)
    # components by union-find over the edges
    parent = np.arange(nv)

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    for a, b in de[de[:, 0] < de[:, 1]]:
        ra, rb = find(a), find(b)
        if ra != rb:
            parent[ra] = rb
    comps = len({find(a) for a in range(nv)})
    out.update(euler_characteristic=int(chi), components=int(comps), genus_sum=int((2 * comps - chi) // 2))
    # vertex manifoldness: the faces around a vertex form ONE cycle -- with a closed, oriented, edge-manifold mesh that is: the number of
    # faces at a vertex equals the number of distinct neighbours, and the neighbour graph of the link is connected (one cycle, not two)
    valence = np.bincount(F.reshape(-1), minlength=nv)
    nbr = np.unique(np.stack([np.minimum(de[:, 0], de[:, 1]), np.maximum(de[:, 0], de[:, 1])], 1), axis=0)
    deg = np.bincount(nbr.reshape(-1), minlength=nv)
    link_ok = bool((valence == deg).all())
    if link_ok:
        # one cycle per vertex: nxt[(v, a)] = b for every face (v, a, b) taken in cyclic order; walking it from any neighbour must come
        # back after exactly valence(v) steps (two fans touching in v would close earlier)
        nxt, start = {}, {}
        for f in F:
            for k in range(3):
                v, a, b = int(f[k]), int(f[(k + 1) % 3]), int(f[(k + 2) % 3])
                nxt[(v, a)] = b
                start.setdefault(v, a)
        for v in range(nv):
            cur, steps, s0 = start[v], 0, start[v]
            while True:
                cur = nxt.get((v, cur))
                steps += 1
                if cur is None or cur == s0 or steps > valence[v]:
                    break
            if cur != s0 or steps != valence[v]:
                link_ok = False
                break
    out["vertex_manifold"] = link_ok
    out["valence"] = valence
    out["ok"] = all(out[k] for k in ("indices_in_range", "no_unused_vertices", "no_degenerate_index_faces", "oriented", "closed", "no_duplicate_faces", "vertex_manifold"))
    return out


def point_triangle_distance(P, A, B, C):
    """Distance of points P [n,3] to triangles (A, B, C) [n,3] each, pairwise (Ericson, Real-Time Collision Detection 5.1.5)."""
    ab, ac, ap = B - A, C - A, P - A
    d1, d2 = np.einsum("ij,ij->i", ab, ap), np.einsum("ij,ij->i", ac, ap)
    bp = P - B
    d3, d4 = np.einsum("ij,ij->i", ab, bp), np.einsum("ij,ij->i", ac, bp)
    cp = P - C
    d5, d6 = np.einsum("ij,ij->i", ab, cp), np.einsum("ij,ij->i", ac, cp)
    vc, vb, va = d1 * d4 - d3 * d2, d5 * d2 - d1 * d6, d3 * d6 - d5 * d4
    with np.errstate(divide="ignore", invalid="ignore"):
        denom = va + vb + vc
        v = np.where(denom != 0, vb / denom, 0.0)
        w = np.where(denom != 0, vc / denom, 0.0)
        Q = A + ab * v[:, None] + ac * w[:, None]                      # interior
        t_ab = np.where((d1 - d3) != 0, d1 / (d1 - d3), 0.0)
        t_ac = np.where((d2 - d6) != 0, d2 / (d2 - d6), 0.0)
        t_bc = np.where(((d4 - d3) + (d5 - d6)) != 0, (d4 - d3) / ((d4 - d3) + (d5 - d6)), 0.0)
    Q = np.where(((va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0))[:, None], B + (C - B) * t_bc[:, None], Q)
    Q = np.where(((vb <= 0) & (d2 >= 0) & (d6 <= 0))[:, None], A + ac * t_ac[:, None], Q)
    Q = np.where(((vc <= 0) & (d1 >= 0) & (d3 <= 0))[:, None], A + ab * t_ab[:, None], Q)
    Q = np.where(((d6 >= 0) & (d5 <= d6))[:, None], C, Q)
    Q = np.where(((d3 >= 0) & (d4 <= d3))[:, None], B, Q)
    Q = np.where(((d1 <= 0) & (d2 <= 0))[:, None], A, Q)
    return np.linalg.norm(P - Q, axis=1)


def distance_to_surface(P, V, F, chunk=128):
    """min over all triangles, brute force in chunks of points: O(len(P) * len(F))."""
    P, V, F = np.asarray(P, dtype=np.float64), np.asarray(V, dtype=np.float64), np.asarray(F, dtype=np.int64)
    A, B, C = V[F[:, 0]], V[F[:, 1]], V[F[:, 2]]
    nf = len(F)
    out = np.empty(len(P))
    for s in range(0, len(P), chunk):
        p = P[s:s + chunk]
        n = len(p)
        d = point_triangle_distance(np.repeat(p, nf, axis=0), np.tile(A, (n, 1)), np.tile(B, (n, 1)), np.tile(C, (n, 1))).reshape(n, nf)
        out[s:s + chunk] = d.min(axis=1)
    return out


def signed_volume(V, F):
    t = np.asarray(V)[np.asarray(F)]
    return float(np.einsum("ij,ij->i", t[:, 0], np.cross(t[:, 1], t[:, 2])).sum() / 6.0)


def check(V_in, F_in, V_out, F_out, target_len, max_surf_dist=1.0, sample_every=1, seed=0, max_samples=4000):
    """Measure the output of a remesh call against the contract of the reference's parameter set.  Returns (report, violations)."""
    V_in, F_in = np.asarray(V_in, dtype=np.float64), np.asarray(F_in, dtype=np.int64)
    V_out, F_out = np.asarray(V_out, dtype=np.float64), np.asarray(F_out, dtype=np.int64)
    L = float(target_len)
    rep, bad = {}, []
    t_in, t_out = topology(V_in, F_in), topology(V_out, F_out)
    rep["topology"] = {k: v for k, v in t_out.items() if k != "valence"}
    if not t_out["ok"]:
        bad.append("output is not a closed oriented 2-manifold: " + ", ".join(k for k, v in t_out.items() if v is False))
        return rep, bad
    if (t_in.get("components"), t_in.get("genus_sum")) != (t_out["components"], t_out["genus_sum"]):
        bad.append(f"topology changed: components/genus {t_in.get('components')}/{t_in.get('genus_sum')} -> {t_out['components']}/{t_out['genus_sum']}")
    # ---- geometry of the faces
    tri = V_out[F_out]
    nrm = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    area2 = np.linalg.norm(nrm, axis=1)
    rep["min_face_area"] = float(area2.min() / 2)
    if not (area2 > 0).all():
        bad.append("zero-area faces")
    de = _directed_edges(F_out)
    und = de[de[:, 0] < de[:, 1]]
    el = np.linalg.norm(V_out[und[:, 0]] - V_out[und[:, 1]], axis=1)
    in_band = float(((el >= 0.8 * L) & (el <= 4.0 / 3.0 * L)).mean())
    rep["edge_length"] = {"mean_over_L": float(el.mean() / L), "min_over_L": float(el.min() / L), "max_over_L": float(el.max() / L), "share_in_band": in_band}
    # shares, not bounds: smoothing + re-projection follow the last split / collapse, and vetoed collapses leave short edges behind
    if in_band < 0.9:
        bad.append(f"only {in_band:.3f} of the edges in [4/5 L, 4/3 L]")
    if not (0.85 <= el.mean() / L <= 1.15):
        bad.append(f"mean edge length {el.mean() / L:.3f} L")
    if el.max() > 4.0 / 3.0 * L * 1.25:          # an edge a quarter above the split threshold survived the last refine + relaxation
        bad.append(f"longest edge {el.max() / L:.3f} L")
    # ---- valences (swap step): a closed surface averages 6 - 12 (1 - g) / V
    val = t_out["valence"]
    rep["valence"] = {"mean_abs_dev_from_6": float(np.abs(val - 6).mean()), "share_5_to_7": float(((val >= 5) & (val <= 7)).mean()), "min": int(val.min()), "max": int(val.max())}
    if rep["valence"]["share_5_to_7"] < 0.85:
        bad.append(f"valences: only {rep['valence']['share_5_to_7']:.3f} in 5..7")
    if val.min() < 3:
        bad.append("a vertex of valence < 3")
    # ---- orientation and folds
    vol_in, vol_out = signed_volume(V_in, F_in), signed_volume(V_out, F_out)
    rep["volume_ratio"] = vol_out / vol_in
    if not (vol_out * vol_in > 0 and abs(vol_out / vol_in - 1) < 0.05):
        bad.append(f"volume ratio {vol_out / vol_in:.4f} (orientation flipped, or the surface moved)")
    # adjacent face normals: for each undirected edge the two faces
    nv = len(V_out)
    face_of = {}
    for fi, f in enumerate(F_out):
        for k in range(3):
            face_of[(int(f[k]), int(f[(k + 1) % 3]))] = fi
    n_unit = nrm / area2[:, None]
    cosd = np.array([np.dot(n_unit[face_of[(a, b)]], n_unit[face_of[(b, a)]]) for a, b in und])
    rep["min_dihedral_cos"] = float(cosd.min())
    if cosd.min() <= -0.9:
        bad.append(f"folded face pair (cos {cosd.min():.3f})")
    # ---- on the input surface (ReprojectFlag) and within MaxSurfDist (CheckSurfDist): one-sided, output -> input
    rng = np.random.default_rng(seed)
    pick = rng.permutation(len(V_out))[:max_samples][::sample_every]
    d_v = distance_to_surface(V_out[pick], V_in, F_in)
    rep["vertex_to_input"] = {"max": float(d_v.max()), "n": int(len(pick))}
    scale = float(np.abs(V_in).max())
    if d_v.max() > 1e-5 * scale + 1e-4:          # float32 positions of a ~100 mm object: 1e-5 relative, plus the float32 tracer's own vertices
        bad.append(f"a vertex is {d_v.max():.2e} off the input surface (re-projection)")
    fpick = rng.permutation(len(F_out))[:max_samples][::sample_every]
    cent = tri[fpick].mean(axis=1)
    mids = 0.5 * (V_out[und[:, 0]] + V_out[und[:, 1]])[rng.permutation(len(und))[:max_samples][::sample_every]]
    d_s = distance_to_surface(np.concatenate([cent, mids]), V_in, F_in)
    q95, q99 = (float(x) for x in np.quantile(d_s, [0.95, 0.99]))
    rep["surface_samples_to_input"] = {"max": float(d_s.max()), "mean": float(d_s.mean()), "p95": q95, "p99": q99, "n": int(len(d_s)),
                                       "share_within_max_surf_dist": float((d_s <= max_surf_dist).mean())}
    # MaxSurfDist is a veto on single collapses / swaps, evaluated at the new faces' barycentres and edge midpoints WHEN THE OPERATION IS
    # MADE; the relaxation and the re-projection of the same iteration move the vertices afterwards, unchecked, and a ridge of the input that
    # an L-long edge cannot follow is cut by up to ~L^2 / (8 r).  So on the FINAL mesh the clause is a share with a bounded tail, not a bound:
    # 95 % of the samples within MaxSurfDist, none beyond MaxSurfDist + L / 4.
    if q95 > max_surf_dist:
        bad.append(f"95th percentile of the face samples is {q95:.3f} from the input surface (MaxSurfDist {max_surf_dist})")
    if d_s.max() > max_surf_dist + 0.25 * L:
        bad.append(f"a face sample is {d_s.max():.3f} from the input surface (MaxSurfDist {max_surf_dist} + L / 4 = {max_surf_dist + 0.25 * L:.3f})")
    return rep, bad
