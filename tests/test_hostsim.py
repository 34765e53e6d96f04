"""Device math (drt_amd/csrc/*.h compiled for the host by tests/hostsim) against the oracle.

This is how the LBVH build logic, the traversal and the hand-derived adjoint are
checked in the GPU-less container; the `-m gpu` tests repeat the comparisons on the
real kernels."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import IOR, data_path, fixture_view, golden
from drt_amd import mesh_io, views
from oracle import diffrender_oracle as orc


class HostScene:
    def __init__(self, hs, faces, verts64):
        self.hs = hs
        self.f32 = np.ascontiguousarray(faces, dtype=np.int32)
        self.v64 = np.ascontiguousarray(verts64, dtype=np.float64)
        self.v32 = self.v64.astype(np.float32)
        self.h = hs.hs_create(self.f32.ctypes.data, len(self.f32), self.v32.ctypes.data, len(self.v32))

    def __del__(self):
        self.hs.hs_destroy(self.h)

    def intersect(self, rays, any_hit=False):
        rays = np.ascontiguousarray(rays, dtype=np.float32)
        n = len(rays)
        T = np.empty(n, np.float32); ID = np.empty(n, np.int32); vis = np.empty(n, np.uint32)
        self.hs.hs_intersect(self.h, rays.ctypes.data, n, T.ctypes.data, ID.ctypes.data, int(any_hit), vis.ctypes.data)
        return T, ID, vis

    def render(self, o, d, ior=IOR):
        n = len(o)
        o = np.ascontiguousarray(o); d = np.ascontiguousarray(d)
        out = dict(out_ori=np.zeros((n, 3)), out_dir=np.zeros((n, 3)), mask=np.zeros((n, 3), np.uint8),
                   f1=np.zeros(n, np.int32), f2=np.zeros(n, np.int32))
        self.hs.hs_render_forward(self.h, self.v64.ctypes.data, o.ctypes.data, d.ctypes.data, n, ior, orc.EXT_IOR,
                                  out["out_ori"].ctypes.data, out["out_dir"].ctypes.data, out["mask"].ctypes.data,
                                  out["f1"].ctypes.data, out["f2"].ctypes.data)
        return out

    def backward(self, o, d, f1, f2, g_ori, g_dir, ior=IOR):
        o = np.ascontiguousarray(o); d = np.ascontiguousarray(d)
        gv = np.zeros_like(self.v64)
        self.hs.hs_render_backward(self.h, self.v64.ctypes.data, o.ctypes.data, d.ctypes.data, len(o), ior, orc.EXT_IOR,
                                   f1.ctypes.data, f2.ctypes.data, None if g_ori is None else g_ori.ctypes.data,
                                   None if g_dir is None else g_dir.ctypes.data, gv.ctypes.data)
        return gv


@pytest.fixture(scope="module")
def hand():
    return mesh_io.read_ply(data_path("hand_vh.ply"))


def _camera_rays(mesh, res, view):
    c, ext = views.mesh_frame(mesh.vertices)
    R, K, Rinv, Kinv = views.turntable_cameras(c, ext, 72, res, res)[view]
    return views.generate_ray(res, res, Kinv, Rinv)


def _rays32(o, d):
    return torch.cat([o.float(), d.float()], 1).numpy()


@pytest.mark.parametrize("name,subdiv", [("hand_vh.ply", 0), ("hand_vh.ply", 1), ("horse_vh.ply", 0), ("mouse_vh.ply", 0)])
def test_lbvh_is_sound(hostsim, name, subdiv):
    m = mesh_io.read_ply(data_path(name))
    for _ in range(subdiv):
        m = mesh_io.subdivide_midpoint(m)
    assert m.is_watertight
    s = HostScene(hostsim, m.faces, m.vertices)
    assert hostsim.hs_check(s.h) == 0
    order = np.empty(len(m.faces), np.int32)
    hostsim.hs_sorted_faces(s.h, order.ctypes.data)
    assert np.array_equal(np.sort(order), np.arange(len(m.faces)))       # a permutation
    h = hostsim.hs_height(s.h)
    assert np.log2(len(m.faces)) <= h <= 30 + np.ceil(np.log2(len(m.faces)))


def test_traversal_equals_bruteforce(hostsim, hand):
    s = HostScene(hostsim, hand.faces, hand.vertices)
    rng = np.random.default_rng(3)
    c, ext = views.mesh_frame(hand.vertices)
    batches = []
    for view in (0, 17, 44):
        batches.append(_rays32(*_camera_rays(hand, 96, view)))
    # random rays from inside and outside, un-normalised directions, axis-aligned directions (zero components)
    o = rng.uniform(-1, 1, (4000, 3)) * ext + c
    d = rng.standard_normal((4000, 3)) * rng.uniform(0.1, 30, (4000, 1))
    d[:300, 0] = 0; d[300:600, 1] = 0; d[600:900, 2] = 0; d[900:1000, :2] = 0
    batches.append(np.concatenate([o, d], 1).astype(np.float32))
    # rays starting exactly on vertices / aimed at vertices and edge midpoints (ties between faces)
    tgt = np.concatenate([hand.vertices[:1500], 0.5 * (hand.vertices[hand.faces[:1500, 0]] + hand.vertices[hand.faces[:1500, 1]])])
    eye = c + np.array([0, 0, 2.5 * ext])
    batches.append(np.concatenate([np.broadcast_to(eye, tgt.shape), tgt - eye], 1).astype(np.float32))
    for rays in batches:
        T, ID, _ = s.intersect(rays)
        To, IDo = orc.trace_closest(s.f32, s.v32, rays)
        assert np.array_equal(ID, IDo)
        assert np.array_equal(T, To)          # bit-exact t
        _, IDa, _ = s.intersect(rays, any_hit=True)
        assert np.array_equal(IDa >= 0, IDo >= 0)


def test_traversal_degenerate_meshes(hostsim):
    rays = np.array([[0.2, 0.2, 1, 0, 0, -1], [5, 5, 1, 0, 0, -1]], np.float32)
    for faces, verts in [
        (np.zeros((0, 3), np.int32), np.zeros((0, 3))),                                      # empty mesh
        (np.array([[0, 1, 2]], np.int32), np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0.]])),      # one triangle
        (np.array([[0, 1, 2], [0, 1, 2], [0, 1, 2]], np.int32), np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0.]])),  # duplicates: equal keys
        (np.array([[0, 1, 2], [0, 0, 0]], np.int32), np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0.]])),  # zero-area triangle
    ]:
        s = HostScene(hostsim, faces, verts)
        assert hostsim.hs_check(s.h) == 0
        T, ID, _ = s.intersect(rays)
        To, IDo = orc.trace_closest(s.f32, s.v32, rays)
        assert np.array_equal(ID, IDo) and np.array_equal(T, To)


@pytest.mark.parametrize("name", ["hand_r64_v5", "hand_r128_v23", "hand_r128_v41"])
def test_render_path_vs_golden_and_oracle(hostsim, hand, name):
    g = golden(name)
    o, d, sp, valid = fixture_view(g)
    s = HostScene(hostsim, hand.faces, hand.vertices)
    r = s.render(o.numpy(), d.numpy())
    vi = np.flatnonzero(r["mask"][:, 0])
    assert np.array_equal(vi, g["valid_ind"])
    f1 = np.full(len(o), -1, np.int64); f1[g["b1_ind"]] = g["b1_face"]
    assert np.array_equal(r["f1"], f1)
    np.testing.assert_allclose(r["out_ori"][vi], g["out_ori"], rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(r["out_dir"][vi], g["out_dir"], rtol=1e-10, atol=1e-11)
    assert not r["out_ori"][r["mask"][:, 0] == 0].any() and np.all(r["f2"][r["mask"][:, 0] == 0] == -1)
    # ray loss + its gradient (golden), then the linear functional that exercises grad_out_ori
    loss = np.zeros(1); gdir = np.zeros((len(o), 3))
    spn = sp.numpy().copy(); vn = valid.numpy().astype(np.uint8)
    hostsim.hs_ray_loss(r["out_ori"].ctypes.data, r["out_dir"].ctypes.data, r["mask"].ctypes.data, spn.ctypes.data, vn.ctypes.data,
                        len(o), loss.ctypes.data, gdir.ctypes.data)
    assert loss[0] == pytest.approx(float(g["ray_loss"]), rel=1e-11)
    gv = s.backward(o.numpy(), d.numpy(), r["f1"], r["f2"], None, gdir)
    np.testing.assert_allclose(gv, g["grad_ray_loss"], rtol=1e-8, atol=1e-10 * np.abs(g["grad_ray_loss"]).max())
    rng = np.random.default_rng(int(g["lin_seed"]))
    w_ori = rng.standard_normal(o.shape); w_dir = rng.standard_normal(o.shape)
    gv = s.backward(o.numpy(), d.numpy(), r["f1"], r["f2"], w_ori, w_dir)
    np.testing.assert_allclose(gv, g["grad_lin"], rtol=1e-8, atol=1e-10 * np.abs(g["grad_lin"]).max())
    # fused single pass == two-pass
    loss_f = np.zeros(1); gv_f = np.zeros_like(s.v64); nv = ctypes.c_int64(0)
    on, dn = o.numpy().copy(), d.numpy().copy()
    hostsim.hs_fused(s.h, s.v64.ctypes.data, on.ctypes.data, dn.ctypes.data, spn.ctypes.data, vn.ctypes.data, len(o), IOR, orc.EXT_IOR,
                     loss_f.ctypes.data, gv_f.ctypes.data, ctypes.byref(nv))
    assert loss_f[0] == pytest.approx(float(g["ray_loss"]), rel=1e-11)
    np.testing.assert_allclose(gv_f, g["grad_ray_loss"], rtol=1e-8, atol=1e-10 * np.abs(g["grad_ray_loss"]).max())
    assert nv.value == int((valid.numpy() & (r["mask"][:, 0] == 1)).sum())


def test_bounce_adjoint_vs_autograd(hostsim):
    """drt_shade.h forward + hand-derived reverse mode against autograd of the oracle's graph,
    entering and exiting hits, including total internal reflection flags."""
    rng = np.random.default_rng(11)
    n = 400
    tri = rng.standard_normal((n, 3, 3)) * 10
    centroid = tri.mean(1)
    d = rng.standard_normal((n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = centroid - d * rng.uniform(5, 50, (n, 1)) + rng.standard_normal((n, 3)) * 0.5
    g_no = rng.standard_normal((n, 3)); g_wt = rng.standard_normal((n, 3))
    new_o = np.zeros((n, 3)); wt = np.zeros((n, 3)); tir = np.zeros(n, np.uint8); t = np.zeros(n)
    g_tri = np.zeros((n, 3, 3)); g_o = np.zeros((n, 3)); g_d = np.zeros((n, 3))
    hostsim.hs_bounce(o.ctypes.data, d.ctypes.data, tri.ctypes.data, n, IOR, orc.EXT_IOR, g_no.ctypes.data, g_wt.ctypes.data,
                      new_o.ctypes.data, wt.ctypes.data, tir.ctypes.data, t.ctypes.data, g_tri.ctypes.data, g_o.ctypes.data, g_d.ctypes.data)
    to, td, ttri = (torch.tensor(a, requires_grad=True) for a in (o, d, tri))
    u, v, tt, nn = orc.moller_trumbore(to, td, ttri)
    wo = -td
    cos_i = orc._dot(wo, nn).clamp(-1, 1)
    exc = torch.logical_not(cos_i > 0)
    sgn = torch.where(exc, -torch.ones_like(tt), torch.ones_like(tt))
    eta_i = torch.where(exc, torch.full_like(tt, IOR), torch.full_like(tt, orc.EXT_IOR))
    eta_t = torch.where(exc, torch.full_like(tt, orc.EXT_IOR), torch.full_like(tt, IOR))
    nn = nn * sgn.view(-1, 1)
    ref_tir = orc.fresnel_tir(cos_i * sgn, eta_i, eta_t)
    ref_wt = orc.refract_dir(wo, nn, eta_i / eta_t)
    ref_no = to + tt.view(-1, 1) * td + 1e-5 * ref_wt
    assert exc.sum() > 50 and (~exc).sum() > 50 and ref_tir.sum() > 5
    assert np.array_equal(tir.astype(bool), ref_tir.numpy())
    np.testing.assert_allclose(t, tt.detach().numpy(), rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(wt, ref_wt.detach().numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(new_o, ref_no.detach().numpy(), rtol=1e-10, atol=1e-10)
    scalar = (ref_no * torch.tensor(g_no)).sum() + (ref_wt * torch.tensor(g_wt)).sum()
    go, gd, gt = torch.autograd.grad(scalar, (to, td, ttri))
    for mine, ref in ((g_tri, gt), (g_o, go), (g_d, gd)):
        ref = ref.numpy()
        np.testing.assert_allclose(mine, ref, rtol=1e-7, atol=1e-9 * np.abs(ref).max())


def _cam50(g):
    return np.concatenate([np.asarray(g[k], dtype=np.float64).reshape(-1) for k in ("R", "K", "Rinv", "Kinv")])


@pytest.mark.parametrize("name", ["hand_r64_v5", "hand_r128_v23", "hand_r128_v41"])
def test_silhouette_branch_vs_golden(hostsim, hand, name):
    g = golden(name)
    topo = golden("hand_topology")
    res = int(g["res"])
    o, d, _, _ = fixture_view(g)
    s = HostScene(hostsim, hand.faces, hand.vertices)
    E2F = np.ascontiguousarray(topo["E2F"]); Edges = topo["Edges"]
    origin3 = o[0].numpy().copy()
    flags = np.zeros(len(E2F), np.uint8)
    hostsim.hs_silhouette_flags(s.v64.ctypes.data, E2F.ctypes.data, len(E2F), origin3.ctypes.data, flags.ctypes.data)
    sil = np.ascontiguousarray(Edges[flags.astype(bool)])
    assert np.array_equal(sil, g["sil_edges"])
    cam = _cam50(g)
    index = np.zeros((len(sil), 2), np.int64); f = np.zeros(len(sil), np.float32)
    hostsim.hs_edge_sample_forward(s.h, s.v64.ctypes.data, sil.ctypes.data, len(sil), cam.ctypes.data, origin3.ctypes.data,
                                   index.ctypes.data, f.ctypes.data)
    valid_edge = np.abs(f) > 1e-5
    idx = index[valid_edge]
    keep = (idx[:, 0] < res - 1) & (idx[:, 1] < res - 1) & (idx[:, 0] >= 0) & (idx[:, 1] >= 0)
    assert np.array_equal(idx[keep], g["vh_index"])
    # vh_loss = sum |soft[y, x] - 0.5| and its gradient: d/d output = -sign(soft - 0.5)
    hit = np.zeros(res * res, dtype=np.uint8); hit[g["b1_ind"]] = 1
    soft = views.process_mask(hit.reshape(res, res))
    diff = soft[idx[:, 1].clip(0, res - 1), idx[:, 0].clip(0, res - 1)] - 0.5
    assert np.abs(diff[keep]).sum() == pytest.approx(float(g["vh_loss"]), rel=1e-12)
    coef = np.zeros(len(sil))
    coef[np.flatnonzero(valid_edge)[keep]] = -np.sign(diff[keep])
    gv = np.zeros_like(s.v64)
    hostsim.hs_edge_sample_backward(s.v64.ctypes.data, sil.ctypes.data, len(sil), cam.ctypes.data, f.ctypes.data, coef.ctypes.data, 1, gv.ctypes.data)
    ref = g["grad_vh"]
    np.testing.assert_allclose(gv, ref, rtol=1e-8, atol=1e-10 * np.abs(ref).max())


def test_dihedral_vs_golden(hostsim):
    g = golden("hand_smooth_sm")
    V = np.ascontiguousarray(g["vertices"].astype(np.float64)); E2F = np.ascontiguousarray(g["E2F"])
    cosv = np.zeros(len(E2F)); loss = np.zeros(1); gv = np.zeros_like(V)
    hostsim.hs_dihedral(V.ctypes.data, E2F.ctypes.data, len(E2F), 2, cosv.ctypes.data, None, loss.ctypes.data, gv.ctypes.data)
    np.testing.assert_allclose(cosv, g["dihedral_cos"], rtol=1e-10, atol=1e-12)
    assert loss[0] == pytest.approx(float(g["sm_loss"]), rel=1e-12)
    ref = g["grad_sm"]
    np.testing.assert_allclose(gv, ref, rtol=1e-8, atol=1e-10 * np.abs(ref).max())
    # explicit-adjoint mode with the same d loss / d cos
    gv2 = np.zeros_like(V); gc = -1.0 / (1.0 + cosv)
    hostsim.hs_dihedral(V.ctypes.data, E2F.ctypes.data, len(E2F), 1, None, gc.ctypes.data, None, gv2.ctypes.data)
    np.testing.assert_allclose(gv2, gv, rtol=1e-12, atol=1e-14 * np.abs(gv).max())


def test_closest_point_matches_bruteforce(hostsim):
    """drt_closest.h (BVH-pruned, Voronoi-region routine) == brute force with an independent formulation."""
    mesh = mesh_io.read_ply(data_path("hand_vh.ply"))
    F = np.ascontiguousarray(mesh.faces, dtype=np.int32)
    V = np.ascontiguousarray(mesh.vertices, dtype=np.float32)
    h = ctypes.c_void_p(hostsim.hs_create(F.ctypes.data, len(F), V.ctypes.data, len(V)))
    rng = np.random.default_rng(3)
    lo, hi = V.min(0), V.max(0)
    pts = np.concatenate([
        rng.uniform(lo - 0.3 * (hi - lo), hi + 0.3 * (hi - lo), size=(600, 3)),          # around and inside the mesh
        V[rng.integers(0, len(V), 100)].astype(np.float64),                                  # exactly on vertices
        V[F[rng.integers(0, len(F), 100)]].astype(np.float64).mean(1),                        # on faces
        V[rng.integers(0, len(V), 100)].astype(np.float64) + rng.normal(0, 1e-3, (100, 3)),   # very close to the surface
        rng.uniform(-1e4, 1e4, size=(20, 3)),                                                 # far away
    ])
    pts = np.ascontiguousarray(pts)
    dist = np.empty(len(pts)); face = np.empty(len(pts), dtype=np.int32); closest = np.empty((len(pts), 3))
    hostsim.hs_closest_point(h, pts.ctypes.data, len(pts), dist.ctypes.data, face.ctypes.data, closest.ctypes.data)
    ref_d, _ = orc.point_mesh_distance(pts, V, F)
    np.testing.assert_allclose(dist, ref_d, rtol=1e-9, atol=1e-9)
    # the reported point lies on the reported face and realises the distance
    np.testing.assert_allclose(np.linalg.norm(pts - closest, axis=1), dist, rtol=1e-12, atol=1e-12)
    d_face, _ = zip(*[orc.point_mesh_distance(pts[i:i + 1], V, F[face[i]:face[i] + 1]) for i in range(0, len(pts), 37)])
    np.testing.assert_allclose(np.concatenate(d_face), dist[::37], rtol=1e-9, atol=1e-9)
    assert dist[600:700].max() < 1e-12                        # a vertex is on the surface
    assert dist[700:800].max() < 1e-4                         # face centroids (float32 vertices, float64 mean)
    # within_distance (the remesher's surface-distance rule: early exit, bound = the radius) gives the verdict  dist <= radius  of the
    # full query -- at the distance itself, one ulp either side of it, and at radii all over the range
    for radius in (dist, np.nextafter(dist, np.inf), np.nextafter(dist, -np.inf), dist * rng.uniform(0.2, 3.0, len(dist)),
                   np.full(len(dist), 0.05), np.zeros(len(dist)), np.full(len(dist), np.nan), np.full(len(dist), np.inf)):
        radius = np.ascontiguousarray(radius)
        got = np.empty(len(pts), dtype=np.uint8)
        hostsim.hs_within_distance(h, pts.ctypes.data, radius.ctypes.data, len(pts), got.ctypes.data)
        with np.errstate(invalid="ignore"):
            assert np.array_equal(got.astype(bool), (dist <= radius) & (radius >= 0))
    # a search that starts bounded (the remesher's projection query) reports the point and the face of the unbounded one -- bit for bit --
    # whether the bound was enough (strictly farther than the surface) or the fallback ran
    for hint in (dist * 1.5 + 1e-9, dist, dist * 0.5, np.full(len(dist), 0.05), np.full(len(dist), 1e9)):
        hint = np.ascontiguousarray(hint)
        c2 = np.empty((len(pts), 3)); f2 = np.empty(len(pts), dtype=np.int32); found = np.empty(len(pts), dtype=np.uint8)
        hostsim.hs_closest_near(h, pts.ctypes.data, hint.ctypes.data, len(pts), c2.ctypes.data, f2.ctypes.data, found.ctypes.data)
        assert np.array_equal(c2, closest) and np.array_equal(f2, face)
        clear = np.abs(hint - dist) > 1e-9 * (1 + dist)          # (at hint == dist the rounding of dist = sqrt(dist2) decides: either way is right)
        assert np.array_equal(found.astype(bool)[clear], (dist < hint)[clear])
    hostsim.hs_destroy(h)


def test_degenerate_faces_vs_golden(hostsim, hand):
    """The device math on the mesh with a zero-length edge (golden from the reference): never hits a zero-area face,
    same silhouette set, NaN dihedral cosines / smoothness gradient in the same places (limit_hook zeroes them later)."""
    g = golden("hand_degenerate")
    topo = golden("hand_topology")
    res = int(g["res"])
    V = np.ascontiguousarray(g["vertices"].astype(np.float64))
    s = HostScene(hostsim, hand.faces, V)
    assert hostsim.hs_check(s.h) == 0
    o, d = views.generate_ray(res, res, g["Kinv"], g["Rinv"])
    T, ID, _ = s.intersect(_rays32(o, d))
    f1 = np.full(res * res, -1, np.int32); f1[g["b1_ind"]] = g["b1_face"]
    assert np.array_equal(ID, f1)
    r = s.render(o.numpy(), d.numpy(), float(g["ior"]))
    assert np.array_equal(np.flatnonzero(r["mask"][:, 0]), g["valid_ind"])
    np.testing.assert_allclose(r["out_dir"][g["valid_ind"]], g["out_dir"], rtol=1e-10, atol=1e-11)
    E2F = np.ascontiguousarray(topo["E2F"]); Edges = topo["Edges"]
    origin3 = o[0].numpy().copy()
    flags = np.zeros(len(E2F), np.uint8)
    hostsim.hs_silhouette_flags(s.v64.ctypes.data, E2F.ctypes.data, len(E2F), origin3.ctypes.data, flags.ctypes.data)
    sil = np.ascontiguousarray(Edges[flags.astype(bool)])
    assert np.array_equal(sil, g["sil_edges"])
    cam = _cam50(g)
    index = np.zeros((len(sil), 2), np.int64); f = np.zeros(len(sil), np.float32)
    hostsim.hs_edge_sample_forward(s.h, s.v64.ctypes.data, sil.ctypes.data, len(sil), cam.ctypes.data, origin3.ctypes.data,
                                   index.ctypes.data, f.ctypes.data)
    valid_edge = np.abs(f) > 1e-5
    idx = index[valid_edge]
    keep = (idx[:, 0] < res - 1) & (idx[:, 1] < res - 1) & (idx[:, 0] >= 0) & (idx[:, 1] >= 0)
    assert np.array_equal(idx[keep], g["vh_index"])
    hit = np.zeros(res * res, dtype=np.uint8); hit[g["b1_ind"]] = 1
    soft = views.process_mask(hit.reshape(res, res))
    diff = soft[idx[:, 1].clip(0, res - 1), idx[:, 0].clip(0, res - 1)] - 0.5
    coef = np.zeros(len(sil))
    coef[np.flatnonzero(valid_edge)[keep]] = -np.sign(diff[keep])
    gv = np.zeros_like(s.v64)
    hostsim.hs_edge_sample_backward(s.v64.ctypes.data, sil.ctypes.data, len(sil), cam.ctypes.data, f.ctypes.data, coef.ctypes.data, 1, gv.ctypes.data)
    ref = g["grad_vh"]
    np.testing.assert_allclose(gv, ref, rtol=1e-8, atol=1e-10 * np.nanmax(np.abs(ref)), equal_nan=True)
    cosv = np.zeros(len(E2F)); loss = np.zeros(1); gsm = np.zeros_like(V)
    hostsim.hs_dihedral(V.ctypes.data, E2F.ctypes.data, len(E2F), 2, cosv.ctypes.data, None, loss.ctypes.data, gsm.ctypes.data)
    np.testing.assert_allclose(cosv, g["dihedral_cos"], rtol=1e-10, atol=1e-12, equal_nan=True)
    assert np.isnan(loss[0])
    ref = g["grad_sm"]
    assert np.array_equal(np.isnan(gsm), np.isnan(ref))
    np.testing.assert_allclose(gsm, ref, rtol=1e-8, atol=1e-10 * np.nanmax(np.abs(ref)), equal_nan=True)


def test_projected_primary_visibility_math(hostsim, hand):
    """drt_raster.h compiled for the host: the pinhole model fitted from four rays of a generate_ray image reproduces K^-1 / R^-1,
    every ray verifies (and perturbed ones do not), and deciding the primary hits by testing each triangle against the pixels of
    its padded projected box equals the exhaustive float32 test bit for bit -- with ~3 tests per triangle."""
    s = HostScene(hostsim, hand.faces, hand.vertices)
    c, ext = views.mesh_frame(hand.vertices)
    for (w, h, view) in ((128, 128, 5), (192, 64, 40), (64, 96, 23)):
        R, K, Rinv, Kinv = views.turntable_cameras(c, ext, 72, w, h)[view]
        o, d = views.generate_ray(h, w, Kinv, Rinv)
        on, dn = np.ascontiguousarray(o.numpy()), np.ascontiguousarray(d.numpy())
        model = np.zeros(14)
        assert hostsim.hs_fit_view(on.ctypes.data, dn.ctypes.data, w, h, model.ctypes.data) == 1 and model[12] == 1 and model[13] == 1
        assert np.array_equal(model[:3], Rinv[:3, 3])
        # minv ~ (R^-1[:3,:3] K^-1)^-1 = K R[:3,:3] up to the scale fixed by the first ray
        M = np.asarray(K) @ np.asarray(R)[:3, :3]
        minv = model[3:12].reshape(3, 3)
        np.testing.assert_allclose(minv / minv[2, 2], M / M[2, 2], rtol=1e-9, atol=1e-9)
        flags = np.zeros(w * h, np.uint8)
        hostsim.hs_verify_rays(model.ctypes.data, on.ctypes.data, dn.ctypes.data, w, h, flags.ctypes.data)
        assert flags.all()
        T = np.zeros(w * h, np.float32); ID = np.zeros(w * h, np.int32)
        tests = hostsim.hs_raster(s.h, model.ctypes.data, on.ctypes.data, dn.ctypes.data, w, h, T.ctypes.data, ID.ctypes.data)
        To, IDo = orc.trace_closest(s.f32, s.v32, _rays32(o, d))
        assert np.array_equal(ID, IDo) and np.array_equal(T, To)
        assert (IDo >= 0).sum() > 50 and 0 < tests < 8 * len(hand.faces) + 4 * (IDo >= 0).sum()
        # rays that are not the grid's: shifted by a pixel, another origin, a tiny rotation of every direction
        bad = dn.copy(); bad[:-1] = dn[1:]
        hostsim.hs_verify_rays(model.ctypes.data, on.ctypes.data, bad.ctypes.data, w, h, flags.ctypes.data)
        assert flags.mean() < 0.02
        off = on.copy(); off[:, 1] += 1e-9
        hostsim.hs_verify_rays(model.ctypes.data, off.ctypes.data, dn.ctypes.data, w, h, flags.ctypes.data)
        assert not flags.any()
        th = 3e-5                                   # ~0.03 px at these focal lengths: above the 1e-3 px tolerance
        rot = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
        hostsim.hs_verify_rays(model.ctypes.data, on.ctypes.data, np.ascontiguousarray(dn @ rot.T).ctypes.data, w, h, flags.ctypes.data)
        assert flags.mean() < 0.5
    # a "camera" inside the object: some triangle has no projection bound
    Kin = np.array([[40.0, 0, 32], [0, 40.0, 32], [0, 0, 1]])
    Rin = np.eye(4); Rin[:3, 3] = c
    o, d = views.generate_ray(64, 64, np.linalg.inv(Kin), Rin)
    on, dn = np.ascontiguousarray(o.numpy()), np.ascontiguousarray(d.numpy())
    model = np.zeros(14)
    assert hostsim.hs_fit_view(on.ctypes.data, dn.ctypes.data, 64, 64, model.ctypes.data) == 1
    T = np.zeros(64 * 64, np.float32); ID = np.zeros(64 * 64, np.int32)
    assert hostsim.hs_raster(s.h, model.ctypes.data, on.ctypes.data, dn.ctypes.data, 64, 64, T.ctypes.data, ID.ctypes.data) == -1


def test_morton_plan_splits_the_longest_cell_axis_first(hostsim):
    """drt_lbvh.h::morton_plan: every key bit halves the axis along which the cells are longest (ties: x, y, z), 30 bits in all,
    at most 20 per axis; keys order points first by the most significant split."""
    def plan(ext):
        e = np.asarray(ext, np.float32); ax = np.zeros(30, np.uint8); pos = np.zeros(30, np.uint8); bits = np.zeros(3, np.uint8)
        hostsim.hs_morton_plan(e.ctypes.data, ax.ctypes.data, pos.ctypes.data, bits.ctypes.data)
        return ax, pos, bits
    ax, pos, bits = plan([159.4, 211.8, 67.9])                     # horse_vh.ply x4
    assert list(bits) == [10, 11, 9] and list(ax[:5]) == [1, 0, 1, 0, 2]
    cell = np.array([159.4, 211.8, 67.9])
    for k in range(30):                                            # the rule itself
        assert ax[k] == int(np.argmax(cell)); cell[ax[k]] /= 2
    for a in range(3):                                             # bit positions of an axis run from its top bit down to 0
        assert list(pos[ax == a]) == list(range(bits[a] - 1, -1, -1))
    ax, _, bits = plan([1.0, 1.0, 1.0])
    assert list(bits) == [10, 10, 10] and list(ax[:6]) == [0, 1, 2, 0, 1, 2]      # a cube: the plain interleave
    _, _, bits = plan([1000.0, 1.0, 0.0])
    assert list(bits) == [20, 10, 0]                               # needle / flat boxes: capped, a zero extent gets no bit
    _, _, bits = plan([0.0, 0.0, 0.0])
    assert bits.sum() == 30 and bits.max() <= 20
    ext = np.array([4.0, 2.0, 1.0], np.float32)
    key = lambda p: hostsim.hs_morton_key(ext.ctypes.data, np.asarray(p, np.float32).ctypes.data)
    assert key([0.1, 1.9, 0.9]) < key([2.1, 0.1, 0.1])             # the first bit splits x (the longest side) ...
    assert key([0.1, 0.1, 0.9]) < key([0.1, 1.1, 0.1]) < key([1.1, 0.1, 0.1])     # ... the second x again (cells 2 x 2 x 1), then y
    assert key([3.999, 1.999, 0.999]) == (1 << 30) - 1 and key([0.0, 0.0, 0.0]) == 0
