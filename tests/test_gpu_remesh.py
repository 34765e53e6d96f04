"""The remesher on the device (drt_amd/remesh_gpu.py, csrc/drt_remesh_gpu.hip) against (i) oracle/remesh_oracle.py -- the contract of the reference's
MeshLab parameter set (optim.py:17-32), measured with numpy alone -- and (ii) the sequential host version
(drt_amd/remesh.py, csrc/drt_remesh.cpp): same invariants -- closed oriented manifold of the same genus, on the input surface, edge
lengths concentrated around the target, no folds, deterministic -- and, statistically, the same mesh: face count, edge-length
histogram, valence distribution, two-sided distance between the two results.  (Neither can be held against MeshLab, reference
optim.py:12-52: it is an external program that is not here -- SURVEY section 8f row 1.)"""
import numpy as np
import pytest
import torch

from conftest import IOR, data_path
from drt_amd import mesh_io, remesh
from oracle import diffrender_oracle as orc

pytestmark = pytest.mark.gpu


def _edge_len(m):
    e = m.edges
    return np.linalg.norm(m.vertices[e[:, 0]] - m.vertices[e[:, 1]], axis=1)


def _volume(m):
    t = m.vertices[m.faces]
    return np.einsum("ij,ij->i", t[:, 0], np.cross(t[:, 1], t[:, 2])).sum() / 6.0


def _oriented_closed(m):
    e = m.edges
    key = e[:, 0] * len(m.vertices) + e[:, 1]
    rev = e[:, 1] * len(m.vertices) + e[:, 0]
    return len(np.unique(key)) == len(key) and np.array_equal(np.sort(key), np.sort(rev))


def _gpu_remesh(mesh, L, **kw):
    from drt_amd import remesh_gpu
    from drt_amd.optix_mesh import optix_mesh
    surf = optix_mesh(0)
    surf.update_mesh(torch.tensor(mesh.faces, dtype=torch.int32, device="cuda"), torch.tensor(mesh.vertices, dtype=torch.float32, device="cuda"))
    V, F, st = remesh_gpu.isotropic_remesh_gpu(torch.tensor(mesh.vertices, dtype=torch.float64, device="cuda"), torch.tensor(mesh.faces, device="cuda"),
                                               L, surface=surf, return_stats=True, **kw)
    out = mesh_io.TriMesh(V.to(torch.float32).to(torch.float64).cpu().numpy(), F.cpu().numpy())
    return out, st


@pytest.fixture(scope="module")
def hand():
    return mesh_io.read_ply(data_path("hand_vh.ply"))


@pytest.mark.parametrize("L", [6.0, 3.0])
def test_device_remesh_has_the_host_versions_invariants_and_statistics(hand, L):
    dev, st = _gpu_remesh(hand, L)
    host, st_h = remesh.isotropic_remesh(hand, L, return_stats=True)
    assert st["iterations"] == 3 and st["split"] > 0 and st["collapsed"] > 0 and st["flipped"] > 0
    # ---- invariants (those of tests/test_remesh.py)
    assert dev.is_watertight and _oriented_closed(dev)
    assert len(dev.vertices) - len(dev.faces) // 2 == 2                       # genus 0 stays genus 0
    assert dev.faces.max() == len(dev.vertices) - 1 and len(np.unique(dev.faces)) == len(dev.vertices)      # compacted
    el, el_h = _edge_len(dev), _edge_len(host)
    assert ((el > 0.8 * L) & (el < 4.0 / 3.0 * L)).mean() > 0.9
    assert abs(el.mean() - L) < 0.15 * L and el.max() <= 4.0 / 3.0 * L * 1.2
    assert abs(_volume(dev) / _volume(hand) - 1) < 0.03
    d_new, _ = orc.point_mesh_distance(dev.vertices, hand.vertices, hand.faces)
    assert d_new.max() < 1e-3                                                 # on the input surface (float32 tracer vertices, float32 output)
    d_old, _ = orc.point_mesh_distance(hand.vertices[::7], dev.vertices, dev.faces)
    assert d_old.max() < 0.6 * L and d_old.mean() < 0.1 * L                  # (ridges of the visual hull that an L-long edge cannot represent are cut)
    e2f = mesh_io.edge_tables(dev)[1]
    tri = dev.vertices[e2f]
    n = np.cross(tri[:, :, 1] - tri[:, :, 0], tri[:, :, 2] - tri[:, :, 0])
    n /= np.linalg.norm(n, axis=2, keepdims=True)
    assert (n[:, 0] * n[:, 1]).sum(1).min() > -0.9                            # no folded pairs
    # ---- the contract of the reference's MeshLab parameter set, measured by the independent oracle (numpy only, no product helpers)
    from oracle import remesh_oracle as ro
    rep, bad = ro.check(hand.vertices, hand.faces, dev.vertices, dev.faces, L, max_surf_dist=1.0, max_samples=1500)
    assert not bad, (bad, rep)
    # ---- the same mesh as the host version's, statistically
    assert abs(len(dev.faces) / len(host.faces) - 1) < 0.03, (len(dev.faces), len(host.faces))
    bins = np.linspace(0.5 * L, 1.6 * L, 12)
    h_d = np.histogram(el, bins)[0] / len(el)
    h_h = np.histogram(el_h, bins)[0] / len(el_h)
    assert np.abs(h_d - h_h).max() < 0.05, (h_d, h_h)                          # edge-length histograms within five points per bin
    assert abs(el.mean() - el_h.mean()) < 0.03 * L and abs(el.std() - el_h.std()) < 0.03 * L
    val_d, val_h = np.bincount(dev.faces.reshape(-1)), np.bincount(host.faces.reshape(-1))
    assert abs(np.abs(val_d - 6).mean() - np.abs(val_h - 6).mean()) < 0.15
    d_dh, _ = orc.point_mesh_distance(dev.vertices[::3], host.vertices, host.faces)      # two-sided distance between the two results
    d_hd, _ = orc.point_mesh_distance(host.vertices[::3], dev.vertices, dev.faces)
    # (the vertices of both lie ON the input surface; their FACES cut its curvature by the chord error of an L-long edge, ~L^2 / 8R -- two
    # different tessellations are that far from each other, not closer: the same bound as for input -> result above)
    assert max(d_dh.mean(), d_hd.mean()) < 0.06 * L and max(d_dh.max(), d_hd.max()) < 0.5 * L
    d_in_h, _ = orc.point_mesh_distance(hand.vertices[::7], host.vertices, host.faces)
    assert abs(d_old.mean() - d_in_h.mean()) < 0.02 * L                        # and they approximate the input equally well


def test_device_remesh_is_deterministic(hand):
    a, _ = _gpu_remesh(hand, 4.0)
    b, _ = _gpu_remesh(hand, 4.0)
    assert np.array_equal(a.vertices, b.vertices) and np.array_equal(a.faces, b.faces)


def test_device_split_alone_equals_the_host_split(hand):
    """The refine step has no order dependence: same vertices (the input's, then the midpoints of the long edges in ascending edge
    order vs. in face order) and, up to that renumbering, the same faces -- compared as a set of triangles over vertex POSITIONS."""
    L = 4.0
    dev, st = _gpu_remesh(hand, L, iterations=1, flags=remesh.SPLIT)
    host, st_h = remesh.isotropic_remesh(hand, L, iterations=1, flags=remesh.SPLIT, return_stats=True)
    assert st["split"] == st_h["split"] and len(dev.faces) == len(host.faces) and len(dev.vertices) == len(host.vertices)
    np.testing.assert_array_equal(dev.vertices[:len(hand.vertices)], host.vertices[:len(hand.vertices)])

    def canon(m):
        t = m.vertices[m.faces].reshape(len(m.faces), 9)
        # rotate every triangle so that its lexicographically smallest corner comes first (orientation kept)
        out = []
        for tri in t.reshape(-1, 3, 3):
            k = min(range(3), key=lambda j: tuple(tri[j]))
            out.append(np.concatenate([tri[k], tri[(k + 1) % 3], tri[(k + 2) % 3]]))
        out = np.array(out)
        return out[np.lexsort(out.T[::-1])]
    np.testing.assert_array_equal(canon(dev), canon(host))


def test_device_remesh_on_a_sphere_and_inside_the_loop():
    from drt_amd import diffrender as Render, remesh_gpu
    s = mesh_io.icosphere(3, radius=50.0)
    out, _ = _gpu_remesh(s, 6.0)
    r = np.linalg.norm(out.vertices, axis=1)
    assert out.is_watertight and _oriented_closed(out) and abs(r.mean() - 50.0) < 0.3 and r.min() > 49.0
    assert np.abs(np.bincount(out.faces.reshape(-1)) - 6).mean() < 0.8        # flips drive valences towards 6
    # as the remesh step of the loop: the scene's topology is replaced on the device, renders go on
    Render.intIOR = IOR
    Render.resx = Render.resy = 64
    scene = Render.Scene(mesh_io.read_ply(data_path("hand_vh.ply")), 0)
    n0 = scene.faces.shape[0]
    remesh_gpu.GpuMeshlabserver().remesh(scene, 5.0)
    assert scene.faces.shape[0] != n0 and scene.E2F.shape[0] == 3 * scene.faces.shape[0] // 2
    bad, _ = scene.optix_mesh.check()
    assert bad == 0
    m = scene.mesh
    assert m.is_watertight and len(m.faces) == scene.faces.shape[0]


def _torus(R=40.0, r=12.0, nu=96, nv=40):
    u = np.linspace(0, 2 * np.pi, nu, endpoint=False)
    v = np.linspace(0, 2 * np.pi, nv, endpoint=False)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    V = np.stack([(R + r * np.cos(vv)) * np.cos(uu), (R + r * np.cos(vv)) * np.sin(uu), r * np.sin(vv)], -1).reshape(-1, 3)
    idx = lambda i, j: (i % nu) * nv + (j % nv)
    F = []
    for i in range(nu):
        for j in range(nv):
            a, b, c, d = idx(i, j), idx(i + 1, j), idx(i + 1, j + 1), idx(i, j + 1)
            F += [[a, b, c], [a, c, d]]
    return mesh_io.TriMesh(V.astype(np.float32).astype(np.float64), np.array(F, dtype=np.int64))


@pytest.mark.parametrize("L", [5.0, 2.0])
def test_device_remesh_keeps_the_genus_of_a_torus(L):
    """A handle must survive: the link condition of the collapses and the existing-edge test of the flips are what keeps a genus-1 surface
    a genus-1 manifold (V - E + F = 0); coarsening (L = 5, from 2.6 / 1.9 mm edges) and refining (L = 2 splits the long diagonals) alike."""
    t = _torus()
    assert t.is_watertight and len(t.vertices) - len(t.faces) // 2 == 0
    dev, st = _gpu_remesh(t, L)
    host = remesh.isotropic_remesh(t, L)
    assert dev.is_watertight and _oriented_closed(dev)
    assert len(dev.vertices) - len(dev.faces) // 2 == 0 and len(host.vertices) - len(host.faces) // 2 == 0
    el = _edge_len(dev)
    assert ((el > 0.8 * L) & (el < 4.0 / 3.0 * L)).mean() > 0.9
    assert abs(len(dev.faces) / len(host.faces) - 1) < 0.05
    rho = np.hypot(np.hypot(dev.vertices[:, 0], dev.vertices[:, 1]) - 40.0, dev.vertices[:, 2])
    assert abs(rho.mean() - 12.0) < 0.15 and rho.min() > 11.0 and rho.max() < 12.2       # still the tube of radius 12 (vertices on the input polyhedron)
    assert abs(_volume(dev) / _volume(t) - 1) < 0.03



@pytest.mark.parametrize("seed", range(6))
def test_device_remesh_on_random_shapes(seed):
    """Random closed shapes (tests/test_gpu_fuzz.py::_shape: noisy stretched icospheres, some with a second component inside or beside
    them), coarsened and refined: a closed oriented manifold comes out, with the input's Euler characteristic (2 per component), on the
    input surface, without folds or degenerate faces, edge lengths around the target -- and the host version agrees on the size."""
    from test_gpu_fuzz import _shape
    rng = np.random.default_rng(100 + seed)
    mesh = _shape(rng, noises=(0.0, 0.03, 0.06))          # (bumps well below the edge length: what a visual hull or an optimised mesh looks like)
    chi = len(mesh.vertices) - len(mesh.faces) // 2
    assert chi in (2, 4)
    mean_len = float(_edge_len(mesh).mean())
    for factor in (1.6, 0.6):
        L = factor * mean_len
        dev, st = _gpu_remesh(mesh, L)
        assert dev.is_watertight and _oriented_closed(dev)
        assert len(dev.vertices) - len(dev.faces) // 2 == chi
        assert len(np.unique(dev.faces)) == len(dev.vertices)
        tri = dev.vertices[dev.faces]
        area = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
        assert area.min() > 1e-6 * L * L
        el = _edge_len(dev)
        assert el.max() <= 4.0 / 3.0 * L * 1.25
        if factor < 1:        # (coarsening a bumpy surface stops where a collapse would leave the input surface by more than MaxSurfDist)
            assert ((el > 0.8 * L) & (el < 4.0 / 3.0 * L)).mean() > 0.7, ((el > 0.8 * L) & (el < 4.0 / 3.0 * L)).mean()
        d_new, _ = orc.point_mesh_distance(dev.vertices, mesh.vertices, mesh.faces)
        # on the input surface -- but for the few vertices whose projection would have folded a face: those stay where the relaxation put them
        assert np.quantile(d_new, 0.97) < 1e-3 * max(1.0, np.abs(mesh.vertices).max() / 50.0) and d_new.max() < 0.25 * L, (np.quantile(d_new, 0.97), d_new.max(), L)
        e2f = mesh_io.edge_tables(dev)[1]
        t2 = dev.vertices[e2f]
        n = np.cross(t2[:, :, 1] - t2[:, :, 0], t2[:, :, 2] - t2[:, :, 0])
        n /= np.linalg.norm(n, axis=2, keepdims=True)
        assert (n[:, 0] * n[:, 1]).sum(1).min() > -0.95
        host = remesh.isotropic_remesh(mesh, L)
        assert abs(len(dev.faces) / len(host.faces) - 1) < 0.08, (len(dev.faces), len(host.faces))
        # (chords cut corners: an 80-face shape coarsened loses 15 % of its volume -- in both versions alike)
        assert abs(_volume(dev) / _volume(mesh) - 1) < 0.25 and abs(_volume(dev) / _volume(host) - 1) < 0.03, (_volume(dev) / _volume(mesh), _volume(host) / _volume(mesh))


def test_the_device_ends_a_step_and_the_batch_size_does_not_matter(hand, monkeypatch):
    """The rounds of a collapse / flip step are enqueued ROUND_BATCH at a time and the DEVICE decides when the step is over
    (drt_rm_round_end): rounds of a batch that come after the end are no-ops, so any batch size gives the same mesh, the same number of
    rounds that ran, and a handful of host read-backs per step instead of one per round."""
    from drt_amd import remesh_gpu
    results = {}
    for batch in (1, 3, 4, 7):
        monkeypatch.setattr(remesh_gpu, "ROUND_BATCH", batch)
        trips = {"n": 0}
        for meth in ("item", "tolist"):
            orig = getattr(torch.Tensor, meth)

            def counted(self, *a, _orig=orig, **k):
                trips["n"] += 1
                return _orig(self, *a, **k)
            monkeypatch.setattr(torch.Tensor, meth, counted)
        m, st = _gpu_remesh(hand, 4.0)
        monkeypatch.undo()
        results[batch] = (m, st, trips["n"])
        assert st["collapse_unfinished"] == 0 and st["flip_unfinished"] == 0
    ref, st_ref, trips_1 = results[1]
    for batch, (m, st, trips) in results.items():
        assert np.array_equal(m.vertices, ref.vertices) and np.array_equal(m.faces, ref.faces), batch
        assert st == st_ref, (batch, st, st_ref)
    assert results[4][2] < trips_1 and results[4][2] <= 3 * (3 + 3 * 2 + 2 + 2 * 3) + 2, (results[4][2], trips_1)


def test_device_remesh_of_an_input_with_unused_vertices(hand):
    """compact() sizes its output from the number of collapses (no read-back): vertices no face of the INPUT uses are counted once, up front."""
    from drt_amd import remesh_gpu
    from drt_amd.optix_mesh import optix_mesh
    V = np.concatenate([hand.vertices, hand.vertices[:7] + 1000.0])           # seven stray vertices behind the mesh
    surf = optix_mesh(0)
    surf.update_mesh(torch.tensor(hand.faces, dtype=torch.int32, device="cuda"), torch.tensor(V, dtype=torch.float32, device="cuda"))
    Vo, Fo = remesh_gpu.isotropic_remesh_gpu(torch.tensor(V, dtype=torch.float64, device="cuda"), torch.tensor(hand.faces, device="cuda"), 4.0, surface=surf)
    ref, _ = _gpu_remesh(hand, 4.0)
    assert int(Fo.max()) == len(Vo) - 1 and len(torch.unique(Fo)) == len(Vo)
    assert np.array_equal(Vo.cpu().numpy().astype(np.float32), ref.vertices.astype(np.float32)) and np.array_equal(Fo.cpu().numpy(), ref.faces)
