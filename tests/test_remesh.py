"""Isotropic remeshing between passes (reference optim.py:12-52 delegates to
meshlabserver, so these are behavioural checks: closed oriented manifold of the same genus, on the input
surface, edge lengths around the target, deterministic)."""
import numpy as np
import pytest

from conftest import data_path
from drt_amd import _lib, mesh_io, remesh
from oracle import diffrender_oracle as orc


def _edge_len(m):
    e = m.edges
    return np.linalg.norm(m.vertices[e[:, 0]] - m.vertices[e[:, 1]], axis=1)


def _volume(m):
    t = m.vertices[m.faces]
    return np.einsum("ij,ij->i", t[:, 0], np.cross(t[:, 1], t[:, 2])).sum() / 6.0


def _oriented_closed(m):
    """every directed edge occurs once and its reverse occurs once"""
    e = m.edges
    key = e[:, 0] * len(m.vertices) + e[:, 1]
    rev = e[:, 1] * len(m.vertices) + e[:, 0]
    return len(np.unique(key)) == len(key) and np.array_equal(np.sort(key), np.sort(rev))


@pytest.fixture(scope="module")
def hand():
    return mesh_io.read_ply(data_path("hand_vh.ply"))


@pytest.mark.parametrize("L", [6.0, 3.0])
def test_remesh_hand(hand, L):
    out, st = remesh.isotropic_remesh(hand, L, return_stats=True)
    assert st["iterations"] == 3 and st["split"] > 0 and st["collapsed"] > 0
    assert out.is_watertight and _oriented_closed(out)
    assert len(out.vertices) - len(out.faces) // 2 == 2            # genus 0 stays genus 0 (V - E + F with E = 3F/2)
    assert out.faces.max() == len(out.vertices) - 1 and len(np.unique(out.faces)) == len(out.vertices)   # compacted
    el = _edge_len(out)
    assert ((el > 0.8 * L) & (el < 4.0 / 3.0 * L)).mean() > 0.9
    assert abs(el.mean() - L) < 0.15 * L and el.max() <= 4.0 / 3.0 * L * 1.2
    assert abs(_volume(out) / _volume(hand) - 1) < 0.03
    # the new vertices lie on the input surface (reprojection; float32 rounding of the output positions) ...
    d_new, _ = orc.point_mesh_distance(out.vertices, hand.vertices, hand.faces)
    assert d_new.max() < 1e-4
    # ... and the input stays close to the new surface (the MaxSurfDist = 1 check is one-sided, new -> input, like
    # MeshLab's; what a coarser mesh cannot represent -- the ridges of a visual hull -- is cut by a fraction of L)
    d_old, _ = orc.point_mesh_distance(hand.vertices[::7], out.vertices, out.faces)
    assert d_old.max() < 0.5 * L and d_old.mean() < 0.1 * L
    # no folded faces: neighbouring normals never oppose each other
    e2f = mesh_io.edge_tables(out)[1]
    tri = out.vertices[e2f]                                        # [E,2,3,3]
    n = np.cross(tri[:, :, 1] - tri[:, :, 0], tri[:, :, 2] - tri[:, :, 0])
    n /= np.linalg.norm(n, axis=2, keepdims=True)
    assert (n[:, 0] * n[:, 1]).sum(1).min() > -0.9


@pytest.mark.parametrize("L", [6.0, 3.0])
def test_host_remesh_meets_the_contract_of_the_references_meshlab_parameters(hand, L):
    """oracle/remesh_oracle.py: the contract of reference optim.py:17-32 (Iterations 3, TargetLen L, MaxSurfDist 1, all five steps),
    measured with numpy alone -- none of the product's mesh helpers."""
    from oracle import remesh_oracle as ro
    out = remesh.isotropic_remesh(hand, L)
    rep, bad = ro.check(hand.vertices, hand.faces, out.vertices, out.faces, L, max_surf_dist=1.0, max_samples=1500)
    assert not bad, (bad, rep)
    assert rep["topology"]["genus_sum"] == 0 and rep["topology"]["components"] == 1


def test_the_remesh_oracle_rejects_broken_outputs(hand):
    """Negative controls: a flipped face, a vertex pulled off the surface, and the untouched input (nothing like the target length)."""
    from oracle import remesh_oracle as ro
    L = 4.0
    out = remesh.isotropic_remesh(hand, L)
    F2 = out.faces.copy()
    F2[0] = F2[0][[1, 0, 2]]
    assert any("manifold" in b for b in ro.check(hand.vertices, hand.faces, out.vertices, F2, L, max_samples=300)[1])
    V2 = out.vertices.copy()
    V2[10] += 3.0
    assert any("off the input surface" in b for b in ro.check(hand.vertices, hand.faces, V2, out.faces, L, max_samples=len(V2))[1])
    bad = ro.check(hand.vertices, hand.faces, hand.vertices, hand.faces, L, max_samples=300)[1]
    assert any("in [4/5 L, 4/3 L]" in b for b in bad)
    # the distance routine of this oracle against the one the rest of the suite uses
    P = np.random.default_rng(1).standard_normal((200, 3)) * 60 + hand.vertices.mean(0)
    d2, _ = orc.point_mesh_distance(P, hand.vertices, hand.faces)
    assert np.abs(ro.distance_to_surface(P, hand.vertices, hand.faces) - np.asarray(d2)).max() < 1e-10


def test_remesh_is_deterministic_and_float32(hand):
    a = remesh.isotropic_remesh(hand, 4.0)
    b = remesh.isotropic_remesh(hand, 4.0)
    assert np.array_equal(a.vertices, b.vertices) and np.array_equal(a.faces, b.faces)
    assert np.array_equal(a.vertices, a.vertices.astype(np.float32).astype(np.float64))    # PLY round trip of the reference


def test_remesh_steps_can_be_selected(hand):
    L = 4.0
    out, st = remesh.isotropic_remesh(hand, L, iterations=4, flags=remesh.SPLIT, return_stats=True)
    assert st["collapsed"] == 0 and st["flipped"] == 0 and out.is_watertight
    assert _edge_len(out).max() <= 4.0 / 3.0 * L * (1 + 1e-6)
    assert len(out.faces) > len(hand.faces)
    np.testing.assert_array_equal(out.vertices[:len(hand.vertices)], hand.vertices)      # a split only adds midpoints
    same = remesh.isotropic_remesh(hand, 1e6, flags=remesh.SPLIT)                             # nothing is longer than that
    assert np.array_equal(same.faces, hand.faces)


def test_remesh_sphere_keeps_its_radius():
    s = mesh_io.icosphere(3, radius=50.0)
    out = remesh.isotropic_remesh(s, 6.0)
    r = np.linalg.norm(out.vertices, axis=1)
    assert out.is_watertight and _oriented_closed(out) and abs(r.mean() - 50.0) < 0.3 and r.min() > 49.0
    val = np.bincount(out.faces.reshape(-1))
    assert np.abs(val - 6).mean() < 0.8                           # flips drive valences towards 6


def test_remesh_rejects_bad_input(hand):
    open_mesh = mesh_io.TriMesh(hand.vertices, hand.faces[:-1])
    with pytest.raises(ValueError):
        remesh.isotropic_remesh(open_mesh, 4.0)
    with pytest.raises(_lib.DrtError):
        remesh.isotropic_remesh(hand, -1.0)
