"""Topology on the device (reference DiffRender.py:303-317, 338-355): the unique-edge tables of Scene.init_edge by
a radix sort in libdrt_hip against the golden tables made by the reference (hand_topology.npz) and against the host tables of
mesh_io on every mesh of data/; the watertightness check; the 1 -> 4 midpoint refinement against mesh_io.subdivide_midpoint;
and the cost of a 184 090-face topology rebuild."""
import time

import numpy as np
import pytest
import torch

from conftest import IOR, data_path, golden
from drt_amd import mesh_io, views

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Render():
    from drt_amd import diffrender
    diffrender.intIOR = IOR
    return diffrender


def _tables(Render, mesh):
    F = torch.tensor(mesh.faces, dtype=torch.long, device="cuda")
    V = torch.tensor(mesh.vertices, dtype=torch.float64, device="cuda")
    return Render.edge_tables(F, V, want_rows=True)


def test_edge_tables_equal_the_reference_golden(Render):
    g = golden("hand_topology")
    hand = mesh_io.read_ply(data_path("hand_vh.ply"))
    Edges, E2F, mean_len, rows = _tables(Render, hand)
    assert Edges.dtype == torch.long and E2F.dtype == torch.long and E2F.shape == (len(g["Edges"]), 2, 3)
    assert np.array_equal(Edges.cpu().numpy(), g["Edges"]) and np.array_equal(E2F.cpu().numpy(), g["E2F"])
    assert mean_len == pytest.approx(float(g["mean_len"]), rel=1e-13)
    # row2edge: directed edge 3f+j = (F[f][j], F[f][(j+1)%3]) belongs to unique edge rows[3f+j]
    d = hand.faces[:, [0, 1, 1, 2, 2, 0]].reshape(-1, 2)
    assert np.array_equal(np.sort(d, axis=1), g["Edges"][rows.cpu().numpy()])


@pytest.mark.parametrize("name", ["mouse_vh", "horse_vh", "monkey_vh", "horse_scan"])
def test_edge_tables_equal_the_host_tables(Render, name):
    mesh = mesh_io.read_ply(data_path(name + ".ply"))
    Edges, E2F, mean_len, _ = _tables(Render, mesh)
    e, e2f, ml = mesh_io.edge_tables(mesh)
    assert np.array_equal(Edges.cpu().numpy(), e) and np.array_equal(E2F.cpu().numpy(), e2f)
    assert mean_len == pytest.approx(ml, rel=1e-12)


def test_non_watertight_meshes_are_refused(Render):
    hand = mesh_io.read_ply(data_path("hand_vh.ply"))
    V = torch.tensor(hand.vertices, dtype=torch.float64, device="cuda")
    with pytest.raises(AssertionError, match="watertight"):                     # an odd number of directed edges
        Render.edge_tables(torch.tensor(hand.faces[:-1], device="cuda"), V)
    with pytest.raises(AssertionError, match="watertight"):                     # even, but two faces missing: open boundary
        Render.edge_tables(torch.tensor(hand.faces[:-2], device="cuda"), V)
    dup = np.concatenate([hand.faces, hand.faces[:2]])                          # an edge shared by four faces
    with pytest.raises(AssertionError, match="watertight"):
        Render.edge_tables(torch.tensor(dup, device="cuda"), V)
    with pytest.raises(AssertionError):
        Render.Scene(mesh_io.TriMesh(hand.vertices, hand.faces[:-2]), 0)        # Scene.update_mesh asserts like DiffRender.py:305


def test_midpoint_refinement_on_the_device(Render):
    """Scene.subdivide_midpoint == mesh_io.subdivide_midpoint (vertices bit for bit, faces, tables), and the refined scene
    traces like a scene built from the host-refined mesh."""
    horse = mesh_io.read_ply(data_path("horse_vh.ply"))
    ref = mesh_io.subdivide_midpoint(horse)
    scene = Render.Scene(horse, 0)
    scene.subdivide_midpoint()
    assert scene.vertices.shape == (len(ref.vertices), 3) and scene.faces.shape == (len(ref.faces), 3) == (50248, 3)
    assert np.array_equal(scene.vertices.cpu().numpy(), ref.vertices) and np.array_equal(scene.faces.cpu().numpy(), ref.faces)
    e, e2f, ml = mesh_io.edge_tables(ref)
    assert np.array_equal(scene.Edges.cpu().numpy(), e) and np.array_equal(scene.E2F.cpu().numpy(), e2f) and scene.mean_len == pytest.approx(ml, rel=1e-12)
    m = scene.mesh                                                              # host record, synced lazily
    assert m.is_watertight and np.array_equal(m.faces, ref.faces) and np.array_equal(m.vertices, ref.vertices)
    other = Render.Scene(ref, 0)
    c, ext = views.mesh_frame(ref.vertices)
    cam = views.turntable_cameras(c, ext, 72, 256, 256)[17]
    o, d = views.generate_ray(256, 256, cam[3], cam[2], device="cuda")
    Render.resx = Render.resy = 256
    with torch.no_grad():
        a = scene.render_transparent(o, d)
        b = other.render_transparent(o, d)
    assert all(torch.equal(x, y) for x, y in zip(a, b)) and a[2][:, 0].float().mean().item() > 0.003
    scene.subdivide_midpoint()                                                  # twice: 200 992 faces
    assert scene.faces.shape[0] == 4 * 50248 and scene.optix_mesh.check()[0] == 0
    # an optimisation step still works on the refined mesh (gradients reach the new vertices)
    V = scene.vertices.detach().clone().requires_grad_(True)
    scene.update_verticex(V)
    oo, od, mk = scene.render_transparent(o, d)
    (od.sum() + scene.sm_loss_fused() * 0).backward()
    assert torch.isfinite(V.grad).all() and (V.grad.abs().sum(dim=1) > 0).sum().item() > 100


def test_topology_rebuild_cost_at_184k_faces(Render):
    """monkey_vh (184 090 faces, 276 135 edges): edge tables + LBVH in well under the per-pass remesh cost; the edge tables
    alone at most 1 ms of device time."""
    mesh = mesh_io.read_ply(data_path("monkey_vh.ply"))
    F = torch.tensor(mesh.faces, dtype=torch.long, device="cuda")
    V = torch.tensor(mesh.vertices, dtype=torch.float64, device="cuda")
    Render.edge_tables(F, V)                      # warm-up (code objects, allocator)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    times = []
    for _ in range(5):
        torch.cuda.synchronize()
        a.record()
        Render.edge_tables(F, V)
        b.record()
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b))
    print(f"edge tables of 184 090 faces: {min(times):.3f} ms (device, incl. the one status copy)")
    assert min(times) <= 1.0, times
