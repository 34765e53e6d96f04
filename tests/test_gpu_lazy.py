"""The drop-in pair silhouette_edge / primary_visibility returns LAZY stand-ins (drt_amd/diffrender.py SampleSet): the reference's own loss
expression (optim.py:78) is recognised and evaluated over the uncompacted samples -- no boolean index, no device->host round trip per
view -- and anything else a caller does with the pair sees the tensors the reference returns.  Both routes must give the same numbers."""
import numpy as np
import pytest
import torch

from conftest import IOR, data_path, golden
from drt_amd import mesh_io, views

pytestmark = pytest.mark.gpu


@pytest.fixture()
def setup():
    from drt_amd import diffrender as Render
    old = Render.LAZY_VISIBILITY
    res = 128
    Render.intIOR = IOR
    Render.resx = Render.resy = res
    g = golden("hand_smooth_sm")
    hand = mesh_io.read_ply(data_path("hand_vh.ply"))
    mesh = mesh_io.TriMesh(g["vertices"].astype(np.float64), hand.faces)
    c, ext = views.mesh_frame(mesh.vertices)
    cams = views.turntable_cameras(c, ext, 8, res, res)
    scene = Render.Scene(mesh, 0)
    rng = np.random.default_rng(2)
    soft = torch.tensor(np.round(rng.random(res * res) * 4) / 4, device="cuda")          # values in {0, .25, .5, .75, 1}: some samples sit exactly on 0.5
    yield Render, scene, cams, soft, res
    Render.LAZY_VISIBILITY = old


def _term(Render, scene, cam, soft, res, w):
    V = scene.vertices.detach().clone().requires_grad_(True)
    scene.update_verticex(V)
    camera_M = tuple(torch.tensor(np.asarray(a), dtype=torch.float64, device="cuda") for a in cam)
    eye = torch.tensor(np.asarray(cam[2])[:3, 3].copy(), dtype=torch.float64, device="cuda")
    edges = scene.silhouette_edge(eye)
    index, output = scene.primary_visibility(edges, camera_M, eye, detach_depth=True)
    # the reference's expression, verbatim (optim.py:78)
    term = (soft.view((res, res))[index[:, 1], index[:, 0]] - output).abs().sum()
    (w * term).backward()
    return float(term.detach()), V.grad.clone(), index, output


def test_reference_expression_is_evaluated_in_place_and_equals_the_materialised_route(setup):
    Render, scene, cams, soft, res = setup
    w = 2e-3 * 217.5 / res
    for k in (1, 4, 6):
        Render.LAZY_VISIBILITY = True
        Render.cache_report(reset=True)
        t_lazy, g_lazy, index, output = _term(Render, scene, cams[k], soft, res, w)
        rep = Render.cache_report()
        assert rep.get("visibility_term_in_place", 0) == 1 and rep.get("visibility_materialised", 0) == 0, rep
        Render.LAZY_VISIBILITY = False
        t_ref, g_ref, index_r, output_r = _term(Render, scene, cams[k], soft, res, w)
        assert isinstance(index_r, torch.Tensor) and not isinstance(index, torch.Tensor)
        assert t_lazy == pytest.approx(t_ref, rel=1e-13) and t_ref > 0
        scale = float(g_ref.abs().max())
        assert scale > 0 and float((g_lazy - g_ref).abs().max()) <= 1e-13 * scale
        # ... and looking at the lazy pair afterwards gives the reference's tensors
        assert torch.equal(index.tensor(), index_r) and torch.equal(output.tensor(), output_r)
        assert index.shape == index_r.shape and output.dtype == torch.float32 and len(output) == len(output_r)


def test_any_other_use_of_the_pair_sees_the_tensors(setup):
    Render, scene, cams, soft, res = setup
    Render.LAZY_VISIBILITY = True
    camera_M = tuple(torch.tensor(np.asarray(a), dtype=torch.float64, device="cuda") for a in cams[3])
    eye = torch.tensor(np.asarray(cams[3][2])[:3, 3].copy(), dtype=torch.float64, device="cuda")
    V = scene.vertices.detach().clone().requires_grad_(True)
    scene.update_verticex(V)
    index, output = scene.primary_visibility(scene.silhouette_edge(eye), camera_M, eye, detach_depth=True)
    img = soft.view(res, res)
    # a different expression: squared difference, columns swapped on purpose, an offset -- nothing the lazy objects recognise
    a = ((img[index[:, 0], index[:, 1]] - output) ** 2).sum()
    b = (img[index[:, 1], index[:, 0]] - 2.0 * output).abs().sum()
    c = (img[index[:, 1], index[:, 0]] - output).abs().mean()
    (a + b + c).backward()
    got = (float(a), float(b), float(c), V.grad.clone())
    Render.LAZY_VISIBILITY = False
    V2 = scene.vertices.detach().clone().requires_grad_(True)
    scene.update_verticex(V2)
    index, output = scene.primary_visibility(scene.silhouette_edge(eye), camera_M, eye, detach_depth=True)
    a = ((img[index[:, 0], index[:, 1]] - output) ** 2).sum()
    b = (img[index[:, 1], index[:, 0]] - 2.0 * output).abs().sum()
    c = (img[index[:, 1], index[:, 0]] - output).abs().mean()
    (a + b + c).backward()
    assert got[:3] == pytest.approx((float(a), float(b), float(c)), rel=1e-13)
    assert float((got[3] - V2.grad).abs().max()) <= 1e-13 * float(V2.grad.abs().max())
