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
        assert type(index_r) is torch.Tensor and type(index) is Render.LazyIndex and isinstance(index, torch.Tensor) and torch.is_tensor(output)
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


def _cam(cam):
    camera_M = tuple(torch.tensor(np.asarray(a), dtype=torch.float64, device="cuda") for a in cam)
    return camera_M, torch.tensor(np.asarray(cam[2])[:3, 3].copy(), dtype=torch.float64, device="cuda")


def _summed(Render, scene, cams, soft, res, ids, lazy):
    """The silhouette loss of several views accumulated exactly like the reference does (optim.py:71-80)."""
    Render.LAZY_VISIBILITY = lazy
    V = scene.vertices.detach().clone().requires_grad_(True)
    scene.update_verticex(V)
    vh_loss = 0
    for k in ids:
        camera_M, eye = _cam(cams[k])
        silhouette_edge = scene.silhouette_edge(eye)
        index, output = scene.primary_visibility(silhouette_edge, camera_M, eye, detach_depth=True)
        vh_loss += (soft.view((res, res))[index[:, 1], index[:, 0]] - output).abs().sum()
    LOSS = 2e-3 * 217.5 / res * vh_loss
    text = f"vh={vh_loss:g}"
    LOSS.backward()
    return float(vh_loss), text, V.grad.clone()


def test_the_summed_terms_of_all_views_are_one_fused_launch(setup):
    """Round 6: nothing is enqueued per view -- no flag kernel, no probe rays, no term -- until the SUM of the views' terms is needed; then
    one drt_vh_loss_fused call serves all of them.  Same value and gradient as the reference's tensors view by view."""
    Render, scene, cams, soft, res = setup
    ids = list(range(8))
    Render.cache_report(reset=True)
    v_lazy, text, g_lazy = _summed(Render, scene, cams, soft, res, ids, True)
    rep = Render.cache_report()
    assert rep.get("visibility_terms_fused", 0) == 8 and rep.get("visibility_term_in_place", 0) == 0 and rep.get("visibility_materialised", 0) == 0, rep
    v_ref, text_ref, g_ref = _summed(Render, scene, cams, soft, res, ids, False)
    assert v_lazy == pytest.approx(v_ref, rel=1e-13) and v_ref > 0 and text == text_ref
    scale = float(g_ref.abs().max())
    assert scale > 0 and float((g_lazy - g_ref).abs().max()) <= 1e-13 * scale


def test_two_terms_on_one_sample_set_keep_their_own_gradients(setup):
    """Two images (say two soft masks) on the SAME (index, output) pair: each term's d / d output is its own (round 5 kept it in one buffer
    of the sample set, so the first term's backward saw the second image's signs)."""
    Render, scene, cams, soft, res = setup
    rng = np.random.default_rng(7)
    soft2 = torch.tensor(np.round(rng.random(res * res) * 4) / 4, device="cuda")
    camera_M, eye = _cam(cams[2])
    out = []
    for lazy in (True, False):
        Render.LAZY_VISIBILITY = lazy
        V = scene.vertices.detach().clone().requires_grad_(True)
        scene.update_verticex(V)
        index, output = scene.primary_visibility(scene.silhouette_edge(eye), camera_M, eye, detach_depth=True)
        t1 = (soft.view((res, res))[index[:, 1], index[:, 0]] - output).abs().sum()
        t2 = (soft2.view((res, res))[index[:, 1], index[:, 0]] - output).abs().sum()
        (t1 + 3.0 * t2).backward()
        out.append((float(t1), float(t2), V.grad.clone()))
    assert out[0][0] == pytest.approx(out[1][0], rel=1e-13) and out[0][1] == pytest.approx(out[1][1], rel=1e-13) and out[0][0] != out[0][1]
    assert float((out[0][2] - out[1][2]).abs().max()) <= 1e-13 * float(out[1][2].abs().max())


def test_a_lazy_pair_cannot_outlive_the_mesh_state_it_was_made_on(setup):
    Render, scene, cams, soft, res = setup
    Render.LAZY_VISIBILITY = True
    camera_M, eye = _cam(cams[1])
    index, output = scene.primary_visibility(scene.silhouette_edge(eye), camera_M, eye, detach_depth=True)
    scene.update_verticex(scene.vertices.detach() * 1.001)
    with pytest.raises(RuntimeError, match="after the scene's vertices / mesh changed"):
        index.shape
    index2, output2 = scene.primary_visibility(scene.silhouette_edge(eye), camera_M, eye, detach_depth=True)
    index2[0] = torch.tensor([1, 2], device="cuda")                      # (assignment through the stand-in, like on the reference's tensor)
    assert index2[0].tolist() == [1, 2] and index2._version >= 1 and len(output2) == len(index2)
