"""RECYCLE_OUTPUTS (drt_amd/diffrender.py, drt_outputs_clean): a trusted-grid call renders into the dense outputs of an earlier call of
the same size once the caller has let go of them, after zeroing only the rows that call set.  What the caller sees must be what fresh,
memset outputs give: zeros wherever mask is false, bit-identical values elsewhere -- with moving vertices, with outputs the caller keeps,
with outputs the caller wrote into before dropping them, on another stream, and for calls of another size in between."""
import numpy as np
import pytest
import torch

from conftest import IOR, data_path
from drt_amd import mesh_io, views

pytestmark = pytest.mark.gpu

RES, NV = 128, 3


@pytest.fixture()
def setup():
    from drt_amd import diffrender as Render
    old = (Render.RECYCLE_OUTPUTS, Render.RECYCLE_MIN_RAYS, Render.PREFILL_NEXT)
    Render.intIOR = IOR
    Render.resx = Render.resy = RES            # (the image size the ray tensors are cut into: whole images -> projected visibility, grid verdict cache)
    mesh = mesh_io.read_ply(data_path("hand_vh.ply"))
    c, ext = views.mesh_frame(mesh.vertices)
    cams = views.turntable_cameras(c, ext, 8, RES, RES)
    rays = [views.generate_ray(RES, RES, cams[k][3], cams[k][2], device="cuda") for k in range(NV)]
    o = torch.cat([r[0] for r in rays]).contiguous()
    d = torch.cat([r[1] for r in rays]).contiguous()
    yield Render, mesh, o, d
    Render.RECYCLE_OUTPUTS, Render.RECYCLE_MIN_RAYS, Render.PREFILL_NEXT = old


def _render(Render, scene, V, o, d):
    scene.update_verticex(V)
    oo, od, mk = scene.render_transparent(o, d)
    return oo, od, mk


def _reference(Render, mesh, Vs, o, d):
    """The same sequence of calls with recycling off (fresh outputs, memset fills)."""
    Render.RECYCLE_OUTPUTS = False
    scene = Render.Scene(mesh, 0)
    out = []
    for V in Vs:
        oo, od, mk = _render(Render, scene, V, o, d)
        out.append((oo.clone(), od.clone(), mk.clone()))
    return out


def _moving(mesh, steps):
    v0 = torch.tensor(mesh.vertices, dtype=torch.float64, device="cuda")
    c = v0.mean(0)
    return [c + (v0 - c) * (1.0 + 0.04 * k) + 0.3 * k for k in range(steps)]


def test_recycled_outputs_equal_fresh_ones_with_moving_vertices(setup):
    Render, mesh, o, d = setup
    Vs = _moving(mesh, 6)
    ref = _reference(Render, mesh, Vs, o, d)
    Render.RECYCLE_OUTPUTS, Render.RECYCLE_MIN_RAYS = True, 0
    scene = Render.Scene(mesh, 0)
    ptrs = []
    for k, V in enumerate(Vs):
        oo, od, mk = _render(Render, scene, V, o, d)
        ptrs.append(oo.data_ptr())
        assert oo._version == 0 and od._version == 0 and mk._version == 0
        assert torch.equal(mk, ref[k][2]) and torch.equal(oo, ref[k][0]) and torch.equal(od, ref[k][1]), k
        assert float(oo[~mk].abs().sum()) == 0.0 and float(od[~mk].abs().sum()) == 0.0
        assert 100 < int(mk[:, 0].sum()) < mk.shape[0] // 2
        del oo, od, mk
    # call 0 establishes the grid verdict, call 1 is the first trusted one (fresh buffers, pooled), calls 2.. render into ITS memory
    pool = scene.optix_mesh._out_pool.entries
    assert len(set(ptrs[1:])) == 1 and len(pool) == 1 and pool[0][3][0].data_ptr() == ptrs[-1]


def test_outputs_the_caller_keeps_are_never_reused(setup):
    Render, mesh, o, d = setup
    Vs = _moving(mesh, 5)
    ref = _reference(Render, mesh, Vs, o, d)
    Render.RECYCLE_OUTPUTS, Render.RECYCLE_MIN_RAYS = True, 0
    scene = Render.Scene(mesh, 0)
    kept = []
    for k, V in enumerate(Vs):
        out = _render(Render, scene, V, o, d)
        kept.append(out if k != 3 else (out[0].detach().view(-1), out[1][10:20], out[2].view(torch.uint8)))     # (step 3: only aliases survive)
        if k == 3:
            del out
    torch.cuda.synchronize()
    assert len({t[0].data_ptr() for t in kept[:3]}) == 3
    for k in (0, 1, 2, 4):
        assert torch.equal(kept[k][0], ref[k][0]) and torch.equal(kept[k][1], ref[k][1]) and torch.equal(kept[k][2], ref[k][2]), k
    assert torch.equal(kept[3][0].view(-1, 3), ref[3][0]) and torch.equal(kept[3][1], ref[3][1][10:20]) and torch.equal(kept[3][2].view(torch.bool), ref[3][2])


def test_outputs_written_by_the_caller_are_not_trusted_to_be_zero(setup):
    Render, mesh, o, d = setup
    Vs = _moving(mesh, 5)
    ref = _reference(Render, mesh, Vs, o, d)
    Render.RECYCLE_OUTPUTS, Render.RECYCLE_MIN_RAYS = True, 0
    scene = Render.Scene(mesh, 0)
    for k, V in enumerate(Vs):
        oo, od, mk = _render(Render, scene, V, o, d)
        assert torch.equal(mk, ref[k][2]) and torch.equal(oo, ref[k][0]) and torch.equal(od, ref[k][1]), k
        if k == 2:
            od.add_(1.0)                         # every row non-zero now: the pool must notice (shared version counter) and drop the buffers
        if k == 3:
            oo.detach()[5:9] = 7.0               # through an alias
        del oo, od, mk


def test_another_size_and_another_stream_in_between(setup):
    Render, mesh, o, d = setup
    Vs = _moving(mesh, 6)
    ref = _reference(Render, mesh, Vs, o, d)
    half = o.shape[0] // NV
    Render.RECYCLE_OUTPUTS, Render.RECYCLE_MIN_RAYS = True, 0
    scene = Render.Scene(mesh, 0)
    side = torch.cuda.Stream()
    for k, V in enumerate(Vs):
        if k == 3:                               # one image only (another size), twice
            for _ in range(2):
                a, b, m = _render(Render, scene, V, o[:half].contiguous(), d[:half].contiguous())
                assert torch.equal(m, ref[k][2][:half]) and torch.equal(b, ref[k][1][:half])
                del a, b, m
        if k == 4:                               # a call on another stream must not take buffers whose last users ran on this one
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                oo, od, mk = _render(Render, scene, V, o, d)
            torch.cuda.current_stream().wait_stream(side)
        else:
            oo, od, mk = _render(Render, scene, V, o, d)
        torch.cuda.synchronize()
        assert torch.equal(mk, ref[k][2]) and torch.equal(oo, ref[k][0]) and torch.equal(od, ref[k][1]), k
        del oo, od, mk


def test_gradients_and_loss_through_recycled_outputs(setup):
    """The step itself (optim.full_batch_step) with and without recycling: same losses, same parameters."""
    Render, mesh, o, d = setup
    from drt_amd import optim as O
    Render.resx = Render.resy = RES
    rng = np.random.default_rng(5)
    c, _ = views.mesh_frame(mesh.vertices)
    sp = torch.tensor(rng.standard_normal((o.shape[0], 3)) * 40.0 + np.asarray(c) + np.array([0.0, 0.0, 150.0]), device="cuda")
    valid = torch.tensor(rng.random(o.shape[0]) > 0.1, device="cuda")
    res = []
    for rec in (False, True):
        Render.RECYCLE_OUTPUTS, Render.RECYCLE_MIN_RAYS = rec, 0
        scene = Render.Scene(mesh, 0)
        init_vertices, parameter, opt = O.setup_opt(scene, 0.1, O.HyperParams, hook=False, fused=True)
        losses = [O.full_batch_step(scene, [(sp, valid, o, d)], init_vertices, parameter, opt, 40 * 217.5 / RES / RES, fused=False).item() for _ in range(6)]
        res.append((np.array(losses), parameter.detach().cpu().numpy()))
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=1e-12)
    scale = np.abs(res[0][1]).max()
    assert scale > 1e-3 and np.abs(res[0][1] - res[1][1]).max() <= 1e-11 * scale


_SPLIT_SCRIPT = r"""
import sys, json, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from conftest import IOR, data_path
from drt_amd import diffrender as Render, mesh_io, optim as O, views
RES, NV = 128, 4
Render.intIOR = IOR; Render.resx = Render.resy = RES
mesh = mesh_io.read_ply(data_path("hand_vh.ply"))
c, ext = views.mesh_frame(mesh.vertices)
cams = views.turntable_cameras(c, ext, 8, RES, RES)
rays = [views.generate_ray(RES, RES, cams[k][3], cams[k][2], device="cuda") for k in range(NV)]
o = torch.cat([r[0] for r in rays]).contiguous(); d = torch.cat([r[1] for r in rays]).contiguous()
rng = np.random.default_rng(5)
sp = torch.tensor(rng.standard_normal((o.shape[0], 3)) * 40.0 + np.asarray(c) + np.array([0.0, 0.0, 150.0]), device="cuda")
valid = torch.tensor(rng.random(o.shape[0]) > 0.1, device="cuda")
scene = Render.Scene(mesh, 0)
init_vertices, parameter, opt = O.setup_opt(scene, 0.1, O.HyperParams, hook=False, fused=True)
losses, early = [], 0
orig = Render._lib.lib().drt_ray_loss_listed_grad_split
for step in range(6):
    losses.append(O.full_batch_step(scene, [(sp, valid, o, d)], init_vertices, parameter, opt, 40 * 217.5 / RES / RES, fused=False).item())
torch.cuda.synchronize()
segs = int(scene.optix_mesh._h is not None)
print(json.dumps({"losses": losses, "param": parameter.detach().cpu().numpy().tolist()}))
"""


def test_loss_started_on_the_first_sub_batch_equals_the_loss_behind_the_join(tmp_path):
    """SPLIT_LOSS / drt_ray_loss_listed_grad_split with a call cut into FOUR sub-batches on two internal streams (DRT_MIN_SUB_LOG2 small):
    the list of completed paths is written by segment and joined behind the pipelines, the loss + gradient pass over its head runs on an
    internal stream.  Same losses and parameters as with one undivided list and the loss behind the join."""
    import json, os, subprocess, sys
    from conftest import ROOT
    script = tmp_path / "split.py"
    script.write_text(_SPLIT_SCRIPT)
    out = {}
    for name, env in (("split", {"DRT_MIN_SUB_LOG2": "14", "DRT_SUB_PER_STREAM": "2", "DRT_SPLIT_LOSS_MIN_RAYS": "0", "DRT_RECYCLE_MIN_RAYS": "0"}),
                      ("plain", {"DRT_SPLIT_LOSS": "0", "DRT_STREAMS": "1", "DRT_RECYCLE_OUTPUTS": "0"})):
        p = subprocess.run([sys.executable, str(script)], cwd=ROOT, env={**os.environ, **env}, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        out[name] = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    np.testing.assert_allclose(out["split"]["losses"], out["plain"]["losses"], rtol=1e-12)
    a, b = np.array(out["split"]["param"]), np.array(out["plain"]["param"])
    assert np.abs(a).max() > 1e-3 and np.abs(a - b).max() <= 1e-11 * np.abs(a).max()


def test_an_explicit_ray_binding_recycles_by_contract_without_the_private_torch_api(setup, monkeypatch):
    """scene.bind_rays(origin, ray_dir) (diffrender.RayBinding, round 6): the handle owns its outputs and the verdict of its rays, so the
    fast path -- trusted grids, rows of the previous call zeroed instead of a dense fill -- needs neither torch._C._storage_Use_Count nor the
    identity heuristics.  Same tensors as fresh outputs, call after call, with moving vertices; loss and gradient equal the plain route's."""
    Render, mesh, o, d = setup
    Vs = _moving(mesh, 6)
    ref = _reference(Render, mesh, Vs, o, d)
    monkeypatch.delattr(torch._C, "_storage_Use_Count")
    Render.RECYCLE_OUTPUTS = False                  # (what a torch without the API gives: the module-level switch goes off at import)
    Render.cache_report(reset=True)
    scene = Render.Scene(mesh, 0)
    rng = np.random.default_rng(5)
    sp = torch.tensor(rng.standard_normal((o.shape[0], 3)) * 40.0 + np.array([0.0, 0.0, 150.0]), device="cuda")
    valid = torch.tensor(rng.random(o.shape[0]) > 0.1, device="cuda")
    handle = scene.bind_rays(o, d, sp, valid)
    ptrs, losses, grads = [], [], []
    for k, V in enumerate(Vs):
        Vg = V.clone().requires_grad_(True)
        scene.update_verticex(Vg)
        oo, od, mk = scene.render_transparent(handle)
        ptrs.append(oo.data_ptr())
        assert torch.equal(mk, ref[k][2]) and torch.equal(oo, ref[k][0]) and torch.equal(od, ref[k][1]), k
        loss = Render.ray_loss(oo, od, mk, sp, valid)
        loss.backward()
        losses.append(float(loss)); grads.append(Vg.grad.clone())
    rep = Render.cache_report()
    assert len(set(ptrs)) == 1 and rep.get("recycle_bound", 0) == 6 and rep.get("recycle_take", 0) == 0
    assert rep.get("grid_establish", 0) == 1 and rep.get("grid_trust", 0) == 5
    # the plain route (fresh outputs, identity heuristics) on a second scene: same loss and gradient
    scene2 = Render.Scene(mesh, 0)
    for k, V in enumerate(Vs):
        Vg = V.clone().requires_grad_(True)
        scene2.update_verticex(Vg)
        oo, od, mk = scene2.render_transparent(o, d)
        loss = Render.ray_loss(oo, od, mk, sp, valid)
        loss.backward()
        assert float(loss) == pytest.approx(losses[k], rel=1e-12)
        assert float((Vg.grad - grads[k]).abs().max()) <= 1e-12 * float(grads[k].abs().max())


def test_backward_through_recycled_outputs_of_a_binding_is_refused(setup):
    Render, mesh, o, d = setup
    scene = Render.Scene(mesh, 0)
    handle = scene.bind_rays(o, d)
    V = torch.tensor(mesh.vertices, dtype=torch.float64, device="cuda", requires_grad=True)
    scene.update_verticex(V)
    oo, od, mk = scene.render_transparent(handle)
    stale = (od * od).sum()                       # a loss that needs the dense gradient path (not ray_loss's stash)
    oo2, od2, mk2 = scene.render_transparent(handle)
    assert oo2.data_ptr() == oo.data_ptr()
    with pytest.raises(RuntimeError, match="recycled by a later call on the same RayBinding"):
        stale.backward()
    (od2 * od2).sum().backward()
    assert float(V.grad.abs().max()) > 0
