"""The list of completed paths of a call cut into sub-batches (drt_render_forward: one segment per sub-batch, k_join_lists closes the gaps
behind the join) when MORE THAN HALF of the rays complete -- the object fills the frame, the segments' final places overlap their
original ones -- and what a failed call leaves behind (include/drt_hip.h: drt_outputs_cancel)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import IOR, ROOT

pytestmark = pytest.mark.gpu

_SCRIPT = r"""
import sys, json, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from conftest import IOR
from drt_amd import diffrender as Render, mesh_io, views
RES, NV = 128, 6
Render.intIOR = IOR; Render.resx = Render.resy = RES
# an icosphere-like ball (octahedron subdivided 4 times, pushed onto the unit sphere, radius 50): every ray that enters it leaves it
V = np.array([[1,0,0],[-1,0,0],[0,1,0],[0,-1,0],[0,0,1],[0,0,-1]], dtype=np.float64)
F = np.array([[0,2,4],[2,1,4],[1,3,4],[3,0,4],[2,0,5],[1,2,5],[3,1,5],[0,3,5]])
m = mesh_io.TriMesh(V, F)
for _ in range(4):
    m = mesh_io.subdivide_midpoint(m, float32_positions=False)
    m = mesh_io.TriMesh(m.vertices / np.linalg.norm(m.vertices, axis=1, keepdims=True), m.faces)
m = mesh_io.TriMesh((50.0 * m.vertices).astype(np.float32).astype(np.float64), m.faces)
cams = views.turntable_cameras(np.zeros(3), 100.0, 8, RES, RES, distance_factor=0.8)       # the ball covers the whole image
rays = [views.generate_ray(RES, RES, cams[k][3], cams[k][2], device="cuda") for k in range(NV)]
o = torch.cat([r[0] for r in rays]).contiguous(); d = torch.cat([r[1] for r in rays]).contiguous()
rng = np.random.default_rng(5)
sp = torch.tensor(rng.standard_normal((o.shape[0], 3)) * 40.0 + np.array([0.0, 0.0, 150.0]), device="cuda")
valid = torch.tensor(rng.random(o.shape[0]) > 0.1, device="cuda")
scene = Render.Scene(m, 0)
out = []
for step in range(4):
    Vt = torch.tensor(m.vertices * (1.0 + 0.01 * step), device="cuda", requires_grad=True)
    scene.update_verticex(Vt)
    oo, od, mk = scene.render_transparent(o, d)
    paths, n_paths = od._drt_link.paths
    n = int(n_paths.item())
    listed = torch.sort(paths[:n].long()).values
    expect = torch.nonzero(mk[:, 0]).flatten()
    assert torch.equal(listed, expect), (step, n, int(expect.numel()))
    loss = Render.ray_loss(oo, od, mk, sp, valid)
    loss.backward()
    out.append({"n": n, "frac": n / o.shape[0], "loss": loss.item(), "gsum": Vt.grad.abs().sum().item(), "gmax": Vt.grad.abs().max().item()})
print(json.dumps(out))
"""


def test_segments_that_overlap_their_final_place_are_joined_correctly(tmp_path):
    script = tmp_path / "join.py"
    script.write_text(_SCRIPT)
    res = {}
    for name, env in (("split", {"DRT_MIN_SUB_LOG2": "13", "DRT_SUB_PER_STREAM": "3", "DRT_SPLIT_LOSS_MIN_RAYS": "0", "DRT_RECYCLE_MIN_RAYS": "0"}),
                      ("plain", {"DRT_SPLIT_LOSS": "0", "DRT_STREAMS": "1", "DRT_RECYCLE_OUTPUTS": "0"})):
        p = subprocess.run([sys.executable, str(script)], cwd=ROOT, env={**os.environ, **env}, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        res[name] = json.loads([l for l in p.stdout.splitlines() if l.startswith("[")][-1])
    for a, b in zip(res["split"], res["plain"]):
        assert a["n"] == b["n"] and a["frac"] > 0.5, a          # more than half of ALL rays complete: every segment but the first overlaps
        assert a["loss"] == pytest.approx(b["loss"], rel=1e-12)
        assert a["gsum"] == pytest.approx(b["gsum"], rel=1e-10) and a["gmax"] == pytest.approx(b["gmax"], rel=1e-10)


def test_a_failed_render_call_leaves_no_pointers_behind():
    """drt_outputs_clean registers raw pointers for the NEXT drt_render_forward; if that call fails (here: a null argument), or never comes
    (drt_outputs_cancel), the request must be gone: the buffers are released and the following call must not touch them."""
    from drt_amd import _lib, diffrender as Render, mesh_io, views
    from drt_amd.optix_mesh import _stream
    from conftest import data_path
    res = 64
    Render.intIOR = IOR
    Render.resx = Render.resy = res
    mesh = mesh_io.read_ply(data_path("hand_vh.ply"))
    c, ext = views.mesh_frame(mesh.vertices)
    cam = views.turntable_cameras(c, ext, 8, res, res)[2]
    o, d = views.generate_ray(res, res, cam[3], cam[2], device="cuda")
    n = o.shape[0]
    scene = Render.Scene(mesh, 0)
    ref = [t.clone() for t in scene.render_transparent(o, d)]
    h = scene.optix_mesh._h
    lib = _lib.lib()
    for how in ("failed call", "cancel"):
        junk = (torch.full((n, 3), 7.0, dtype=torch.float64, device="cuda"), torch.full((n, 3), 7.0, dtype=torch.float64, device="cuda"),
                torch.full((n, 3), 1, dtype=torch.uint8, device="cuda"))
        rows = torch.arange(n, dtype=torch.int32, device="cuda")
        n_rows = torch.tensor([n], dtype=torch.int64, device="cuda")
        _lib.check(lib.drt_outputs_clean(h, junk[0].data_ptr(), junk[1].data_ptr(), junk[2].data_ptr(), n, rows.data_ptr(), n_rows.data_ptr(), _stream()))
        if how == "failed call":
            rc = lib.drt_render_forward(h, None, o.data_ptr(), d.data_ptr(), n, IOR, 1.00029, junk[0].data_ptr(), junk[1].data_ptr(), junk[2].data_ptr(),
                                        rows.data_ptr(), rows.data_ptr(), None, None, res, res, 0, None, _stream())
            assert rc != 0
        else:
            _lib.check(lib.drt_outputs_cancel(h))
        got = scene.render_transparent(o, d)              # would run k_unwrite_rows over `junk` if the request were still registered
        torch.cuda.synchronize()
        assert all(float(t.min()) == (7.0 if k < 2 else 1) for k, t in enumerate(junk)), how      # untouched
        for a, b in zip(got, ref):
            assert torch.equal(a, b), how
