"""View-parallel step on CPU: two gloo processes shard the views, all-reduce the vertex
gradient once, apply limit_hook + SGD, and must land on bit-identical parameters that equal
the single-process result.  The ranks run drt_amd.optim.full_batch_step -- the SAME function bench.py and the
2-rank GPU test (tests/test_gpu_dist.py) call -- on a scene whose per-view loss is answered by the oracle (test
infrastructure: there is no GPU here); the single-process reference is written out by hand."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import IOR, ROOT, data_path
from drt_amd import dist as ddist, mesh_io, views

N_VIEWS, RES = 6, 24
RAY_W = 40 * 217.5 / RES / RES          # reference optim.py:127


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _view_grad(mesh, center, extent, k, V, n_views):
    from oracle import diffrender_oracle as orc
    R, K, Rinv, Kinv = views.turntable_cameras(center, extent, n_views, RES, RES)[k]
    o, d = views.generate_ray(RES, RES, Kinv, Rinv)
    rng = np.random.default_rng(100 + k)
    sp = torch.tensor(rng.standard_normal((RES * RES, 3)) * 40.0 + np.asarray(center))
    valid = torch.tensor(rng.random(RES * RES) > 0.1)
    Vr = V.clone().requires_grad_(True)
    oo, od, mk = orc.render_transparent(orc.Mesh(mesh.faces, Vr), o, d, IOR)
    loss = orc.ray_loss(oo, od, mk, sp, valid)
    g, = torch.autograd.grad(loss, Vr)
    return loss.detach(), g


class _OracleScene:
    """What drt_amd.optim.full_batch_step needs from a scene, answered by the CPU oracle (no GPU in this test):
    the code under test is the step itself -- sharding, the rank without views, ONE all-reduce, clamp after the sum, SGD."""

    def __init__(self, mesh):
        self.mesh_np = mesh
        self.vertices = torch.tensor(mesh.vertices, dtype=torch.float64)

    def update_verticex(self, vertices):
        self.vertices = vertices

    def ray_loss_fused(self, origin, ray_dir, target, valid):
        from oracle import diffrender_oracle as orc
        oo, od, mk = orc.render_transparent(orc.Mesh(self.mesh_np.faces, self.vertices), origin, ray_dir, IOR)
        return orc.ray_loss(oo, od, mk, target, valid)


def _views(mesh, center, extent, ids, n_views):
    out = []
    for k in ids:
        R, K, Rinv, Kinv = views.turntable_cameras(center, extent, n_views, RES, RES)[k]
        o, d = views.generate_ray(RES, RES, Kinv, Rinv)
        rng = np.random.default_rng(100 + k)
        sp = torch.tensor(rng.standard_normal((RES * RES, 3)) * 40.0 + np.asarray(center))
        out.append((sp, torch.tensor(rng.random(RES * RES) > 0.1), o, d))
    return out


def _reference_step(mesh, center, extent, n_views, V0, param, buf, ray_w, lr=0.1, mom=0.95):
    """The single-process full batch written out by hand (independent of drt_amd.optim): sum of the per-view
    gradients, limit_hook on the SUM (reference optim.py:155-162), SGD with nesterov momentum (optim.py:169)."""
    from drt_amd.optim import limit_hook
    g = torch.zeros_like(V0)
    loss = torch.zeros((), dtype=torch.float64)
    for k in range(n_views):
        l, gk = _view_grad(mesh, center, extent, k, V0 + param, n_views)
        g += ray_w * gk
        loss += l
    g = limit_hook(g)
    buf = g.clone() if buf is None else mom * buf + g
    return param - lr * (g + mom * buf), buf, loss


def _worker(rank, world, port, out_dir, n_views):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    r, lr_, w = ddist.init(backend="gloo")
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    from drt_amd import optim as O
    mesh = mesh_io.read_ply(data_path("hand_vh.ply"))
    center, extent = views.mesh_frame(mesh.vertices)
    scene = _OracleScene(mesh)
    mine = ddist.shard_views(n_views, rank, world)
    local = _views(mesh, center, extent, mine, n_views)
    hp = dict(O.HyperParams)
    init_vertices, parameter, opt = O.setup_opt(scene, 0.1, hp, hook=False)
    hooked = O.setup_opt(scene, 0.1, hp)[1]
    with pytest.raises(RuntimeError, match="hook"):
        O.full_batch_step(scene, local, init_vertices, hooked, opt, RAY_W, fused=True)
    losses = []
    for _ in range(2):
        loss = O.full_batch_step(scene, local, init_vertices, parameter, opt, RAY_W, fused=True).detach().clone()
        ddist.allreduce_sum_(loss)
        losses.append(loss.item())
    ddist.barrier()
    assert ddist.allreduce_max_float(float(rank), "cpu") == world - 1
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), param=parameter.detach().numpy(), losses=np.array(losses), views=np.array(mine, dtype=np.int64))
    dist.destroy_process_group()


def test_shard_views_partitions_all_views():
    for world in (1, 2, 4, 8):
        got = sorted(sum((ddist.shard_views(72, r, world) for r in range(world)), []))
        assert got == list(range(72))
        assert {len(ddist.shard_views(72, r, world)) for r in range(world)} == {72 // world}
    assert ddist.shard_views(72, 3, 8) == [3, 11, 19, 27, 35, 43, 51, 59, 67]


@pytest.mark.parametrize("world,n_views", [(2, N_VIEWS), (3, 2)])     # (3, 2): the third rank owns no view
def test_multi_rank_step_equals_single_process(tmp_path, world, n_views):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), n_views), nprocs=world, join=True)
    rs = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    assert sorted(sum((list(r["views"]) for r in rs), [])) == list(range(n_views))
    # every rank holds the same bits after the step (no broadcast needed)
    for r in rs[1:]:
        assert np.array_equal(rs[0]["param"], r["param"]) and np.array_equal(rs[0]["losses"], r["losses"])
    # and they equal the single-process full batch up to summation order
    mesh = mesh_io.read_ply(data_path("hand_vh.ply"))
    center, extent = views.mesh_frame(mesh.vertices)
    V0 = torch.tensor(mesh.vertices, dtype=torch.float64)
    param, buf = torch.zeros_like(V0), None
    losses = []
    for _ in range(2):
        param, buf, loss = _reference_step(mesh, center, extent, n_views, V0, param, buf, RAY_W)
        losses.append(loss.item())
    np.testing.assert_allclose(rs[0]["param"], param.numpy(), rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(rs[0]["losses"], losses, rtol=1e-12)
    assert np.abs(param.numpy()).max() > 0
