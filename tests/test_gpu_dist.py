"""The multi-GPU step on the HIP path: two ranks (gloo, both on cuda:0 -- the GPU box has one device; on an 8-GPU
node the same code runs one rank per GPU over RCCL) run drt_amd.optim.full_batch_step -- the function bench.py times --
for three steps.  Parameters must be bit-identical across ranks and agree with the single-rank run to 1e-12 relative
(the only difference is the summation order of float64 atomics and of the all-reduce)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import IOR, data_path

pytestmark = pytest.mark.gpu

N_VIEWS, RES, STEPS = 6, 128, 3


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(rank, world, fused):
    """Three full-batch steps of this rank's share of the views; returns (parameter, per-step global losses)."""
    from drt_amd import diffrender as Render, dist as ddist, mesh_io, optim as O, views
    Render.intIOR = IOR
    Render.resx = Render.resy = RES
    mesh = mesh_io.read_ply(data_path("hand_vh.ply"))
    center, extent = views.mesh_frame(mesh.vertices)
    scene = Render.Scene(mesh, 0)
    cams = views.turntable_cameras(center, extent, N_VIEWS, RES, RES)
    local = []
    for k in ddist.shard_views(N_VIEWS, rank, world):
        o, d = views.generate_ray(RES, RES, cams[k][3], cams[k][2], device="cuda")
        rng = np.random.default_rng(100 + k)
        sp = torch.tensor(rng.standard_normal((RES * RES, 3)) * 40.0 + np.asarray(center) + np.array([0.0, 0.0, 150.0]), device="cuda")
        local.append((sp, torch.tensor(rng.random(RES * RES) > 0.1, device="cuda"), o, d))
    init_vertices, parameter, opt = O.setup_opt(scene, 0.1, O.HyperParams, hook=False, fused=fused)     # (the fused terms with the one-kernel optimiser)
    ray_w = 40 * 217.5 / RES / RES
    losses = []
    for _ in range(STEPS):
        loss = O.full_batch_step(scene, local, init_vertices, parameter, opt, ray_w, fused=fused).detach().clone()
        ddist.allreduce_sum_(loss)
        losses.append(loss.item())
    torch.cuda.synchronize()
    return parameter.detach().cpu().numpy(), np.array(losses)


def _worker(rank, world, port, out_dir, fused):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world),
                      DRT_DIST_BACKEND="gloo")
    from drt_amd import dist as ddist
    torch.cuda.set_device(0)
    r, _, w = ddist.init(backend="gloo")
    assert (r, w) == (rank, world)
    param, losses = _run(rank, world, fused)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), param=param, losses=losses)
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("fused", [False, True])
def test_two_ranks_on_the_hip_path_equal_one_rank(tmp_path, fused):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), fused), nprocs=world, join=True)
    r0, r1 = (np.load(tmp_path / f"rank{r}.npz") for r in range(world))
    assert np.array_equal(r0["param"], r1["param"]) and np.array_equal(r0["losses"], r1["losses"])   # bit-identical ranks
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    param, losses = _run(0, 1, fused)
    assert np.abs(param).max() > 1e-3 and np.isfinite(param).all()
    scale = np.abs(param).max()
    assert np.abs(r0["param"] - param).max() <= 1e-12 * scale, np.abs(r0["param"] - param).max() / scale
    np.testing.assert_allclose(r0["losses"], losses, rtol=1e-12)


def _graph_run(fused, graph, steps=4, recycle_min=None):
    """`steps` full-batch steps on one rank, eagerly or as replays of ONE captured hipGraph (bench.py --graph 1)."""
    from drt_amd import diffrender as Render, mesh_io, optim as O, views
    if recycle_min is not None:
        Render.RECYCLE_MIN_RAYS = recycle_min
    Render.intIOR = IOR
    Render.resx = Render.resy = RES
    mesh = mesh_io.read_ply(data_path("hand_vh.ply"))
    center, extent = views.mesh_frame(mesh.vertices)
    scene = Render.Scene(mesh, 0)
    cams = views.turntable_cameras(center, extent, N_VIEWS, RES, RES)
    parts = []
    for k in range(3):
        o, d = views.generate_ray(RES, RES, cams[k][3], cams[k][2], device="cuda")
        rng = np.random.default_rng(100 + k)
        sp = torch.tensor(rng.standard_normal((RES * RES, 3)) * 40.0 + np.asarray(center) + np.array([0.0, 0.0, 150.0]), device="cuda")
        parts.append((sp, torch.tensor(rng.random(RES * RES) > 0.1, device="cuda"), o, d))
    local = [tuple(torch.cat([p[j] for p in parts]).contiguous() for j in range(4))]       # one call over three images, like bench.py
    init_vertices, parameter, opt = O.setup_opt(scene, 0.1, O.HyperParams, hook=False, fused=True)
    ray_w = 40 * 217.5 / RES / RES
    step = lambda: O.full_batch_step(scene, local, init_vertices, parameter, opt, ray_w, fused=fused)
    warm = 3                                             # establishes the grid verdict, reads it back, sizes every workspace
    if graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warm):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        for _ in range(steps):           # (capturing records the step, it does not run it)
            g.replay()
    else:
        for _ in range(warm + steps):
            step()
    torch.cuda.synchronize()
    return parameter.detach().cpu().numpy()


@pytest.mark.parametrize("fused", [False, True])
def test_a_whole_step_captured_as_a_hip_graph_equals_eager_steps(fused):
    """The step has no host-side data dependence, so it can be captured once and replayed (internal streams, asynchronous build,
    late output fills and all): the parameters after warm-up + 4 steps equal the eager run's up to the order of the atomics."""
    eager, replayed = _graph_run(fused, False), _graph_run(fused, True)
    scale = np.abs(eager).max()
    assert scale > 1e-3 and np.isfinite(replayed).all()
    assert np.abs(eager - replayed).max() <= 1e-11 * scale, np.abs(eager - replayed).max() / scale


def test_a_captured_step_recycles_its_outputs_replay_after_replay():
    """With recycling on for this size the capture takes a pooled output set for good (diffrender `graph_set`): the captured call zeroes the
    rows of the set's row list and writes its own list into the same buffers, so every replay undoes what its predecessor set -- no dense
    fill inside the graph.  Same parameters as eager steps with fresh outputs, over 6 replays with moving vertices."""
    from drt_amd import diffrender as Render
    old = (Render.RECYCLE_MIN_RAYS, Render.RECYCLE_OUTPUTS)
    try:
        Render.RECYCLE_OUTPUTS = False
        eager = _graph_run(False, False, steps=6)
        Render.RECYCLE_OUTPUTS = old[1]
        if not Render.RECYCLE_OUTPUTS:
            pytest.skip("this torch has no storage use count: no recycling")
        Render.cache_report(reset=True)
        replayed = _graph_run(False, True, steps=6, recycle_min=0)
        rep = Render.cache_report()
    finally:
        Render.RECYCLE_MIN_RAYS, Render.RECYCLE_OUTPUTS = old
    assert rep.get("recycle_graph_set", 0) == 1 and rep.get("recycle_off_capture", 0) == 0, rep      # the capture DID take a pooled set
    scale = np.abs(eager).max()
    assert scale > 1e-3 and np.isfinite(replayed).all()
    assert np.abs(eager - replayed).max() <= 1e-11 * scale, np.abs(eager - replayed).max() / scale


def _nccl_worker(rank, world, port, out_dir, fused, graph):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      DRT_DIST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if world == 1:
        os.environ["DRT_DIST_FORCE"] = "1"          # create the process group and issue the collectives with ONE rank
    import importlib
    from drt_amd import dist as ddist
    importlib.reload(ddist)                         # (DRT_DIST_FORCE is read at import)
    import drt_amd.optim as O
    O.ddist = ddist
    torch.cuda.set_device(rank)
    r, _, w = ddist.init(backend="nccl")
    assert (r, w) == (rank, world)
    # SURVEY section 8e rests on the all-reduce delivering the SAME bits to every rank: checked on a vector whose partial sums round
    probe = torch.tensor(np.random.default_rng(7 + rank).standard_normal(75378) * 10.0 ** np.random.default_rng(3).integers(-8, 8, 75378), device="cuda")
    ddist.allreduce_sum_(probe)
    from drt_amd import diffrender as Render, mesh_io, views
    Render.intIOR = IOR
    Render.resx = Render.resy = RES
    mesh = mesh_io.read_ply(data_path("hand_vh.ply"))
    center, extent = views.mesh_frame(mesh.vertices)
    scene = Render.Scene(mesh, rank)
    cams = views.turntable_cameras(center, extent, N_VIEWS, RES, RES)
    parts = []
    for k in (range(3) if world == 1 else ddist.shard_views(N_VIEWS, rank, world)):       # (one rank: the three views of _graph_run)
        o, d = views.generate_ray(RES, RES, cams[k][3], cams[k][2], device="cuda")
        rng = np.random.default_rng(100 + k)
        sp = torch.tensor(rng.standard_normal((RES * RES, 3)) * 40.0 + np.asarray(center) + np.array([0.0, 0.0, 150.0]), device="cuda")
        parts.append((sp, torch.tensor(rng.random(RES * RES) > 0.1, device="cuda"), o, d))
    local = [tuple(torch.cat([p[j] for p in parts]).contiguous() for j in range(4))]
    init_vertices, parameter, opt = O.setup_opt(scene, 0.1, O.HyperParams, hook=False, fused=True)
    ray_w = 40 * 217.5 / RES / RES
    step = lambda: O.full_batch_step(scene, local, init_vertices, parameter, opt, ray_w, fused=fused)
    if graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        for _ in range(STEPS):
            g.replay()
    else:
        for _ in range(3 + STEPS):
            step()
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"nccl{rank}.npz"), param=parameter.detach().cpu().numpy(), probe=probe.cpu().numpy())
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("graph", [False, True], ids=["eager", "whole-step-graph"])
def test_rccl_step_with_one_rank_per_gpu(tmp_path, graph):
    """The step over RCCL (backend "nccl"), eagerly and as a captured hipGraph with the all-reduce INSIDE the graph -- what bench.py runs
    for N > 1.  With two or more GPUs: two ranks, one per GPU, bit-identical parameters and all-reduce results on both (the
    replicas-stay-identical-without-a-broadcast assumption of SURVEY section 8e), equal to the one-rank run up to summation order.  On a
    one-GPU box: ONE rank with the process group forced on (DRT_DIST_FORCE), so that communicator creation, the collective's launch
    and its capture have run at least once before the first multi-GPU driver run."""
    world = 2 if torch.cuda.device_count() >= 2 else 1
    mp.spawn(_nccl_worker, args=(world, _free_port(), str(tmp_path), False, graph), nprocs=world, join=True)
    r = [np.load(tmp_path / f"nccl{k}.npz") for k in range(world)]
    assert np.isfinite(r[0]["param"]).all() and np.abs(r[0]["param"]).max() > 1e-3
    for other in r[1:]:
        assert np.array_equal(r[0]["param"], other["param"]) and np.array_equal(r[0]["probe"], other["probe"])
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "DRT_DIST_FORCE", "DRT_DIST_BACKEND"):
        os.environ.pop(k, None)
    ref = _graph_run(False, False, steps=STEPS) if world == 1 else None
    if ref is not None:      # one rank: the same three views in one call as _graph_run -> same numbers up to the order of the atomics
        scale = np.abs(ref).max()
        assert np.abs(r[0]["param"] - ref).max() <= 1e-11 * scale


def test_a_replayed_update_rebuilds_the_scene_box_from_the_vertices_of_the_replay():
    """A captured `update_verticex` must gather the scene box from the vertices present at REPLAY time: the reset of the box
    accumulators is enqueued work (a kernel of the build puts them back), not host state.  The mesh shrinks between the replays; the box
    and the leaf padding of every replay equal those of an eager update with the same vertices (a box that can only grow -- what a
    host-side double buffer gave -- fails at the second replay)."""
    from drt_amd import diffrender as Render, mesh_io
    mesh = mesh_io.read_ply(data_path("hand_vh.ply"))
    scene, ref = Render.Scene(mesh, 0), Render.Scene(mesh, 0)
    v0 = torch.tensor(mesh.vertices, dtype=torch.float64, device="cuda")
    c = v0.mean(0)
    static = v0.clone()
    rays = torch.cat([c.float() + torch.tensor([0.0, 0.0, 400.0], device="cuda"), torch.tensor([0.0, 0.0, -1.0], device="cuda")]).repeat(64, 1).contiguous()
    step = lambda: (scene.update_verticex(static), scene.optix_mesh.intersect_any(rays))[1]     # (a consumer of the tree joins the build stream)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        hit = step()
    pads = []
    for scale in (1.0, 3.0, 0.5, 0.1, 0.7):              # grows once, then shrinks well below the first box
        static.copy_(c + (v0 - c) * scale)
        g.replay()
        torch.cuda.synchronize()
        got = scene.optix_mesh.build_params()
        ref.update_verticex(static.clone())
        want = ref.optix_mesh.build_params()
        assert got == want, (scale, got, want)
        assert scene.optix_mesh.check()[0] == 0
        pads.append(got[6])
    assert pads[3] < 0.2 * pads[0] < pads[1]


def test_first_overflow_area_use_under_capture_is_refused_with_a_message():
    """The overflow area of the one-thread-per-item queries is allocated by their first call; under stream capture that call is refused
    with a message that says what to do, and works after one eager call."""
    from drt_amd import diffrender as Render, mesh_io
    mesh = mesh_io.read_ply(data_path("hand_vh.ply"))
    scene = Render.Scene(mesh, 0)
    pts = torch.tensor(mesh.vertices[:64], dtype=torch.float64, device="cuda") + 1.0
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with pytest.raises(RuntimeError, match="eagerly before capturing"):
        with torch.cuda.graph(g):
            scene.optix_mesh.closest_point(pts)
    torch.cuda.synchronize()
    eager = scene.optix_mesh.closest_point(pts)[0].clone()
    g2 = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        scene.optix_mesh.closest_point(pts)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    with torch.cuda.graph(g2):
        out = scene.optix_mesh.closest_point(pts)[0]
    g2.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
