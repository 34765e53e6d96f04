"""DRT_DETERMINISTIC (include/drt_hip.h drt_deterministic, drt_amd/det.py, csrc/drt_fixed.h): the vertex gradient the reference clamps
after summing it (optim.py:155-171) is accumulated as 128-bit fixed-point integers, so that it no longer depends on the order in which the
hardware served the atomics ("race detection / sanitizers").  Asserted here: bit equality run to run, eager vs graph
replay, across every loss term of an iteration; a sum split over two "ranks" within one rounding of its parts; and agreement with the
float64-atomic mode to 1e-12."""
import numpy as np
import pytest
import torch

from conftest import IOR, data_path, golden

pytestmark = pytest.mark.gpu

RES, N_VIEWS = 128, 6


@pytest.fixture()
def det_on():
    from drt_amd import det
    was = det.enable(True)
    yield det
    det.enable(was)


def _setup():
    from drt_amd import diffrender as Render, mesh_io, views
    Render.intIOR = IOR
    Render.resx = Render.resy = RES
    hand = mesh_io.read_ply(data_path("hand_vh.ply"))
    mesh = mesh_io.TriMesh(golden("hand_smooth_sm")["vertices"].astype(np.float64), hand.faces)      # (smoothed: every dihedral term is finite)
    center, extent = views.mesh_frame(mesh.vertices)
    scene = Render.Scene(mesh, 0)
    cams = views.turntable_cameras(center, extent, N_VIEWS, RES, RES)
    parts = []
    for k in range(N_VIEWS):
        o, d = views.generate_ray(RES, RES, cams[k][3], cams[k][2], device="cuda")
        rng = np.random.default_rng(100 + k)
        sp = torch.tensor(rng.standard_normal((RES * RES, 3)) * 40.0 + np.asarray(center) + np.array([0.0, 0.0, 150.0]), device="cuda")
        parts.append((sp, torch.tensor(rng.random(RES * RES) > 0.1, device="cuda"), o, d))
    return Render, scene, cams, parts


def _cat(parts, ids):
    return [tuple(torch.cat([parts[k][j] for k in ids]).contiguous() for j in range(4))]


def _ray_grad(scene, local, fused):
    """(loss, d loss / d vertices) of ONE call over the views in `local`, through autograd."""
    from drt_amd import optim as O
    init = scene.vertices.detach().clone()
    parameter = torch.zeros_like(init, requires_grad=True)
    loss = O.local_loss_backward(scene, local, init, parameter, 1.0, fused=fused)
    torch.cuda.synchronize()
    return loss.detach().clone(), parameter.grad.detach().clone()


@pytest.mark.parametrize("fused", [False, True])
def test_ray_term_is_bit_identical_run_to_run_and_agrees_with_the_atomic_mode(det_on, fused):
    Render, scene, cams, parts = _setup()
    local = _cat(parts, range(N_VIEWS))
    runs = [_ray_grad(scene, local, fused) for _ in range(4)]
    for l, g in runs[1:]:
        assert torch.equal(l, runs[0][0]) and torch.equal(g, runs[0][1])
    # a second scene object (other streams, other workspaces), recycled vs fresh outputs: still the same bits
    scene2 = Render.Scene(scene.mesh, 0)
    l2, g2 = _ray_grad(scene2, local, fused)
    assert torch.equal(l2, runs[0][0]) and torch.equal(g2, runs[0][1])
    det_on.enable(False)
    la, ga = _ray_grad(scene, local, fused)
    det_on.enable(True)
    scale = float(ga.abs().max())
    assert scale > 0 and float((ga - runs[0][1]).abs().max()) <= 1e-12 * scale
    assert abs(float(la) - float(runs[0][0])) <= 1e-12 * abs(float(la))


def test_a_sum_split_over_two_ranks_is_within_one_rounding_of_its_parts(det_on):
    """View k -> rank k mod 2 (drt_amd/dist.py): each rank's gradient is the correctly rounded exact sum of ITS contributions, the
    all-reduce adds the two float64 values.  Against the one-rank result (the correctly rounded sum of everything) that leaves half an ulp
    of each part and of the total -- no more, whatever the order of anything."""
    Render, scene, cams, parts = _setup()
    _, g_all = _ray_grad(scene, _cat(parts, range(N_VIEWS)), False)
    _, g_a = _ray_grad(scene, _cat(parts, range(0, N_VIEWS, 2)), False)
    _, g_b = _ray_grad(scene, _cat(parts, range(1, N_VIEWS, 2)), False)
    two = g_a + g_b
    # truncation of sub-unit contributions (< 2^-80 each, a few thousand per vertex at most) is far below these bounds
    bound = (g_a.abs() + g_b.abs() + two.abs()) * 2.0 ** -53 + 1e-20
    assert bool(((two - g_all).abs() <= bound).all()), float(((two - g_all).abs() / bound).max())


@pytest.mark.parametrize("fused", [False, True])
def test_eager_steps_and_graph_replays_give_the_same_parameters(det_on, fused):
    from test_gpu_dist import _graph_run
    eager, replayed = _graph_run(fused, False), _graph_run(fused, True)
    assert np.abs(eager).max() > 1e-3
    assert np.array_equal(eager, replayed)
    assert np.array_equal(eager, _graph_run(fused, False))


def _iteration_grads(Render, scene, cams, soft, lazy):
    """The silhouette and smoothness terms of one iteration (reference optim.py:67-89), drop-in expressions, through autograd."""
    Render.LAZY_VISIBILITY = lazy
    V = scene.vertices.detach().clone().requires_grad_(True)
    scene.update_verticex(V)
    total = torch.zeros((), dtype=torch.float64, device="cuda")
    for k in range(0, N_VIEWS, 2):
        camera_M = tuple(torch.tensor(np.asarray(a), dtype=torch.float64, device="cuda") for a in cams[k])
        eye = torch.tensor(np.asarray(cams[k][2])[:3, 3].copy(), dtype=torch.float64, device="cuda")
        index, output = scene.primary_visibility(scene.silhouette_edge(eye), camera_M, eye, detach_depth=True)
        total = total + 0.3 * (soft.view((RES, RES))[index[:, 1], index[:, 0]] - output).abs().sum()
    total = total + 0.01 * (-torch.log(1 + scene.dihedral_angle())).sum()
    total.backward()
    torch.cuda.synchronize()
    return total.detach().clone(), V.grad.clone()


def test_silhouette_and_smoothness_terms_are_bit_identical_run_to_run(det_on):
    Render, scene, cams, parts = _setup()
    old = Render.LAZY_VISIBILITY
    rng = np.random.default_rng(2)
    soft = torch.tensor(np.round(rng.random(RES * RES) * 4) / 4, device="cuda")
    try:
        for lazy in (True, False):
            a = _iteration_grads(Render, scene, cams, soft, lazy)
            b = _iteration_grads(Render, scene, cams, soft, lazy)
            assert float(a[1].abs().max()) > 0 and bool(torch.isfinite(a[1]).all())
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        # the one-pass forms of the same terms
        views_ = []
        for k in range(0, N_VIEWS, 2):
            camera_M = tuple(torch.tensor(np.asarray(a), dtype=torch.float64, device="cuda") for a in cams[k])
            views_.append((camera_M, torch.tensor(np.asarray(cams[k][2])[:3, 3].copy(), dtype=torch.float64, device="cuda"), soft))
        outs = []
        for _ in range(2):
            V = scene.vertices.detach().clone().requires_grad_(True)
            scene.update_verticex(V)
            t = scene.vh_loss_fused_views(views_) + 0.01 * scene.sm_loss_fused()
            t.backward()
            outs.append((t.detach().clone(), V.grad.clone()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    finally:
        Render.LAZY_VISIBILITY = old


def test_whole_iterations_of_the_one_pass_loop_repeat_bit_for_bit(det_on):
    """drt_amd.optim.FusedIteration (refraction + silhouette + smoothness terms, limit_hook, SGD) for five iterations, twice from scratch."""
    from drt_amd import captured_data, diffrender as Render, mesh_io, optim as O, views
    Render.intIOR = IOR
    Render.resx = Render.resy = RES
    mesh = mesh_io.read_ply(data_path("hand_vh.ply"))
    center, extent = views.mesh_frame(mesh.vertices)
    gt = Render.Scene(views.displaced_ground_truth(mesh, sigma=0.3, seed=0), 0)
    data = captured_data.SyntheticData(gt, center, extent, RES, RES, num_view=8, n_total=8)
    results = []
    for _ in range(2):
        data.rng = np.random.RandomState(0)
        np.random.seed(0)
        scene = Render.Scene(mesh, 0)
        it = O.FusedIteration(scene, data, dict(O.HyperParams, num_view=8), 0.1)
        for _ in range(5):
            total, parts = it.step()
        torch.cuda.synchronize()
        results.append((it.parameter.clone(), parts.clone()))
    assert float(results[0][0].abs().max()) > 0
    assert torch.equal(results[0][0], results[1][0]) and torch.equal(results[0][1], results[1][1])


def test_cells_survive_the_exchange_format_and_sum_exactly(det_on):
    """drt_fx_to_limbs / drt_fx_from_limbs (the words an int64 all-reduce sums) and drt_fx_add: a round trip is the identity, and the sum of
    two ranks' words is the cell-wise 128-bit sum -- for sums of either sign and for the sticky flags."""
    from drt_amd import _lib
    lib = _lib.lib()
    n = 4096
    rng = np.random.default_rng(11)
    def cells_of(x):
        # accumulate each value alone into its cell through a kernel that takes a gradient target: k_dihedral... simpler: finalize's inverse
        # does not exist, so build cells from Python integers (value * 2^80, two's complement) on the host
        out = np.zeros((len(x), 3), dtype=np.int64)
        for i, v in enumerate(x):
            if not np.isfinite(v):
                out[i, 2] = 1 if np.isnan(v) else (2 if v > 0 else 4)
                continue
            q = int(v * 2.0 ** 40) << 40                      # (an integer multiple of 2^-40: exact in float64 and in the cells)
            q %= 1 << 128
            lo, hi = q & ((1 << 64) - 1), q >> 64
            out[i, 0] = hi - (1 << 64) if hi >= 1 << 63 else hi
            out[i, 1] = lo - (1 << 64) if lo >= 1 << 63 else lo
        return torch.tensor(out.reshape(-1), device="cuda")
    a = np.round(rng.standard_normal(n) * 10.0 ** rng.uniform(-3, 6, n) * 2.0 ** 40) / 2.0 ** 40
    b = np.round(rng.standard_normal(n) * 10.0 ** rng.uniform(-3, 6, n) * 2.0 ** 40) / 2.0 ** 40
    b[:8] = -a[:8]                                            # exact cancellation
    a[8], b[9], a[10], b[10] = np.inf, np.nan, np.inf, -np.inf
    ca, cb = cells_of(a), cells_of(b)
    la, lb = (torch.empty(4 * n, dtype=torch.int64, device="cuda") for _ in range(2))
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.drt_fx_to_limbs(ca.data_ptr(), n, la.data_ptr(), s))
    _lib.check(lib.drt_fx_to_limbs(cb.data_ptr(), n, lb.data_ptr(), s))
    back = torch.empty_like(ca)
    _lib.check(lib.drt_fx_from_limbs(la.data_ptr(), n, back.data_ptr(), s))
    assert torch.equal(back, ca)
    summed = torch.empty_like(ca)
    _lib.check(lib.drt_fx_from_limbs((la + lb).data_ptr(), n, summed.data_ptr(), s))          # what an all-reduce(SUM) over two ranks delivers
    direct = ca.clone()
    _lib.check(lib.drt_fx_add(direct.data_ptr(), cb.data_ptr(), n, s))
    assert torch.equal(summed, direct)
    got = det_on.value(summed, torch.empty(n, dtype=torch.float64, device="cuda")).cpu().numpy()
    with np.errstate(invalid="ignore"):
        want = a + b                                          # (multiples of 2^-40 below 2^63: the float64 sum is exact too)
    assert np.array_equal(got[:8], np.zeros(8)) and got[8] == np.inf and np.isnan(got[9]) and np.isnan(got[10])
    assert np.array_equal(got[11:], want[11:])


def _worker_det(rank, world, port, out_dir):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world),
                      DRT_DIST_BACKEND="gloo", DRT_DETERMINISTIC="1")
    from drt_amd import dist as ddist
    from test_gpu_dist import _run
    torch.cuda.set_device(0)
    ddist.init(backend="gloo")
    param, losses = _run(rank, world, False)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), param=param, losses=losses)
    torch.distributed.destroy_process_group()


def test_two_ranks_give_the_bits_one_rank_gives(det_on, tmp_path):
    """full_batch_step in deterministic mode exchanges the ranks' accumulators as integers and converts once: the parameters after three
    steps on two ranks (views k -> rank k mod 2, gloo on this box's one GPU) are bit-identical to one rank's -- not merely within an ulp."""
    import os
    import torch.multiprocessing as mp
    from test_gpu_dist import _free_port, _run
    mp.spawn(_worker_det, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (np.load(tmp_path / f"rank{r}.npz") for r in range(2))
    assert np.array_equal(r0["param"], r1["param"])
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    param, losses = _run(0, 1, False)
    assert np.abs(param).max() > 1e-3
    assert np.array_equal(r0["param"], param)

