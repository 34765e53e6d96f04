"""The optimisation loop the refraction path plugs into -- host-side mirror of the
reference's optim.py, on the HIP-backed ``drt_amd.diffrender``.

    Loss_calculator.ray_loss / vh_loss / sm_loss / all_loss   reference optim.py:59-130
    limit_hook, setup_opt, interp_L / interp_R, optimize       reference optim.py:145-219

Differences from the reference, all outside the per-view math:
  * the capture comes from any object with the reference's ``Data`` interface (``get_view``,
    ``ray_view_generator``, ``silh_view_generator``, ``resx``, ``resy``): drt_amd.captured_data has the
    reference's capture classes and ``SyntheticData``, which stands in for the HDF5 captures (not distributed);
  * the MeshLab remesh between passes (optim.py:12-52, an external program) is done in-process by
    drt_amd.remesh (same algorithm and parameters); ``remesh=`` takes any other callable, or None;
  * multi-GPU: ``full_batch_step`` shards views over ranks and all-reduces the vertex gradient
    once per step (drt_amd.dist); the reference is single-GPU and one view per step.
"""
from __future__ import annotations

import time

import numpy as np
import torch

from . import diffrender as Render
from . import dist as ddist
from . import mesh_io, views
from .captured_data import Data, Data_Pointgray, Data_Redmi, SyntheticData, get_data  # noqa: F401  (reference optim.py:7, 132-143)

Float = torch.float64

HyperParams = {          # reference config.py:18-39
    "name": "hand", "IOR": 1.4723, "Pass": 20, "Iters": 200,
    "ray_w": 40, "sm_w": 0.08, "vh_w": 2e-3,
    "momentum": 0.95, "start_lr": 0.1, "lr_decay": 0.5, "start_len": 10, "end_len": 1, "num_view": 72,
}


def loss_weights(hp, resy, mean_len):
    """(w_ray, w_vh, w_sm): the fixed scalings of reference optim.py:127-129."""
    return hp["ray_w"] * 217.5 / resy / resy, hp["vh_w"] * 217.5 / resy, hp["sm_w"] * mean_len / 10


class Loss_calculator:
    """Same role and method names as the reference class (optim.py:59-130); every term is evaluated by
    the HIP kernels behind ``drt_amd.diffrender``.  ``fused=True`` uses the one-pass kernels
    (``Scene.ray_loss_fused`` / ``Scene.vh_loss_fused`` / ``Scene.sm_loss_fused``) -- same values, no dense
    intermediates and no host synchronisation."""

    N_SILHOUETTE_VIEWS = 8          # the reference loops over np.arange(0, 72, 9)

    def __init__(self, scene, data, HyperParams, fused=False):
        self.scene, self.data, self.HyperParams, self.fused = scene, data, HyperParams, fused
        self.ray_view = data.ray_view_generator()
        self.silh_view = data.silh_view_generator()

    def _silhouette_term(self, view_id):
        """sum |soft_mask[y, x] - 0.5| over the visible silhouette samples of one view (optim.py:74-78)."""
        _, _, soft_mask, origin, _, camera_M = self.data.get_view(view_id)
        eye = origin[0]
        edges = self.scene.silhouette_edge(eye)
        pix, out = self.scene.primary_visibility(edges, camera_M, eye, detach_depth=True)
        image = soft_mask.view((self.data.resy, self.data.resx))
        return (image[pix[:, 1], pix[:, 0]] - out).abs().sum()

    def vh_loss(self):
        if self.fused:
            views = []
            for _ in range(self.N_SILHOUETTE_VIEWS):
                _, _, soft_mask, origin, _, camera_M = self.data.get_view(next(self.silh_view))
                views.append((camera_M, origin[0], soft_mask))
            return self.scene.vh_loss_fused_views(views)
        # accumulated like the reference does (optim.py:71-78: `vh_loss = 0; vh_loss += term`): with the lazy pair of primary_visibility the
        # terms are not evaluated one by one -- the sum is ONE fused launch over the eight views, enqueued here (Render.force), i.e. on the
        # stream this method is called under
        total = 0
        for _ in range(self.N_SILHOUETTE_VIEWS):
            total += self._silhouette_term(next(self.silh_view))
        return Render.force(total)

    def sm_loss(self):
        if self.fused:
            return self.scene.sm_loss_fused()
        cos_dihedral = self.scene.dihedral_angle()
        return (-torch.log(1 + cos_dihedral)).sum()

    def ray_loss(self):
        view_id = next(self.ray_view)
        target, valid, _, origin, ray_dir, _ = self.data.get_view(view_id)
        bind = getattr(self.data, "get_binding", None)       # a capture whose views are resident constants offers a handle per view (RayBinding)
        if bind is not None:
            origin, ray_dir = bind(self.scene, view_id), None
        if self.fused:
            return self.scene.ray_loss_fused(origin, ray_dir, target, valid)
        exit_o, exit_d, exit_mask = self.scene.render_transparent(origin, ray_dir)
        return Render.ray_loss(exit_o, exit_d, exit_mask, target, valid)

    # The three terms of an iteration are independent given the mesh.  The silhouette and smoothness terms are enqueued on a SIDE stream and
    # run beside the refraction term -- at the reference's iteration size (one 960x1280 view, ~50 k primary hits) each term is a chain of
    # small launches that leaves most of the chip idle, so the iteration costs the longest chain instead of their sum.  With the drop-in
    # methods (fused=False) the silhouette term synchronises with the host once per view (its outputs are dynamically sized): on its own
    # stream that wait covers its own launches only, not the refraction term enqueued before it.  Values are unchanged (same kernels, same
    # inputs).
    CONCURRENT_TERMS = True

    def all_loss(self):
        hp = self.HyperParams
        none = torch.zeros((), dtype=Float, device=self.scene.vertices.device)
        if self.CONCURRENT_TERMS and self.scene.vertices.is_cuda and not torch.cuda.is_current_stream_capturing():
            main = torch.cuda.current_stream()
            side = getattr(self, "_side_stream", None)
            if side is None:
                side = self._side_stream = torch.cuda.Stream(device=self.scene.vertices.device)
            side.wait_stream(main)                      # the rebuild (update_verticex) and the vertices are enqueued on `main`
            # (the refraction term first: it is the longest chain, and at this size the HOST's enqueue order is the GPU's start order)
            ray = self.ray_loss() if hp["ray_w"] != 0 else none
            with torch.cuda.stream(side):
                vh = self.vh_loss() if hp["vh_w"] != 0 else none
                sm = self.sm_loss() if hp["sm_w"] != 0 else none
            main.wait_stream(side)
            for t in (vh, sm):
                t.record_stream(main)                   # allocated on the side stream, consumed on `main` from here on
            parts = (ray, vh, sm)
        else:
            parts = (self.ray_loss() if hp["ray_w"] != 0 else none,
                     self.vh_loss() if hp["vh_w"] != 0 else none,
                     self.sm_loss() if hp["sm_w"] != 0 else none)
        w = loss_weights(hp, self.data.resy, self.scene.mean_len)
        if self.fused:
            # one stack + one dot instead of five scalar kernels (and five more in the backward pass): the iteration is a chain of small launches
            key = (w, parts[0].device)
            if getattr(self, "_w_key", None) != key:
                self._w_key, self._w_vec = key, torch.tensor(w, dtype=Float, device=parts[0].device)
            total = torch.dot(torch.stack(parts), self._w_vec)
        else:
            total = w[0] * parts[0] + w[1] * parts[1] + w[2] * parts[2]
        return total, parts


def loss_string(parts):
    ray_loss, vh_loss, sm_loss = parts
    return f"ray={float(ray_loss.detach()):g} vh={float(vh_loss.detach()):g} sm={float(sm_loss.detach()):g}"


def interp_L(start, end, it, Pass):
    assert it <= Pass - 1
    return it * ((end - start) / (Pass - 1)) + start


def interp_R(start, end, it, Pass):
    return 1 / interp_L(1 / start, 1 / end, it, Pass)


def limit_hook(grad, max_abs=1.0):
    """NaN -> 0, clamp to +-1 (reference optim.py:155-162); +-inf clamps like any large value."""
    g = torch.nan_to_num(grad, nan=0.0, posinf=None, neginf=None)
    return g.clamp_(-max_abs, max_abs)


class FusedLimitSGD:
    """limit_hook + torch.optim.SGD(momentum, nesterov) for ONE float64 parameter in one kernel (drt_limit_sgd_step) instead
    of ~8 elementwise launches: ``step()`` sanitises ``parameter.grad`` in place (NaN -> 0, clamp +-max_abs: the
    reference's hook, optim.py:155-162) and applies the update.  Same numbers as the hook + torch.optim.SGD."""

    applies_limit = True

    def __init__(self, parameter, lr, momentum=0.0, nesterov=False, max_abs=1.0):
        self.parameter, self.lr, self.momentum, self.nesterov, self.max_abs = parameter, float(lr), float(momentum), bool(nesterov), float(max_abs)
        self.buf = None
        self.param_groups = [{"params": [parameter], "lr": self.lr, "momentum": self.momentum, "nesterov": self.nesterov}]

    def zero_grad(self, set_to_none=True):
        if set_to_none:
            self.parameter.grad = None
        elif self.parameter.grad is not None:
            self.parameter.grad.zero_()

    def step(self):
        from . import _lib
        from .optix_mesh import _stream, _on
        p, g = self.parameter, self.parameter.grad
        if g is None:
            return
        assert p.is_cuda and p.dtype == torch.float64 and g.dtype == torch.float64 and p.is_contiguous() and g.is_contiguous()
        first = self.buf is None
        if first and self.momentum != 0.0:
            self.buf = torch.empty_like(p)
        with _on(p.device):
            _lib.check(_lib.lib().drt_limit_sgd_step(p.data_ptr(), g.data_ptr(), _lib.ptr(self.buf), p.numel(), self.param_groups[0]["lr"], self.momentum,
                                                      int(self.nesterov), int(first), self.max_abs, _stream()))


class FusedIteration:
    """One iteration of the reference's loop (optim.py:198-215: vertices = init + parameter, update_verticex, all_loss, backward,
    limit_hook, SGD) on the one-pass kernels WITHOUT the autograd graph around them.

    With ``Loss_calculator(fused=True)`` every term already returns its loss together with d term / d vertices; autograd only
    multiplies those by the weights and adds them up -- ~15 tiny torch ops, three Function nodes and a backward pass whose HOST cost
    (0.7 ms) exceeds the GPU time of the whole iteration at the reference's size (one 960x1280 refraction view, 8 silhouette views).
    Here the three entry points write into one [3, V, 3] buffer, the weighted sum is one matrix product and the update is the
    one-kernel limit_hook + SGD(nesterov): the same kernels, inputs and arithmetic, ~10 host calls.  The silhouette and smoothness
    terms run on a side stream beside the refraction term.  Same view schedule generators as Loss_calculator."""

    N_SILHOUETTE_VIEWS = 8

    def __init__(self, scene, data, HyperParams, lr, concurrent=True):
        from . import _lib
        self._lib = _lib
        self.scene, self.data, self.hp = scene, data, HyperParams
        self.ray_view = data.ray_view_generator()
        self.silh_view = data.silh_view_generator()
        dev = scene.vertices.device
        self.init_vertices = scene.vertices.detach().clone()
        self.parameter = torch.zeros_like(self.init_vertices)
        self.grads = torch.zeros((3,) + tuple(self.init_vertices.shape), dtype=Float, device=dev)
        self.losses = torch.zeros(3, dtype=Float, device=dev)
        self.total = torch.zeros((), dtype=Float, device=dev)
        self.total_grad = torch.empty_like(self.init_vertices)
        self.buf = torch.empty_like(self.init_vertices) if HyperParams["momentum"] != 0 else None
        self.first = True
        self.lr, self.momentum = float(lr), float(HyperParams["momentum"])
        # The side stream of the silhouette / smoothness terms must not share a hardware queue with the caller's stream: a process gets four
        # queues, further streams are multiplexed onto them, and a stream that lands in the caller's queue sits BEHIND the barrier with
        # which the caller's stream waits for the refraction term -- the terms then run one after the other whatever the streams say
        # (rocprofv3: the fresh torch stream shared queue 1 with stream 0).  The library's second pipeline stream is idle during a call of
        # this size and has a queue of its own.
        self.side = None
        if concurrent:
            import ctypes
            h = ctypes.c_void_p()
            rc = _lib.lib().drt_internal_stream(scene.optix_mesh._h, 2, ctypes.byref(h))
            self.side = torch.cuda.ExternalStream(h.value, device=dev) if rc == 0 and h.value else torch.cuda.Stream(device=dev)
        self._w = None

    def step(self):
        """Runs the iteration; returns (weighted total, parts [ray, vh, sm]) as device tensors of THIS iteration (views of
        buffers that the next call overwrites: read them, or clone them, before stepping again)."""
        from . import diffrender as R
        from .optix_mesh import _stream
        lib, check, ptr = self._lib.lib(), self._lib.check, self._lib.ptr
        scene, hp, data = self.scene, self.hp, self.data
        dev = self.init_vertices.device
        with torch.no_grad(), torch.cuda.device(dev):
            vertices = self.init_vertices + self.parameter
            scene.update_verticex(vertices)
            from . import det
            if det.on():
                # deterministic mode: the three terms accumulate into fixed-point cells (drt_amd/det.py), converted once all of them are in
                if getattr(self, "_g_acc", None) is None:
                    self._g_acc = torch.zeros(self.grads.numel() * 3, dtype=torch.int64, device=dev)
                    self._l_acc = torch.zeros(3 * 3, dtype=torch.int64, device=dev)
                self._g_acc.zero_()
                self._l_acc.zero_()
                per = self.grads[0].numel() * 3
                g_ptr = [self._g_acc[k * per:].data_ptr() for k in range(3)]
                l_ptr = [self._l_acc[3 * k:].data_ptr() for k in range(3)]
            else:
                self.grads.zero_()
                self.losses.zero_()
                g_ptr = [self.grads[k].data_ptr() for k in range(3)]
                l_ptr = [self.losses[k:].data_ptr() for k in range(3)]
            h = scene.optix_mesh._h
            main = torch.cuda.current_stream()
            if self.side is not None:
                self.side.wait_stream(main)
            if hp["ray_w"] != 0:
                target, valid, _, origin, ray_dir, _ = data.get_view(next(self.ray_view))
                n = origin.shape[0]
                o, d, sp = R._f64c(origin, "origin"), R._f64c(ray_dir, "ray_dir"), R._f64c(target, "screen_pixel")
                va = R._flag_bytes(valid, "valid", n)
                grid = R._grid_cache(origin, ray_dir, n, *R._tile_hint(n)) if origin.is_contiguous() and ray_dir.is_contiguous() else (0, None)
                R._arm_seed(h, grid, n)
                check(lib.drt_render_ray_loss_fused(h, vertices.data_ptr(), o.data_ptr(), d.data_ptr(), sp.data_ptr(), va.data_ptr(), n,
                                                    float(R.intIOR), float(R.extIOR), l_ptr[0], g_ptr[0], None,
                                                    *R._tile_hint(n), grid[0], ptr(grid[1]), _stream()))
            ctx = torch.cuda.stream(self.side) if self.side is not None else torch.no_grad()
            with ctx:
                if hp["sm_w"] != 0:        # (first: it needs no tree, so it runs while the build finishes; the silhouette probes wait for the tree)
                    check(lib.drt_sm_loss_fused(vertices.data_ptr(), scene.E2F.data_ptr(), scene.E2F.shape[0], l_ptr[2], g_ptr[2], _stream()))
                if hp["vh_w"] != 0:
                    import ctypes
                    k = self.N_SILHOUETTE_VIEWS
                    cams, orgs, softs = (ctypes.c_void_p * k)(), (ctypes.c_void_p * k)(), (ctypes.c_void_p * k)()
                    keep = []
                    for j in range(k):
                        _, _, soft_mask, origin, _, camera_M = data.get_view(next(self.silh_view))
                        cam, o3, sm_ = R.pack_camera(camera_M), R._f64c(origin[0], "origin"), R._f64c(soft_mask, "soft_mask")
                        keep += [cam, o3, sm_]
                        cams[j], orgs[j], softs[j] = cam.data_ptr(), o3.data_ptr(), sm_.data_ptr()
                    check(lib.drt_vh_loss_fused(h, vertices.data_ptr(), scene.Edges.data_ptr(), scene.E2F.data_ptr(), scene.E2F.shape[0], k,
                                                cams, orgs, softs, int(data.resx), int(data.resy), 1, l_ptr[1], g_ptr[1], _stream()))
            if self.side is not None:
                main.wait_stream(self.side)
            if det.on():
                det.value_into(self._g_acc, self.grads)
                det.value_into(self._l_acc, self.losses)
            w = loss_weights(hp, data.resy, scene.mean_len)
            if self._w is None or self._w[0] != w:
                self._w = (w, torch.tensor(w, dtype=Float, device=dev))
            wv = self._w[1]
            # d total / d vertices (= d total / d parameter) = the weighted sum of the three terms' gradients, limit_hook, SGD(nesterov): one kernel
            check(lib.drt_limit_sgd_step3(self.parameter.data_ptr(), self.total_grad.data_ptr(), ptr(self.buf), self.parameter.numel(), self.lr,
                                          self.momentum, 1, int(self.first), 1.0, self.grads.data_ptr(), wv.data_ptr(), self.losses.data_ptr(),
                                          self.total.data_ptr(), _stream()))
            total = self.total
            self.first = False
            self._vertices = vertices          # (alive until the next step: kernels enqueued above read it)
        return total, self.losses


def setup_opt(scene, lr, HyperParams, hook=True, fused=False):
    """``fused=True`` (needs ``hook=False``): a FusedLimitSGD, which applies limit_hook itself inside its one-kernel step."""
    init_vertices = scene.vertices.detach().clone()
    parameter = torch.zeros(init_vertices.shape, dtype=Float, requires_grad=True, device=init_vertices.device)
    if fused:
        assert not hook, "FusedLimitSGD applies the limit itself: build it with hook=False"
        return init_vertices, parameter, FusedLimitSGD(parameter, lr, HyperParams["momentum"], nesterov=True)
    if hook:
        parameter.register_hook(limit_hook)
    # foreach=False: the multi-tensor kernels of the default implementation take ~35 us each for this ONE small tensor
    # (four per step: 0.14 ms of a 1.2 ms iteration); the single-tensor path does the same arithmetic in ~5 us ops
    opt = torch.optim.SGD([parameter], lr=lr, momentum=HyperParams["momentum"], nesterov=True, foreach=False)
    return init_vertices, parameter, opt


def optimize(scene, data, HyperParams, remesh="isotropic", output=True, fused=False):
    """The reference's pass / iteration loop (optim.py:190-215) for an existing scene and data object.
    ``remesh``: "isotropic" (default) re-tessellates to ``remesh_len`` before every pass like the reference's
    ``meshlabserver.remesh`` (optim.py:195), with the device remesher of drt_amd.remesh_gpu ("isotropic-host": the sequential
    host version of drt_amd.remesh); ``None`` keeps the topology; or any callable ``remesh(scene, remesh_len)``."""
    if remesh == "isotropic":               # on the device (drt_amd.remesh_gpu); "isotropic-host": the sequential host version, its checker
        from .remesh_gpu import GpuMeshlabserver
        remesh = GpuMeshlabserver().remesh
    elif remesh == "isotropic-host":
        from .remesh import Meshlabserver
        remesh = Meshlabserver().remesh
    Render.intIOR = HyperParams["IOR"]
    Render.resy, Render.resx = data.resy, data.resx
    loss_calculator = Loss_calculator(scene, data, HyperParams, fused=fused)
    start_time = time.time()
    history = []
    for i_pass in range(HyperParams["Pass"]):
        if HyperParams["Pass"] > 1:
            remesh_len = interp_R(HyperParams["start_len"], HyperParams["end_len"], i_pass, HyperParams["Pass"])
            lr = interp_R(HyperParams["start_lr"], HyperParams["lr_decay"] * HyperParams["start_lr"], i_pass, HyperParams["Pass"])
        else:
            remesh_len, lr = HyperParams["start_len"], HyperParams["start_lr"]
        if output:
            print(f"remesh_len {remesh_len:g} lr {lr:g}")
        if remesh is not None:
            remesh(scene, remesh_len)
        if fused:      # the one-pass terms without the autograd graph around them (same arithmetic, a third of the host work)
            stepper = FusedIteration(scene, data, HyperParams, lr)
            stepper.ray_view, stepper.silh_view = loss_calculator.ray_view, loss_calculator.silh_view      # one view schedule across passes
            for it in range(HyperParams["Iters"]):
                total, parts = stepper.step()
                if it % 100 == 0:
                    if output:
                        print(f"Iteration {it}: {loss_string(tuple(parts))} maxgrad={stepper.total_grad.abs().max():g}")
                    history.append(float(total))
            continue
        init_vertices, parameter, opt = setup_opt(scene, lr, HyperParams)
        for it in range(HyperParams["Iters"]):
            opt.zero_grad()
            vertices = init_vertices + parameter
            scene.update_verticex(vertices)
            loss, parts = loss_calculator.all_loss()
            loss.backward()
            if it % 100 == 0 and output:
                print(f"Iteration {it}: {loss_string(parts)} maxgrad={parameter.grad.abs().max():g}")
            history.append(float(loss.detach())) if (it % 100 == 0) else None
            opt.step()
    if output:
        print(f"optimize time : {time.time() - start_time}")
    return scene, history


def local_loss_backward(scene, local_views, init_vertices, parameter, ray_w, fused=False):
    """The rank-local part of a full-batch step: rebuild, every local view's ray loss, backward.  Leaves d(ray_w * loss)/d
    parameter in ``parameter.grad`` (None for a rank without views) and returns the unweighted loss."""
    vertices = init_vertices + parameter
    scene.update_verticex(vertices)
    parts = []
    for view in local_views:
        # (target, valid, origin, ray_dir) -- the tensors of Data.get_view -- or (target, valid, handle) with handle = scene.bind_rays(...):
        # the explicit form of "these rays are constants" (diffrender.RayBinding)
        target, valid, origin = view[0], view[1], view[2]
        ray_dir = view[3] if len(view) > 3 else None
        if fused:
            parts.append(scene.ray_loss_fused(origin, ray_dir, target, valid))
        else:
            out_ori, out_dir, mask = scene.render_transparent(origin, ray_dir)
            parts.append(Render.ray_loss(out_ori, out_dir, mask, target, valid))
    if not parts:
        return torch.zeros((), dtype=Float, device=vertices.device)
    loss = parts[0] if len(parts) == 1 else torch.stack(parts).sum()
    if loss.requires_grad:
        # d(ray_w * loss): the weight goes in as the seed of the backward pass (one cached scalar) instead of a multiplication
        # node -- the step is a chain of small launches, and each elementwise kernel of the loss arithmetic is ~5 us of it
        loss.backward(_seed(ray_w, loss))
    return loss


_SEEDS = {}


def _seed(w, like):
    key = (float(w), like.dtype, like.device)
    t = _SEEDS.get(key)
    if t is None:
        if len(_SEEDS) > 64:
            _SEEDS.clear()
        t = _SEEDS[key] = torch.full((), float(w), dtype=like.dtype, device=like.device)
    return t


def full_batch_step(scene, local_views, init_vertices, parameter, opt, ray_w, fused=False):
    """One step over ALL views of this rank (BASELINE.json's '72 views forward+backward per iter'):
    rebuild, per-view ray loss, backward, ONE all-reduce of grad[V,3], limit_hook, SGD step.

    ``parameter`` must NOT carry the gradient hook (``setup_opt(..., hook=False)``): the reference clamps the
    gradient of the whole step (optim.py:155-162), i.e. AFTER the sum over views, so here the clamp follows the
    all-reduce; a hook would clamp every rank's partial sum first.  A rank without views still takes part in the
    all-reduce (with zeros) and applies the same update, so parameters stay identical on every rank."""
    hooks = getattr(parameter, "_backward_hooks", None)
    if hooks:
        raise RuntimeError("full_batch_step: `parameter` has a gradient hook; build it with setup_opt(..., hook=False) "
                           "(limit_hook is applied here, after the all-reduce)")
    opt.zero_grad(set_to_none=True)
    from . import det
    sink = det.begin_sink()                 # deterministic mode: the refraction terms leave their exact accumulators here instead of float64 gradients
    try:
        loss = local_loss_backward(scene, local_views, init_vertices, parameter, ray_w, fused)
    finally:
        entries = det.end_sink()
    if sink is not None:
        # ONE exchange, exact: the 128-bit sums of every call of every rank are added as integers and converted once -- N GPUs, one GPU, any
        # split of the views give the same bits (what autograd did not route through the sink, nothing in this step, is added as before)
        g = det.collect(entries, parameter, ddist.allreduce_sum_, _seed(ray_w, parameter))
        if not (fused or (Render.EAGER_LOSS_GRAD and Render.SPARSE_LOSS_GRAD)):
            # (module switches that route ray_loss's gradient through autograd as float64 instead: the same on every rank, so every rank
            # joins this second collective or none does)
            g = g + ddist.allreduce_sum_(parameter.grad if parameter.grad is not None else torch.zeros_like(parameter))
        elif parameter.grad is not None:
            raise RuntimeError("full_batch_step (deterministic mode): a gradient reached `parameter` outside the exact sink")
    else:
        g = parameter.grad if parameter.grad is not None else torch.zeros_like(parameter)   # a rank with no views
        ddist.allreduce_sum_(g)             # the only exchange of the step
    parameter.grad = g if getattr(opt, "applies_limit", False) else limit_hook(g)     # clamp after the sum over views, as on one GPU
    opt.step()
    return loss
