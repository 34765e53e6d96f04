"""The silhouette (visual-hull) and smoothness branches behind ``Scene`` (reference DiffRender.py:149-267, 440-479; optim.py:67-89):
per-edge kernels of libdrt_hip.so behind ``torch.autograd.Function``s, and the LAZY stand-ins that let the reference's own loop
(``silhouette_edge`` -> ``primary_visibility`` -> the loss expression of optim.py:78, summed over eight views) run as one fused launch
without a host round trip per view.  ``drt_amd.diffrender`` re-exports everything here; the switches (LAZY_SILHOUETTE, LAZY_VISIBILITY)
live there, next to the Scene methods that read them."""
from __future__ import annotations

import ctypes
import weakref

import torch

from . import _lib, det
from ._util import _f64c, _stats
from .optix_mesh import _stream, _on

_camera_cache = {}


def pack_camera(camera_M):
    """camera_M = (R 4x4, K 3x3, R^-1, K^-1) -> one float64 [50] device tensor (layout of drt_edge.h Camera).
    Cached per camera tuple (keyed on the identity and in-place version of its four tensors): a capture's
    cameras are constants and the silhouette loss packs eight of them per iteration."""
    key = tuple(id(t) for t in camera_M)
    ent = _camera_cache.get(key)
    if ent is not None and all(r() is t and ver == t._version for r, ver, t in zip(ent[0], ent[1], camera_M)):
        return ent[2]
    R, K, R_inverse, K_inverse = camera_M
    packed = torch.cat([R.reshape(-1), K.reshape(-1), R_inverse.reshape(-1), K_inverse.reshape(-1)]).to(torch.float64).contiguous()
    if len(_camera_cache) > 4096:
        _camera_cache.clear()
    _camera_cache[key] = (tuple(weakref.ref(t) for t in camera_M), tuple(t._version for t in camera_M), packed)
    return packed




class SilhouetteEdges:
    """What ``Scene.silhouette_edge`` returns: the int64 [Es,2] tensor ``Edges[flags]`` of the reference (DiffRender.py:445-457), materialised
    only when somebody looks at it.  The reference's loop hands it straight to ``primary_visibility`` (optim.py:76-77), which here reads
    the per-edge flags on the device instead -- so that the boolean-mask indexing, a device->host synchronisation per silhouette view, never
    happens.  Anything else (indexing, ``len``, ``.shape``, torch functions, attribute access) sees the materialised tensor.
    Round 6: the FLAGS are lazy too (``scene`` / ``origin`` given instead of ``flags``): when the pair of the view ends up in the summed
    silhouette term (LazySum below) the fused kernel finds them itself, and the per-view flag launch never happens."""

    def __init__(self, edges, flags, scene=None, origin=None):
        self._edges, self._flag_t, self._t = edges, flags, None
        self._scene, self._origin = scene, origin
        self._epoch = scene._epoch if scene is not None else None

    @property
    def _flags(self):
        if self._flag_t is None:
            scene = self._scene
            _check_epoch(scene, self._epoch, "silhouette_edge")
            v = _f64c(scene.vertices.detach(), "vertices")
            o = _f64c(self._origin.detach(), "origin")
            n = scene.E2F.shape[0]
            flags = torch.empty(n, dtype=torch.uint8, device=v.device)
            with _on(v.device):
                _lib.check(_lib.lib().drt_silhouette_flags(v.data_ptr(), scene.E2F.data_ptr(), n, o.data_ptr(), flags.data_ptr(), _stream()))
            self._flag_t = flags
        return self._flag_t

    def tensor(self):
        if self._t is None:
            self._t = self._edges[self._flags.view(torch.bool)]
        return self._t

    def __getattr__(self, name):                     # (only reached for names this object does not define itself)
        if name.startswith("_"):                     # (its own fields, before __init__ has run -- copy / pickle probe for them: no recursion)
            raise AttributeError(name)
        return getattr(self.tensor(), name)

    def __getitem__(self, k):
        return self.tensor()[k]

    def __len__(self):
        return len(self.tensor())

    def __iter__(self):
        return iter(self.tensor())

    def __repr__(self):
        return f"SilhouetteEdges({self.tensor()!r})"

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        un = lambda a: a.tensor() if isinstance(a, SilhouetteEdges) else a
        args = tuple(un(a) if not isinstance(a, (list, tuple)) else type(a)(un(b) for b in a) for a in args)
        kwargs = {k: un(v) for k, v in (kwargs or {}).items()}
        return func(*args, **kwargs)


def _check_epoch(scene, epoch, what):
    """A lazy object stands for a result on the mesh state of the call that made it: once the scene has moved on it cannot be computed any more."""
    if scene is not None and scene._epoch != epoch:
        raise RuntimeError(f"drt_amd.diffrender: the lazy result of {what} is being evaluated after the scene's vertices / mesh changed "
                           "(update_verticex / update_mesh since that call): look at it before the next update, or set LAZY_VISIBILITY = "
                           "LAZY_SILHOUETTE = False (DRT_LAZY_VISIBILITY=0) to get plain tensors right away")




class SampleSet:
    """The silhouette samples of one ``primary_visibility`` call: for ALL E unique edges an index row, f = hit(+) - hit(-) and a `keep` flag
    (|f| > 1e-5 and inside the view: DiffRender.py:244, 478).  The reference returns the compacted (index [M,2], output [M]); that
    compaction is ``materialise()`` -- taken by anything that looks at the pair as tensors.  The one expression the reference's loop applies
    to the pair (optim.py:78) is recognised step by step by the lazy objects below and becomes a ``LazyTerm``; the reference then SUMS the
    terms of its eight views (optim.py:72-80), which ``LazySum`` serves with ONE fused launch over all of them (drt_vh_loss_fused: flags,
    probe rays, terms and vertex gradient) -- nothing at all is enqueued per view.  A term that is used on its own is evaluated over the
    uncompacted rows of its view (``term_now``): same samples, same terms, a float64 sum in another order.
    Everything deferred is tied to the scene state of THIS call (`_check_epoch`)."""

    def __init__(self, scene, vertices, sil, camera_M, origin, detach_depth, res_x, res_y):
        self.scene, self.vertices, self.sil, self.camera_M, self.origin = scene, vertices, sil, camera_M, origin
        self.detach_depth, self.res_x, self.res_y = detach_depth, res_x, res_y
        self._epoch = scene._epoch
        self._v_version = vertices._version
        self._ran = False
        self._pair = None
        _stats["visibility_lazy"] += 1

    def _check(self, what):
        _check_epoch(self.scene, self._epoch, what)
        if self.vertices._version != self._v_version:
            raise RuntimeError("drt_amd.diffrender: the vertices tensor was modified in place between primary_visibility and the evaluation of its lazy result")

    def _run(self):
        """The sampling kernel of this view (projection, probe rays), once."""
        if self._ran:
            return
        self._check("primary_visibility")
        sil = self.sil
        if isinstance(sil, SilhouetteEdges):
            sil = (sil._edges, sil._flags)
        self.v = _f64c(self.vertices.detach(), "vertices")
        edges, flags = sil
        self.edges = edges.contiguous()
        assert self.edges.dtype == torch.long and self.edges.dim() == 2 and self.edges.shape[1] == 2
        self.cam = pack_camera(self.camera_M)
        o = _f64c(self.origin.detach(), "origin")
        n = self.edges.shape[0]
        dev = self.v.device
        w1 = torch.empty(5 * n, dtype=torch.uint8, device=dev)           # (f | keep in one allocation)
        self.index = torch.empty((n, 2), dtype=torch.long, device=dev)
        self.f, self.keep = w1[:4 * n].view(torch.float32), w1[4 * n:]
        self._flags = flags
        with _on(dev):
            _lib.check(_lib.lib().drt_edge_sample_forward(self.scene.optix_mesh._h, self.v.data_ptr(), self.edges.data_ptr(), n, self.cam.data_ptr(),
                                                          o.data_ptr(), self.index.data_ptr(), self.f.data_ptr(), self.keep.data_ptr(), self.res_x, self.res_y,
                                                          _lib.ptr(flags), _stream()))
        self._ran = True

    def materialise(self):
        """(index int64 [M,2], output float32 [M]) as the reference returns them, differentiable w.r.t. the vertices."""
        if self._pair is None:
            self._run()
            _stats["visibility_materialised"] += 1
            self._pair = _EdgeSample.apply(self.vertices, (self.edges, self._flags), self.camera_M, self.origin, self.scene, self.detach_depth, self.res_x, self.res_y,
                                           (self.index, self.f, self.keep))
        return self._pair

    def term(self, image):
        """sum |image[y, x] - output| over the samples (optim.py:78), deferred: a LazyTerm."""
        return LazyTerm(self, image)

    def term_now(self, image):
        """... evaluated for this view alone: a scalar differentiable w.r.t. the vertices."""
        self._run()
        _stats["visibility_term_in_place"] += 1
        return _VhTermLazy.apply(self.vertices, self, image)


class _LazyTensor:
    """A stand-in that behaves as the tensor ``self.tensor()`` for everything it does not recognise."""

    # isinstance(x, torch.Tensor) / torch.is_tensor(x) hold for the stand-ins: callers that branch on them take their tensor path, whose
    # operations reach __torch_function__ / the delegating operators below and see the materialised tensor
    __class__ = property(lambda self: torch.Tensor)

    _OWN = ("_ss", "_col", "_image", "_stage", "_terms", "_t")

    def __getattr__(self, name):
        if name in _LazyTensor._OWN or (name.startswith("__") and name.endswith("__")):
            raise AttributeError(name)                # (its own fields before __init__ has run -- copy / pickle probe for them -- and dunder probes)
        return getattr(self.tensor(), name)           # incl. private tensor attributes (`_version`, `_base`, ...)

    def __getitem__(self, k):
        return self.tensor()[_unlazy(k)]

    def __setitem__(self, k, v):
        self.tensor()[_unlazy(k)] = _unlazy(v)        # (the materialised tensor is cached: later reads see the assignment, as with the reference's tensor)

    def __len__(self):
        return len(self.tensor())

    def __iter__(self):
        return iter(self.tensor())

    def __repr__(self):
        return f"{type(self).__name__}({self.tensor()!r})"

    def __format__(self, spec):
        return format(self.tensor(), spec)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        if func is torch.Tensor.__getitem__ and len(args) == 2 and isinstance(args[1], tuple) and len(args[1]) == 2:
            # image[index[:, 1], index[:, 0]]  (optim.py:78)
            img, (iy, ix) = args
            if (type(iy) is LazyColumn and type(ix) is LazyColumn and iy._ss is ix._ss and (iy._col, ix._col) == (1, 0) and type(img) is torch.Tensor
                    and img.dim() == 2 and img.shape == (iy._ss.res_y, iy._ss.res_x) and img.dtype == torch.float64 and img.is_cuda and img.is_contiguous()
                    and not img.requires_grad and iy._ss._pair is None):
                return LazyGather(iy._ss, img)
        un = _unlazy
        return func(*un(args), **{k: un(v) for k, v in (kwargs or {}).items()})


def _delegate(name):
    def op(self, *args, **kwargs):
        return getattr(self.tensor(), name)(*_unlazy(args), **{k: _unlazy(v) for k, v in kwargs.items()})
    op.__name__ = name
    return op


# operators are looked up on the TYPE, not through __getattr__: every one a tensor has goes to the materialised tensor
for _name in ("add radd iadd sub rsub isub mul rmul imul truediv rtruediv floordiv rfloordiv mod rmod pow rpow matmul rmatmul neg pos abs invert and rand or ror xor rxor "
              "lshift rshift eq ne lt le gt ge bool float int index contains").split():
    if _name in ("iadd", "isub", "imul"):
        # (in-place on a stand-in: the out-of-place result -- Python rebinds the name, like `vh_loss += term` on the reference's int 0)
        setattr(_LazyTensor, f"__{_name}__", _delegate(f"__{_name[1:]}__"))
    elif not hasattr(_LazyTensor, f"__{_name}__") or _name in ("eq", "ne", "lt", "le", "gt", "ge"):
        setattr(_LazyTensor, f"__{_name}__", _delegate(f"__{_name}__"))
_LazyTensor.__hash__ = lambda self: id(self)


def _is_lazy(a):
    return isinstance(type(a), type) and issubclass(type(a), _LazyTensor)


def _unlazy(a):
    if _is_lazy(a):
        return a.tensor()
    if type(a) in (list, tuple):
        return type(a)(_unlazy(b) for b in a)
    return a


class LazyIndex(_LazyTensor):
    def __init__(self, ss):
        self._ss = ss

    def tensor(self):
        return self._ss.materialise()[0]

    def __getitem__(self, k):
        if (type(k) is tuple and len(k) == 2 and type(k[0]) is slice and k[0] == slice(None) and type(k[1]) is int and k[1] in (0, 1)
                and self._ss._pair is None):
            return LazyColumn(self._ss, k[1])
        return self.tensor()[_unlazy(k)]


class LazyColumn(_LazyTensor):
    def __init__(self, ss, col):
        self._ss, self._col = ss, col

    def tensor(self):
        return self._ss.materialise()[0][:, self._col]


class LazyOutput(_LazyTensor):
    def __init__(self, ss):
        self._ss = ss

    def tensor(self):
        return self._ss.materialise()[1]


class LazyGather(_LazyTensor):
    """image[index[:, 1], index[:, 0]]"""

    def __init__(self, ss, image):
        self._ss, self._image = ss, image

    def tensor(self):
        idx = self._ss.materialise()[0]
        return self._image[idx[:, 1], idx[:, 0]]

    def __sub__(self, other):
        if type(other) is LazyOutput and other._ss is self._ss and self._ss._pair is None:
            return LazyDiff(self._ss, self._image, 0)
        return self.tensor() - _unlazy(other)


class LazyDiff(_LazyTensor):
    """image[...] - output (stage 0), its .abs() (stage 1); .sum() of stage 1 is SampleSet.term."""

    def __init__(self, ss, image, stage):
        self._ss, self._image, self._stage = ss, image, stage

    def tensor(self):
        idx, out = self._ss.materialise()
        d = self._image[idx[:, 1], idx[:, 0]] - out
        return d.abs() if self._stage else d

    def abs(self):
        if self._stage == 0 and self._ss._pair is None:
            return LazyDiff(self._ss, self._image, 1)
        return self.tensor().abs()

    def sum(self, *args, **kwargs):
        if self._stage == 1 and not args and not kwargs and self._ss._pair is None:
            return self._ss.term(self._image)
        return self.tensor().sum(*args, **kwargs)


def _zero_number(x):
    return type(x) in (int, float) and x == 0


class LazyTerm(_LazyTensor):
    """The silhouette term of ONE view, not evaluated yet.  ``0 + term`` / ``term + term`` / ``total += term`` -- the accumulation of
    reference optim.py:71-78 -- builds a LazySum; any other use evaluates this term by itself."""

    def __init__(self, ss, image):
        self._ss, self._image, self._t = ss, image, None

    def tensor(self):
        if self._t is None:
            self._t = self._ss.term_now(self._image)
        return self._t

    def _join(self, other, swapped=False):
        mine = [self]
        if type(other) is LazyTerm:
            theirs = [other]
        elif type(other) is LazySum:
            theirs = other._terms
        elif _zero_number(other):
            theirs = []
        else:
            return None
        return LazySum((theirs + mine) if swapped else (mine + theirs))

    def __add__(self, other):
        s = self._join(other)
        return s if s is not None else self.tensor() + _unlazy(other)

    def __radd__(self, other):
        s = self._join(other, swapped=True)
        return s if s is not None else _unlazy(other) + self.tensor()

    __iadd__ = __add__


class LazySum(_LazyTensor):
    """The sum of several views' silhouette terms (reference optim.py:71-80), evaluated by ONE ``drt_vh_loss_fused`` launch pair over all of
    them when somebody needs the value -- the weighting in ``all_loss`` (optim.py:128), ``.backward()``, a format string."""

    def __init__(self, terms):
        self._terms, self._t = list(terms), None

    def _join(self, other, swapped=False):
        if type(other) is LazyTerm:
            theirs = [other]
        elif type(other) is LazySum:
            theirs = other._terms
        elif _zero_number(other):
            theirs = []
        else:
            return None
        if self._t is not None:
            return None
        return LazySum((theirs + self._terms) if swapped else (self._terms + theirs))

    def __add__(self, other):
        s = self._join(other)
        return s if s is not None else self.tensor() + _unlazy(other)

    def __radd__(self, other):
        s = self._join(other, swapped=True)
        return s if s is not None else _unlazy(other) + self.tensor()

    __iadd__ = __add__

    def tensor(self):
        if self._t is not None:
            return self._t
        # the terms that can share one fused launch: nothing of theirs has run yet, same scene state, same vertices, same image size
        first = next((t for t in self._terms if t._t is None and not t._ss._ran), None)
        batch, rest = [], []
        for t in self._terms:
            ss = t._ss
            ok = (first is not None and t._t is None and not ss._ran and ss._pair is None and ss.scene is first._ss.scene and ss.vertices is first._ss.vertices
                  and (ss.res_x, ss.res_y, ss.detach_depth) == (first._ss.res_x, first._ss.res_y, first._ss.detach_depth)
                  and isinstance(ss.sil, SilhouetteEdges) and ss.sil._t is None and ss.sil._edges is ss.scene.Edges)
            (batch if ok else rest).append(t)
        parts = []
        if len(batch) >= 2:
            ss0 = batch[0]._ss
            for t in batch:
                t._ss._check("primary_visibility")
            flat = []
            for t in batch:
                flat += [pack_camera(t._ss.camera_M), t._ss.origin, t._image]
            _stats["visibility_terms_fused"] += len(batch)
            parts.append(_VhLossFused.apply(ss0.vertices, ss0.scene, ss0.res_x, ss0.res_y, bool(ss0.detach_depth), *flat))
        else:
            rest = self._terms
        parts += [t.tensor() for t in rest]
        total = parts[0]
        for p in parts[1:]:
            total = total + p
        self._t = total
        return total


def force(x):
    """The tensor a lazy stand-in stands for (evaluated now, on the current stream); anything else is returned as it is."""
    return x.tensor() if _is_lazy(x) or type(x) is SilhouetteEdges else x


class _VhTermLazy(torch.autograd.Function):
    """sum over the kept samples of |image[y, x] - 0.5| as a function of the vertices (SampleSet.term_now)."""

    @staticmethod
    def forward(ctx, vertices, ss, image):
        n = ss.edges.shape[0]
        dev = ss.v.device
        loss = det.scalar(dev)
        dterm = torch.empty(n, dtype=torch.float64, device=dev)     # d term / d output per row: THIS call's own (a second term on the same samples
        with _on(dev):                                              #  with another image must not overwrite it before the first backward runs)
            _lib.check(_lib.lib().drt_vh_term(ss.index.data_ptr(), ss.keep.data_ptr(), n, image.data_ptr(), ss.res_x, ss.res_y,
                                              loss.data_ptr(), dterm.data_ptr(), _stream()))
        ctx.ss = ss
        ctx.save_for_backward(dterm)
        return det.value(loss)

    @staticmethod
    def backward(ctx, g_loss):
        (dterm,) = ctx.saved_tensors
        ss = ctx.ss
        grad_v = det.acc(ss.v)
        # (the reference's `output` is float32: the incoming gradient reaches primary_edge_sample.backward rounded to float32, DiffRender.py:251, 263-267)
        g = g_loss if (g_loss.dtype == torch.float64 and g_loss.is_cuda and g_loss.numel() == 1) else g_loss.to(device=ss.v.device, dtype=torch.float64).reshape(1)
        with _on(ss.v.device):
            _lib.check(_lib.lib().drt_edge_sample_backward_term(ss.v.data_ptr(), ss.edges.data_ptr(), ss.edges.shape[0], ss.cam.data_ptr(), ss.f.data_ptr(),
                                                                dterm.data_ptr(), g.data_ptr(), int(ss.detach_depth), grad_v.data_ptr(), _stream()))
        return det.value(grad_v, ss.v), None, None


class _Dihedral(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vertices, E2F):
        v = _f64c(vertices.detach(), "vertices")
        e2f = E2F.contiguous()
        assert e2f.dtype == torch.long and e2f.shape[1:] == (2, 3)
        n = e2f.shape[0]
        out = torch.empty(n, dtype=torch.float64, device=v.device)
        with _on(v.device):
            _lib.check(_lib.lib().drt_dihedral_forward(v.data_ptr(), e2f.data_ptr(), n, out.data_ptr(), _stream()))
        ctx.save_for_backward(v, e2f)
        return out

    @staticmethod
    def backward(ctx, g_cos):
        v, e2f = ctx.saved_tensors
        grad_v = det.acc(v)
        g = _f64c(g_cos, "grad")
        with _on(v.device):
            _lib.check(_lib.lib().drt_dihedral_backward(v.data_ptr(), e2f.data_ptr(), e2f.shape[0], g.data_ptr(), grad_v.data_ptr(), _stream()))
        return det.value(grad_v, v), None


class _SmLossFused(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vertices, E2F):
        v = _f64c(vertices.detach(), "vertices")
        e2f = E2F.contiguous()
        loss = det.scalar(v.device)
        grad_v = det.acc(v)
        with _on(v.device):
            _lib.check(_lib.lib().drt_sm_loss_fused(v.data_ptr(), e2f.data_ptr(), e2f.shape[0], loss.data_ptr(), grad_v.data_ptr(), _stream()))
        ctx.save_for_backward(det.value(grad_v, v))
        return det.value(loss)

    @staticmethod
    def backward(ctx, g_loss):
        (grad_v,) = ctx.saved_tensors
        return grad_v * g_loss, None


class _EdgeSample(torch.autograd.Function):
    """primary_visibility's projection + primary_edge_sample (reference DiffRender.py:189-267, 464-475)
    as one function of the vertices."""

    @staticmethod
    def forward(ctx, vertices, sil_edges, camera_M, origin, scene, detach_depth, res_x, res_y, computed=None):
        v = _f64c(vertices.detach(), "vertices")
        flags = None
        if isinstance(sil_edges, tuple):
            # straight from silhouette_edge: every unique edge with its flag, no compaction (and no host round trip) in between
            edges, flags = sil_edges[0].contiguous(), sil_edges[1]
        else:
            edges = sil_edges.contiguous()
        assert edges.dtype == torch.long and edges.dim() == 2 and edges.shape[1] == 2
        cam = pack_camera(camera_M)
        o = _f64c(origin.detach(), "origin")
        n = edges.shape[0]
        if computed is not None:           # (a SampleSet that is being materialised: the kernel ran when primary_visibility was called)
            index, f, keep = computed
        else:
            index = torch.empty((n, 2), dtype=torch.long, device=v.device)
            f = torch.empty(n, dtype=torch.float32, device=v.device)           # (the kernel writes f and keep of every row, 0 for unflagged edges)
            keep = torch.empty(n, dtype=torch.uint8, device=v.device)
            with _on(v.device):
                _lib.check(_lib.lib().drt_edge_sample_forward(scene.optix_mesh._h, v.data_ptr(), edges.data_ptr(), n, cam.data_ptr(),
                                                              o.data_ptr(), index.data_ptr(), f.data_ptr(), keep.data_ptr(), int(res_x), int(res_y),
                                                              _lib.ptr(flags), _stream()))
        # |f| > 1e-5 (DiffRender.py:244) and inside the view (DiffRender.py:478), decided by the kernel: ONE boolean index, one host sync.
        # (An ordered compaction by one block of our own in place of the library's three-launch select: 58 us against 34 -- not kept.)
        sel = torch.nonzero(keep).squeeze(1)             # (the host sync; the row numbers also serve the backward, which then needs none)
        index = index.index_select(0, sel)
        output = torch.full((sel.shape[0],), 0.5, device=v.device)   # float32, like the reference (DiffRender.py:251)
        ctx.mark_non_differentiable(index)
        ctx.save_for_backward(v, edges, cam, f, sel)
        ctx.detach_depth = detach_depth
        return index, output

    @staticmethod
    def backward(ctx, grad_index, grad_output):
        v, edges, cam, f, sel = ctx.saved_tensors
        grad_v = det.acc(v)
        g = grad_output if grad_output.dtype == torch.float32 and grad_output.is_contiguous() else grad_output.to(torch.float32).contiguous()
        with _on(v.device):
            # (the kept rows and their float32 gradients as they are: no zero-filled [Es] coefficient vector, cast and scatter per view)
            _lib.check(_lib.lib().drt_edge_sample_backward_rows(v.data_ptr(), edges.data_ptr(), edges.shape[0], cam.data_ptr(), f.data_ptr(),
                                                                sel.data_ptr(), sel.shape[0], g.data_ptr(), int(ctx.detach_depth),
                                                                grad_v.data_ptr(), _stream()))
        return det.value(grad_v, v), None, None, None, None, None, None, None, None, None


class _VhLossFused(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vertices, scene, res_x, res_y, detach_depth, *flat):
        v = _f64c(vertices.detach(), "vertices")
        loss = det.scalar(v.device)
        grad_v = det.acc(v)
        n = len(flat) // 3
        cams, orgs, softs = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)()
        keep = []
        for k in range(n):
            o = _f64c(flat[3 * k + 1].detach(), "origin")
            sm = _f64c(flat[3 * k + 2], "soft_mask")
            assert sm.numel() == res_x * res_y and o.numel() == 3 and flat[3 * k].numel() == 50
            keep += [o, sm]
            cams[k], orgs[k], softs[k] = flat[3 * k].data_ptr(), o.data_ptr(), sm.data_ptr()
        edges, e2f = scene.Edges, scene.E2F
        with _on(v.device):
            _lib.check(_lib.lib().drt_vh_loss_fused(scene.optix_mesh._h, v.data_ptr(), edges.data_ptr(), e2f.data_ptr(), e2f.shape[0], n,
                                                    cams, orgs, softs, res_x, res_y, int(bool(detach_depth)), loss.data_ptr(), grad_v.data_ptr(), _stream()))
        ctx.save_for_backward(det.value(grad_v, v))
        ctx.n_in = len(flat)
        return det.value(loss)

    @staticmethod
    def backward(ctx, g_loss):
        (grad_v,) = ctx.saved_tensors
        # The reference's silhouette samples are a float32 tensor (`output`, torch's default dtype, DiffRender.py:251): autograd casts the
        # incoming d loss / d output to float32 before primary_edge_sample.backward multiplies it in (DiffRender.py:263-267).  The drop-in
        # pair does the same by construction; here the scalar is rounded the same way (tests/test_gpu_trajectory.py).
        return (grad_v * g_loss.to(torch.float32).to(torch.float64), None, None, None, None) + (None,) * ctx.n_in
