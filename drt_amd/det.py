"""Deterministic (order-independent) accumulation of gradients and losses: the Python side of ``drt_deterministic`` (include/drt_hip.h).

    DRT_DETERMINISTIC=1 python ...          or          drt_amd.det.enable(True)

In this mode every accumulation target handed to the library -- the vertex gradients of ``render_transparent`` / ``ray_loss`` / the
silhouette and smoothness terms (reference optim.py:59-130) and their scalar losses -- is an array of 24-byte fixed-point cells instead of
float64; the kernels add contributions as 128-bit integers (drt_amd/csrc/drt_fixed.h), ``value`` converts the exact sums to float64 with
one rounding.  Two runs, an eager step and a graph replay, one call and the same call cut into other sub-batches then give the same bits.
What is NOT covered: the ``stepwise`` route (Dintersect / refract_ray as torch ops: torch's own index kernels) and the caller's torch code.
"""
from __future__ import annotations

import torch

from . import _lib

_CELL_WORDS = 3            # DRT_FX_BYTES_PER_VALUE / 8
_state = [None]


def on() -> bool:
    if _state[0] is None:
        _state[0] = bool(_lib.lib().drt_deterministic(-1))
    return _state[0]


def enable(flag=True) -> bool:
    """Switch the process (library + this module) to deterministic accumulation; returns the previous mode."""
    was = bool(_lib.lib().drt_deterministic(1 if flag else 0))
    _state[0] = bool(flag)
    return was


def acc(like):
    """A zeroed accumulation target for a float64 tensor shaped ``like`` (a float64 tensor itself when the mode is off)."""
    if not on():
        return torch.zeros_like(like)
    return torch.zeros(like.numel() * _CELL_WORDS, dtype=torch.int64, device=like.device)


def scalar(device):
    if not on():
        return torch.zeros((), dtype=torch.float64, device=device)
    return torch.zeros(_CELL_WORDS, dtype=torch.int64, device=device)


def value(a, like=None, stream=None):
    """The float64 tensor an accumulation target stands for (shaped ``like``; a scalar without it)."""
    if a.dtype != torch.int64:
        return a
    from .optix_mesh import _stream, _on
    n = a.numel() // _CELL_WORDS
    out = torch.empty(like.shape if like is not None else (), dtype=torch.float64, device=a.device)
    with _on(a.device):
        _lib.check(_lib.lib().drt_fx_finalize(a.data_ptr(), n, out.data_ptr(), 0, _stream() if stream is None else stream))
    return out


def value_into(a, out, stream=None):
    """``out[...] = value(a)`` for a preallocated contiguous float64 ``out``."""
    from .optix_mesh import _stream, _on
    assert a.dtype == torch.int64 and out.dtype == torch.float64 and out.is_contiguous() and a.numel() == out.numel() * _CELL_WORDS
    with _on(a.device):
        _lib.check(_lib.lib().drt_fx_finalize(a.data_ptr(), out.numel(), out.data_ptr(), 0, _stream() if stream is None else stream))
    return out


# ---- the step's gradient as ONE exact sum over calls and ranks (drt_amd.optim.full_batch_step) -------------------------------------------
# While a sink is armed, the backward passes of the refraction terms do not hand autograd a float64 gradient: they leave their
# accumulator cells (and the scalar they would have been scaled by) here.  `collect` adds the cells of all calls (integers), all-reduces
# them over the ranks as integers (drt_fx_to_limbs: per-word sums that cannot overflow), converts ONCE and scales -- every rank converts
# the same integer, so N GPUs produce the bits one GPU produces.
SINK = None


def begin_sink():
    global SINK
    SINK = [] if on() else None
    return SINK


def end_sink():
    global SINK
    entries, SINK = SINK, None
    return entries or []


def collect(entries, like, allreduce, scale):
    """float64 gradient shaped ``like`` = ``scale`` x the exact sum of the sink's entries [(cells, incoming gradient)] over calls and ranks,
    all-reduced with ``allreduce(int64 tensor)`` (drt_amd.dist.allreduce_sum_).  ``scale``: the ONE incoming gradient every entry saw (the
    seed of full_batch_step's backward pass; a sum node hands it to every term unchanged).  A rank without entries takes part with zeros."""
    from .optix_mesh import _stream, _on
    lib, n = _lib.lib(), like.numel()
    total = entries[0][0] if entries else torch.zeros(n * _CELL_WORDS, dtype=torch.int64, device=like.device)
    with _on(like.device):
        if len(entries) > 1:
            total = total.clone()
            for cells, _ in entries[1:]:
                _lib.check(lib.drt_fx_add(total.data_ptr(), cells.data_ptr(), n, _stream()))
        limbs = torch.empty(n * 4, dtype=torch.int64, device=like.device)
        _lib.check(lib.drt_fx_to_limbs(total.data_ptr(), n, limbs.data_ptr(), _stream()))
        allreduce(limbs)
        summed = torch.empty_like(total)
        _lib.check(lib.drt_fx_from_limbs(limbs.data_ptr(), n, summed.data_ptr(), _stream()))
    return value(summed, like) * scale

