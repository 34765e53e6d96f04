"""View-parallel execution: one process per GPU, views sharded, one all-reduce per step.

The reference is single-GPU (no torch.distributed).  The views
of one optimisation step are independent given the (replicated) mesh, so they shard
with no data-path exchange: view k belongs to rank ``k % world``; every rank holds
the full mesh, rebuilds its own LBVH, accumulates a private float64 ``grad[V,3]``
and ONE all-reduce(sum) of that buffer (0.6 MB at 25 k vertices) precedes the
reference's ``limit_hook`` + SGD step (optim.py:155-171), which every rank then
applies identically, so vertices stay bit-identical without a broadcast
(SURVEY.md section 8e).  Backend ``nccl`` is RCCL over xGMI on ROCm; ``gloo`` is used
by the CPU tests.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


# DRT_DIST_FORCE=1: create the process group and issue the collectives even with ONE rank -- the RCCL path of a step
# (communicator, its streams, the all-reduce launch) exercised on a one-GPU box
_FORCE = os.environ.get("DRT_DIST_FORCE", "") not in ("", "0")


def _active():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE)


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None):
    """Initialise the default process group from the torchrun environment (no-op for world 1)."""
    rank, local_rank, world = env_world()
    if (world > 1 or _FORCE) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("DRT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_views(n_views, rank, world):
    """Indices of the views rank ``rank`` owns (round-robin, so a turntable is spread evenly)."""
    return list(range(rank, n_views, world))


def allreduce_sum_(t):
    """In-place sum over ranks; identity for a single process."""
    if _active():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def allreduce_max_float(x, device):
    t = torch.tensor([x], dtype=torch.float64, device=device)
    if _active():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if _active():
        dist.barrier()
