"""Small helpers shared by drt_amd.diffrender (refraction path, Scene) and drt_amd.silhouette (silhouette / smoothness branches)."""
from __future__ import annotations

import collections
import warnings

import torch

# What the caches below did since import (or since cache_report(reset=True)): cache_report().  They are transparent by design -- a call that
# cannot use one simply takes the slower path -- so this is where a caller sees WHICH path its calls took, and why.
_stats = collections.Counter()
_warned = set()


def _warn_once(key, msg):
    if key not in _warned:
        _warned.add(key)
        warnings.warn("drt_amd.diffrender: " + msg, RuntimeWarning, stacklevel=3)



def _f64c(t, name):
    if t.dtype != torch.float64:
        raise RuntimeError(f"{name} must be float64, got {t.dtype}")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor")
    return t.contiguous()



def _flag_bytes(t, name, n):
    """bool / uint8 [n] flags as a contiguous uint8 view (the kernels read them with 4-byte packed loads)."""
    if t.dtype not in (torch.bool, torch.uint8):
        raise RuntimeError(f"{name} must be bool or uint8, got {t.dtype}")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor")
    if t.numel() != n:
        raise RuntimeError(f"{name} must have {n} elements, got {t.numel()}")
    return t.contiguous().view(torch.uint8)
