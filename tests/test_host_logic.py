"""Host-side logic that needs no GPU: the lazily materialised silhouette edge set, the provenance stamp of the PMC counters that
bench.py prices its roofline with, the C-ABI surface of the device remesher."""
import importlib.util
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT


def test_silhouette_edges_behave_like_the_tensor_the_reference_returns():
    """Scene.silhouette_edge returns Edges[flags] lazily (reference DiffRender.py:445-457 returns the tensor itself): whoever looks at it --
    indexing, len, shape, torch functions, .cpu() -- sees exactly that tensor; until then nothing has been compacted."""
    from drt_amd.diffrender import SilhouetteEdges
    edges = torch.arange(40).reshape(20, 2)
    flags = (torch.arange(20) % 3 == 0).to(torch.uint8)
    ref = edges[flags.bool()]
    s = SilhouetteEdges(edges, flags)
    assert s._t is None                                  # nothing materialised by construction
    assert s.shape == ref.shape and len(s) == len(ref) and s.dtype == torch.long
    assert s._t is not None
    assert torch.equal(s[2], ref[2]) and torch.equal(s[:, 1], ref[:, 1])
    assert torch.equal(torch.sort(s, dim=0).values, torch.sort(ref, dim=0).values)       # torch functions unwrap it
    assert torch.equal(torch.cat([s, s]), torch.cat([ref, ref]))
    assert np.array_equal(s.cpu().numpy(), ref.numpy()) and [tuple(r.tolist()) for r in s] == [tuple(r.tolist()) for r in ref]
    assert "SilhouetteEdges" in repr(s)


def _load_bench():
    spec = importlib.util.spec_from_file_location("bench_module_for_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_pmc_counters_are_only_used_with_the_kernel_sources_they_were_collected_on(tmp_path, monkeypatch):
    """profiles/pmc.json carries a sha256 of drt_amd/csrc + include (tools/make_pmc_json.py); bench.py recomputes it and refuses to mix
    counters of other kernels with this run's launch times (`pmc_stale`)."""
    from drt_amd import build
    h = build.source_hash()
    assert re.fullmatch(r"[0-9a-f]{64}", h) and h == build.source_hash()
    bench = _load_bench()
    fake_root = tmp_path
    (fake_root / "profiles").mkdir()
    rec = {"workload": "horse res 1024 views 72 streams default", "dropin": {"k_trace<false, 0>": {"SQ_INSTS_VALU": 1.5e8, "launches": 4}},
           "source_sha256": h, "git_head": "abc"}
    (fake_root / "profiles" / "pmc.json").write_text(json.dumps(rec))
    monkeypatch.setattr(bench, "ROOT", str(fake_root))
    pmc, prov = bench._pmc("dropin", rec["workload"])
    assert pmc and prov["stale"] is False and prov["source_sha256"] == h
    pmc2, _ = bench._pmc("dropin", "another workload")
    assert pmc2 == {}                                     # counters of another workload are not used at all
    rec["source_sha256"] = "0" * 64
    (fake_root / "profiles" / "pmc.json").write_text(json.dumps(rec))
    pmc3, prov3 = bench._pmc("dropin", rec["workload"])
    assert prov3["stale"] is True and prov3["sources_now_sha256"] == h
    (fake_root / "profiles" / "pmc.json").unlink()
    assert bench._pmc("dropin", rec["workload"])[1]["stale"] is True


def test_make_pmc_json_stamps_the_sources(tmp_path):
    src = tmp_path / "run"
    src.mkdir()
    (src / "pmc_sq1.txt").write_text("void k_trace<false, 0>  launches=4\n   SQ_INSTS_VALU total=6e8 per_launch=1.5e8\n")
    dst = tmp_path / "pmc.json"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_pmc_json.py"), str(src), str(dst), "dropin"], stdout=subprocess.DEVNULL)
    rec = json.loads(dst.read_text())
    from drt_amd import build
    assert rec["source_sha256"] == build.source_hash() and rec["dropin"]["k_trace<false, 0>"]["SQ_INSTS_VALU"] == 1.5e8


def test_fused_terms_and_weights_host_arithmetic():
    """loss_weights mirrors reference optim.py:127-129; interp_R / interp_L the pass schedule (optim.py:145-153)."""
    from drt_amd import optim as O
    hp = dict(O.HyperParams)
    w = O.loss_weights(hp, 960, 2.5)
    assert w == (40 * 217.5 / 960 / 960, 2e-3 * 217.5 / 960, 0.08 * 2.5 / 10)
    assert O.interp_R(10, 1, 0, 20) == pytest.approx(10) and O.interp_R(10, 1, 19, 20) == pytest.approx(1)
    assert O.interp_L(0.0, 1.9, 10, 20) == pytest.approx(1.0)


def test_output_pool_hands_buffers_out_only_when_nobody_else_holds_them():
    """diffrender._OutputPool (RECYCLE_OUTPUTS) on CPU tensors: an entry is taken only when the storages' use counts are back to the pool's
    own, the version counters have not moved, and size / device / stream match; at most two entries are kept."""
    import torch
    from drt_amd import diffrender as R
    if not hasattr(torch._C, "_storage_Use_Count"):
        import pytest
        pytest.skip("this torch has no storage use count")

    def fresh(n):
        bases = (torch.zeros(n, 3, dtype=torch.float64), torch.zeros(n, 3, dtype=torch.float64), torch.zeros(n, 3, dtype=torch.uint8))
        return bases, tuple(R._use_count(t) for t in bases)

    pool = R._OutputPool()
    bases, counts = fresh(8)
    lst, cnt = torch.arange(3, dtype=torch.int32), torch.tensor([3])
    pool.put(8, torch.device("cpu"), 7, bases, counts, lst, cnt)
    out = [b.detach() for b in bases]                      # what the caller holds: aliases with the same version counter
    assert out[0]._version == 0 and out[0].data_ptr() == bases[0].data_ptr()
    assert pool.take(8, torch.device("cpu"), 7) is None    # the caller still holds them
    view = out[1][2:4]
    del out
    assert pool.take(8, torch.device("cpu"), 7) is None    # a view of one of them survives
    del view
    assert pool.take(9, torch.device("cpu"), 7) is None and pool.take(8, torch.device("cpu"), 8) is None     # another size / stream
    ent = pool.take(8, torch.device("cpu"), 7)
    assert ent is not None and ent[3][0] is bases[0] and ent[5] is lst and len(pool.entries) == 0
    # a caller that wrote into its outputs before dropping them: the shared version counter shows it
    pool.put(8, torch.device("cpu"), 7, bases, counts, lst, cnt)
    alias = bases[1].detach()
    alias.add_(1.0)
    del alias
    assert bases[1]._version == 1 and pool.take(8, torch.device("cpu"), 7) is None
    # eviction: two entries at most
    for k in range(3):
        b, c = fresh(4 + k)
        pool.put(4 + k, torch.device("cpu"), 7, b, c, lst, cnt)
    assert len(pool.entries) == R._OutputPool.MAX == 2 and [e[0] for e in pool.entries] == [5, 6]


def test_targets_are_known_complete_only_when_seen_before_the_render_call():
    """diffrender._targets_seen_before (SPLIT_LOSS): the early start of the loss pass is taken only for target tensors that went through
    ray_loss as the same objects, storage and version BEFORE the render call whose loss is taken was enqueued (the internal stream is
    ordered behind that call's fork only)."""
    import torch
    from drt_amd import diffrender as R
    sp, valid = torch.zeros(5, 3, dtype=torch.float64), torch.ones(5, dtype=torch.bool)

    def render():                                           # what _RenderTransparent.forward does to the clock
        R._render_seq[0] += 1
        return R._render_seq[0]

    a = render()
    assert not R._targets_seen_before(sp, valid, a)        # first use: registered now
    b = render()
    assert R._targets_seen_before(sp, valid, b)            # the usual loop: seen in the previous iteration's loss
    sp.add_(1.0)                                           # written since: not known to be complete any more ...
    c = render()
    assert not R._targets_seen_before(sp, valid, c)
    d = render()
    assert R._targets_seen_before(sp, valid, d)            # ... until it has been seen again in that state, before the call
    other = torch.zeros(5, 3, dtype=torch.float64)
    assert not R._targets_seen_before(other, valid, render())
    # render A, render B, targets built on the stream, ray_loss(A, tgt), ray_loss(B, tgt): registered by A's loss -- but behind B's fork
    tgt, v2 = torch.zeros(5, 3, dtype=torch.float64), torch.ones(5, dtype=torch.bool)
    seq_a, seq_b = render(), render()
    assert not R._targets_seen_before(tgt, v2, seq_a)
    assert not R._targets_seen_before(tgt, v2, seq_b)
    assert R._targets_seen_before(tgt, v2, render())
