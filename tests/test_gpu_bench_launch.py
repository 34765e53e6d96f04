"""The driver's own launch line for N > 1 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
--master-port P bench.py --gpus N --steps K --warmup W`), run as a subprocess: one rank per GPU over RCCL where the box has >= 2 GPUs,
two ranks sharing the one GPU over gloo otherwise (DRT_DIST_BACKEND=gloo: a functional check of the same code path).  The JSON line of
rank 0 must say n_gpus = 2 and carry the loss of ALL views, equal to the one-process run's after the same number of steps (SURVEY.md
section 8e: the ranks' parameters stay identical, the all-reduce only changes the order of a float64 sum)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu

ARGS = ["--steps", "2", "--warmup", "1", "--res", "256", "--views", "8", "--no-cpu-baseline", "--no-extras", "--repeats", "3"]       # (a fixed number of repeats: the two runs must take the same number of steps)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def _run(cmd, env):
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    return _json_line(p.stdout)


def test_the_drivers_launch_line_with_two_ranks_equals_one_process():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    one = _run([sys.executable, "bench.py", "--gpus", "1"] + ARGS, env)
    assert one["n_gpus"] == 1 and one["config"]["views_per_gpu"] == 8
    env2 = dict(env)
    rccl = torch.cuda.device_count() >= 2
    if not rccl:
        env2["DRT_DIST_BACKEND"] = "gloo"          # two ranks on the one GPU of the box: the all-reduce goes through the host
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port()), "bench.py", "--gpus", "2"] + ARGS, env2)
    assert two["n_gpus"] == 2 and two["steps"] == 2 and two["warmup"] == 1
    assert two["metric"] == one["metric"] and two["unit"] == "M camera-rays/s" and two["scaling"] == "strong"
    assert two["config"]["views_per_gpu"] == 4 and two["config"]["workload"] == one["config"]["workload"]
    # a share of 4 x 256^2 rays is below the whole-step-graph threshold: captured over RCCL on every rank or on none (never over gloo)
    assert two["config"]["hip_graph"] in ((True, False) if rccl else (False,))
    a, b = one["config"]["final_loss"], two["config"]["final_loss"]
    assert a > 0 and abs(a - b) <= 1e-12 * abs(a), (a, b)
    assert two["value"] > 0 and two["ms_per_step"] > 0
    # the line explains its own exchange: process group as torch.distributed sees it, the all-reduce alone on every rank
    mg = two["multi_gpu"]
    assert mg["world_size_seen_by_torch_distributed"] == 2 and mg["backend"] == ("nccl" if rccl else "gloo")
    assert mg["views_per_rank"] == [4, 4] and len(mg["allreduce_ms_per_rank"]) == 2 and all(t > 0 for t in mg["allreduce_ms_per_rank"])
    assert "multi_gpu" not in one
    rp = two["repeats"]
    assert rp["n"] >= 1 and rp["min"] <= rp["median"] <= rp["max"] and two["ms_per_step"] == rp["median"]


def test_bench_as_a_whole_step_graph_recycles_its_outputs():
    """`bench.py --graph 1` (what an N > 1 run with small shares does): the step is captured, the capture takes a pooled output set
    (config.outputs_recycled)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["DRT_RECYCLE_MIN_RAYS"] = "0"
    args = ["--steps", "3", "--warmup", "3", "--res", "256", "--views", "8", "--no-cpu-baseline", "--no-extras", "--repeats", "2", "--bind", "0"]      # (the drop-in signature: a pooled set becomes the graph's; the default, a RayBinding, brings its own)
    eager = _run([sys.executable, "bench.py", "--gpus", "1", "--graph", "0"] + args, env)
    graph = _run([sys.executable, "bench.py", "--gpus", "1", "--graph", "1"] + args, env)
    assert eager["config"]["hip_graph"] is False and eager["config"]["outputs_recycled"] is True
    assert graph["config"]["hip_graph"] is True and graph["config"]["outputs_recycled"] is True
    # (the two runs take a different number of untimed steps -- the capture needs eager warm-up steps and one replay of its own -- so their
    # final losses are those of neighbouring iterations; that replays equal eager steps is tests/test_gpu_dist.py's business)
    a, b = eager["config"]["final_loss"], graph["config"]["final_loss"]
    assert 0 < b < a, (a, b)          # (more steps taken: further down the same descent)
    assert graph["repeats"]["n"] == 2 and len(graph["repeats"]["ms_per_step"]) == 2


def test_eight_ranks_of_nine_views_the_drivers_8_gpu_line_on_one_gpu():
    """BASELINE.json config 4 as the driver launches it -- `--nproc-per-node 8 ... bench.py --gpus 8`, 72 views = 9 per rank -- with the eight
    ranks sharing this box's one GPU over gloo (a functional check of the N = 8 path: eight scenes resident at once, the all-reduce of
    grad[V,3] across eight ranks, rank 0's JSON line; no timing is read off it).  Reduced image size, the full mesh."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env["DRT_DIST_BACKEND"] = "gloo"
    args = ["--steps", "2", "--warmup", "1", "--res", "256", "--views", "72", "--no-cpu-baseline", "--no-extras", "--repeats", "2"]
    one = _run([sys.executable, "bench.py", "--gpus", "1"] + args, env)
    eight = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                  "--master-port", str(_free_port()), "bench.py", "--gpus", "8"] + args, env)
    assert eight["n_gpus"] == 8 and eight["config"]["views_per_gpu"] == 9 and eight["scaling"] == "strong"
    mg = eight["multi_gpu"]
    assert mg["world_size_seen_by_torch_distributed"] == 8 and mg["views_per_rank"] == [9] * 8 and len(mg["allreduce_ms_per_rank"]) == 8
    assert mg["allreduce_bytes"] == 25126 * 3 * 8
    a, b = one["config"]["final_loss"], eight["config"]["final_loss"]
    assert a > 0 and abs(a - b) <= 1e-12 * abs(a), (a, b)
    assert eight["value"] > 0 and eight["config"]["workload"] == one["config"]["workload"]
