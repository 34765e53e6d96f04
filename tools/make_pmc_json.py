#!/usr/bin/env python3
"""profiles/pmc.json from the PMC summaries of a tools/profile.sh run: per kernel, per launch -- VALU wave-instructions,
SALU, vector-memory and LDS instructions, busy / wait quad-cycles, L2 hits and misses, HBM bytes (FETCH_SIZE x 2 on gfx950 +
WRITE_SIZE, KiB units).  bench.py reads it for the issue-rate bound of the traversal kernel and the `traffic` of the
HBM stage.  usage: tools/make_pmc_json.py gpurun_out/<tag> profiles/pmc.json [mode]"""
import json
import os
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
mode = sys.argv[3] if len(sys.argv) > 3 else "dropin"
KEEP = re.compile(r"^(?:void )?((?:k_|__amd_rocclr_fill)\w*(?:<[^>]*>)?)\s+launches=(\d+)")
out = {}
for name in ("pmc_sq1", "pmc_sq2", "pmc_tcc", "pmc_tcp", "pmc_fetch", "pmc_write", "pmc_grbm"):
    path = os.path.join(src, name + ".txt")
    if not os.path.exists(path):
        continue
    cur = None
    for line in open(path):
        m = KEEP.match(line)
        if m:
            cur = out.setdefault(m.group(1), {"launches": int(m.group(2))})
            continue
        if line and not line.startswith(" "):
            cur = None
            continue
        m = re.match(r"\s+(\w+)\s+total=(\S+)\s+per_launch=(\S+)", line)
        if m and cur is not None:
            cur[m.group(1)] = float(m.group(3))
for k, v in out.items():
    if "FETCH_SIZE" in v or "WRITE_SIZE" in v:
        v["hbm_bytes_per_launch"] = round(2.0 * 1024 * v.get("FETCH_SIZE", 0.0) + 1024 * v.get("WRITE_SIZE", 0.0))
try:
    res = json.load(open(dst))
except Exception:
    res = {}
res[mode] = out
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drt_amd import build as _build  # noqa: E402
import subprocess  # noqa: E402
res["source_sha256"] = _build.source_hash()          # bench.py refuses to price live launch times against counters of other kernels
try:
    res["git_head"] = subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip() or None
except Exception:
    res["git_head"] = None
wl = os.environ.get("PMC_WORKLOAD", "horse res 1024 views 72 streams default")
res.setdefault("workloads", {})[mode] = wl          # (per mode: "tight" = the same step with the cameras at 1.1 extents)
if mode in ("dropin", "fused") or "workload" not in res:
    res["workload"] = wl
res["note"] = ("per-launch means from separate rocprofv3 --pmc passes of `bench.py --steps 2 --warmup 1 --no-extras --random-targets` "
               "(tools/profile.sh); SQ_* cycle counters are quad-cycles; FETCH_SIZE x 1024 x 2 (gfx950 tallies 128-byte requests as 64, "
               "MI355X_MICROARCH.md) + WRITE_SIZE x 1024")
json.dump(res, open(dst, "w"), indent=1, sort_keys=True)
print(dst, len(out), "kernels")
