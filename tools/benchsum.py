#!/usr/bin/env python3
"""Print the numbers of a bench.py JSON line that are looked at while tuning: step, per-stage (overlapped / isolated)."""
import json
import sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{d['value']:.0f} Mrays/s  {d['ms_per_step']} ms/step   fused: {d.get('fused_mode', {}).get('ms_per_step')}")
st = d["roofline"]["stages"]
iso = d["roofline"].get("stages_alone_avg_launch_ms") or {}
r = d["roofline"]
print("  dominant:", r.get("dominant_by_kernel_name"), {k: r.get(k) for k in ("bound", "achieved", "peak", "frac")}, "hbm:", {k: r["hbm"].get(k) for k in ("achieved", "frac", "traffic", "alg_bytes_per_launch")})
for k, v in st.items():
    extra = " ".join(f"{a}={v[a]}" for a in ("node_visits_per_ray", "lane_utilisation", "longest_wave_visits") if a in v)
    print(f"  {k:15s} {v['ms_per_step']:7.3f} ms/step  {v['launches']:3d} launches  {v['avg_launch_ms']:7.4f} ms/launch  alone {iso.get(k, float('nan')):7.4f}  items/launch {v['items_per_launch']:9d}  {v['alg_GBps']:7.1f} GB/s {extra}")
