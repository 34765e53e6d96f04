#!/usr/bin/env python3
"""Aggregate a rocprofv3 counter_collection CSV per kernel: launches, and per-counter sum / mean per launch."""
import collections
import csv
import sys

path = sys.argv[1]
keep = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
with open(path) as f:
    for row in csv.DictReader(f):
        k = row.get("Kernel_Name", "?")
        if keep and keep not in k:
            continue
        k = k.split("(")[0][:60]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        disp[k].add(row.get("Dispatch_Id"))
for k in sorted(agg):
    n = max(1, len(disp[k]))
    print(f"{k}  launches={n}")
    for c, v in sorted(agg[k].items()):
        print(f"    {c:34s} total={v:.6g}  per_launch={v / n:.6g}")
