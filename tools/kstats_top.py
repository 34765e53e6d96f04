#!/usr/bin/env python3
"""Top kernels of a rocprofv3 *kernel_stats.csv by total time, and the total: tools/kstats_top.py <file> [n]"""
import csv
import re
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} launches")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:n]:
    name = re.sub(r"\(anonymous namespace\)::|drt::|^void ", "", r["Name"]).split("(")[0][:72]
    print(f"{name:72s} calls {int(r['Calls']):6d}  total {float(r['TotalDurationNs']) / 1e6:8.2f} ms  avg {float(r['AverageNs']) / 1e3:8.1f} us")
