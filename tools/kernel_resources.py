#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage remarks (stdin) as one line per kernel."""
import re
import sys

cur = None
rows = []
for line in sys.stdin:
    m = re.search(r"remark:\s+Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z][\w \[\]/]*?): (\S+) \[-Rpass", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
for r in rows:
    print(f"{r['name'][:44]:44s} vgpr={r.get('VGPRs','?'):>4} sgpr={r.get('TotalSGPRs','?'):>4} "
          f"scratch={r.get('ScratchSize [bytes/lane]','?'):>4} occ={r.get('Occupancy [waves/SIMD]','?'):>2} "
          f"lds={r.get('LDS Size [bytes/block]','?'):>6} spill={r.get('VGPRs Spill','?')}")
