#!/bin/bash
O=gpurun_out/r06e; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rmprof -o rm -- python tools/ubench/remesh_probe.py 0.9 > $O/probe_under_rocprof.txt 2>&1

f=$(find /tmp/rmprof -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY' | tee $O/remesh_kernels.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot / 1e6, 'launches', sum(int(r['Calls']) for r in rows))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:32]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.3f} ms {int(r['Calls']):6d} calls {float(r['AverageNs'])/1e3:9.2f} us  {r['Name'][:110]}")
PY
