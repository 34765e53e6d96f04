#!/bin/bash
# Kernel timeline (start offset, duration) of the last kernels of a python script under rocprofv3 --kernel-trace.
# usage (via gpurun): bash tools/timeline.sh <script.py> [n_last]
export DRT_BENCH_REPEATS=1      # (one timed region per profiled run, whatever the caller exported)
export TMPDIR=/tmp
rm -rf /tmp/rp_tl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_tl -o t -- python $1 > /tmp/rp_tl.log 2>&1
f=$(find /tmp/rp_tl -name "*kernel_trace.csv" | head -1)
[ -z "$f" ] && { tail -5 /tmp/rp_tl.log; exit 1; }
python - "$f" "${2:-40}" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = rows[-int(sys.argv[2]):]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-44s start %9.1f us  dur %8.1f us  grid %s wg %s stream %s" % (r["Kernel_Name"].split("(")[0][:44], (st - t0) / 1e3, (en - st) / 1e3, r.get("Grid_Size"), r.get("Workgroup_Size"), str(r.get("Stream_Id")) + " queue " + str(r.get("Queue_Id"))))
PY
