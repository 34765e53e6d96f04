#!/bin/bash
# Per-launch durations of the kernels whose name matches a pattern, in launch order, under rocprofv3 --kernel-trace.
# usage (via gpurun): bash tools/kernel_durations.sh <name-regex> <n_last> <command ...>
export DRT_BENCH_REPEATS=1      # (one timed region per profiled run, whatever the caller exported)
export TMPDIR=/tmp
pat=$1; n=$2; shift 2
rm -rf /tmp/rp_kd
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_kd -o t -- "$@" > /tmp/rp_kd.log 2>&1
f=$(find /tmp/rp_kd -name "*kernel_trace.csv" | head -1)
[ -z "$f" ] && { tail -5 /tmp/rp_kd.log; exit 1; }
python - "$f" "$pat" "$n" <<'PY'
import csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if re.search(sys.argv[2], r["Kernel_Name"])]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = rows[-int(sys.argv[3]):]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(anonymous namespace\)::|drt::|^void ", "", r["Kernel_Name"])
    print("%-40s start %10.1f us  dur %8.1f us  grid %s" % (name.split("(")[0][:40], (st - t0) / 1e3, (en - st) / 1e3, r.get("Grid_Size")))
PY
