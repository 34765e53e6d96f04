#!/bin/bash
# Gantt of ONE step of bench.py (the second-to-last one of the timed region) under rocprofv3 --kernel-trace: every dispatch with its
# start offset from the step's k_cast_verts, duration and hardware queue, in start order, plus the idle gaps of the chip.
# usage (via gpurun): bash tools/step_timeline.sh <out.txt> [bench args...]
export DRT_BENCH_REPEATS=1      # (one timed region per profiled run, whatever the caller exported)
export TMPDIR=/tmp
out=$1; shift
rm -rf /tmp/rp_st
DRT_BENCH_NOPROF=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_st -o t -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras "$@" > /tmp/rp_st.log 2>&1
f=$(find /tmp/rp_st -name "*kernel_trace.csv" | head -1)
[ -z "$f" ] && { tail -5 /tmp/rp_st.log; exit 1; }
python - "$f" > "$out" <<'PY'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id"), r.get("Grid_Size"), r.get("Workgroup_Size")) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_cast_verts")]
a, b = starts[-3], starts[-2]
sel = rows[a:b]
t0 = sel[0][0]
print(f"step: {(rows[b][0] - t0) / 1e3:.1f} us from k_cast_verts to the next one; {len(sel)} dispatches")
queues = sorted({r[3] for r in sel})
busy_until = 0
for s, e, k, q, g, w in sel:
    gap = (s - busy_until) / 1e3 if busy_until and s > busy_until else 0.0
    col = queues.index(q)
    print("%9.1f %8.1f  q%d %s%-46s grid %-9s wg %-4s%s" % ((s - t0) / 1e3, (e - s) / 1e3, col, "    " * col, k[:46], g, w, f"   <- chip idle {gap:.1f} us before" if gap > 1.0 else ""))
    busy_until = max(busy_until, e)
PY
tail -3 /tmp/rp_st.log | head -2
