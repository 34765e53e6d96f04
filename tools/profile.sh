#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + separate PMC passes of bench.py.
# (PMC passes: one warm-up step -- it ESTABLISHES the grid verdict -- and two steady-state steps; per-kernel means.)
# Big raw outputs stay in /tmp; only summaries land in gpurun_out/<tag>/.
# usage: tools/profile.sh <tag> [bench args...]
export DRT_BENCH_REPEATS=1      # (one timed region per profiled run, whatever the caller exported)
set -u
TAG=${1:-prof}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
export TMPDIR=/tmp
W=/tmp/prof_$TAG; rm -rf "$W"; mkdir -p "$W"
cd "$R"
rocprofv3 --kernel-trace --stats --output-format csv -d "$W/kt" -o kt -- python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-extras "$@" > "$O/bench_under_rocprof.log" 2>&1
find "$W/kt" -name '*kernel_stats.csv' -exec cp {} "$O/kernel_stats.csv" \;
python - "$W/kt" "$O" <<'PY'
import csv, glob, sys, collections
src, out = sys.argv[1], sys.argv[2]
files = glob.glob(src + '/**/*kernel_trace.csv', recursive=True)
agg = collections.defaultdict(list)
meta = {}
for f in files:
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        agg[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        meta[k] = (r.get('VGPR_Count'), r.get('SGPR_Count'), r.get('LDS_Block_Size'), r.get('Workgroup_Size'), r.get('Grid_Size'))
tot = sum(sum(v) for v in agg.values()) or 1
with open(out + '/kernel_summary.txt', 'w') as o:
    o.write(f"{'kernel':60s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'max_us':>10s} {'pct':>6s}  vgpr sgpr lds wg grid\n")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        o.write(f"{k[:60]:60s} {len(v):6d} {sum(v)/1e6:10.3f} {sum(v)/len(v)/1e3:10.2f} {max(v)/1e3:10.2f} {100*sum(v)/tot:6.2f}  {meta[k]}\n")
    # the timed region of bench.py is the tail of the trace: per-kernel totals over the last `steps` steps
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0]))
    rows.sort()
    # a step starts with k_cast_verts (update_verticex); keep the last 3 steps
    starts = [i for i, r in enumerate(rows) if r[2].startswith('k_cast_verts')]
    if len(starts) >= 3:
        sel = rows[starts[-3]:]
        span = (sel[-1][1] - sel[0][0]) / 1e6
        per = collections.defaultdict(lambda: [0, 0])
        for a, b, k in sel:
            per[k][0] += 1; per[k][1] += b - a
        busy = sum(v[1] for v in per.values()) / 1e6
        o.write(f"\n== last 3 steps: wall span {span:.3f} ms, sum of kernel time {busy:.3f} ms ({span/3:.3f} ms/step)\n")
        for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
            o.write(f"{k[:60]:60s} {c:6d} {t/1e6:10.3f} ms  {t/c/1e3:10.2f} us/call  {100*t/1e6/span:6.2f}% of span\n")
PY
run_pmc () {  # name counters...
  local name=$1; shift
  # Counter passes run with the internal streams SERIALISED (one pipeline, fills in front of the projection pass, no ahead-of-time fills):
  # TCC / FETCH_SIZE / WRITE_SIZE are chip-wide, and with two pipelines + fills side by side a 2.4 MB kernel was charged 422 MB of its
  # neighbours' traffic (round 3).  Counters are per-kernel properties; launch TIMES come from the default run above.
  # (DRT_PREFILL_NEXT=0 also keeps the number of output fills equal to the number of patch lists.)
  DRT_STREAMS=1 DRT_FILL_OVERLAP=0 DRT_PREFILL_NEXT=0 DRT_ASYNC_BUILD=0 rocprofv3 --pmc "$@" --output-format csv -d "$W/$name" -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --random-targets "${BARGS[@]}" > "$O/$name.log" 2>&1
  local f=$(find "$W/$name" -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python tools/pmc_summary.py "$f" > "$O/$name.txt"; else echo "no counter file" > "$O/$name.txt"; fi
}
BARGS=("$@")
if [ -n "${PROFILE_SKIP_PMC:-}" ]; then du -sh "$O"; grep -A30 "last 3 steps" "$O/kernel_summary.txt"; exit 0; fi
run_pmc pmc_sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run_pmc pmc_sq2 SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR
run_pmc pmc_tcc TCC_HIT_sum TCC_MISS_sum
run_pmc pmc_tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
run_pmc pmc_fetch FETCH_SIZE
run_pmc pmc_write WRITE_SIZE
run_pmc pmc_grbm GRBM_GUI_ACTIVE
python - "$O" <<'PY'
# HBM traffic per launch of every pipeline kernel from the FETCH_SIZE / WRITE_SIZE passes (KB units).
# gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 128-byte requests as 64 bytes -> x2,
# calibrated here on k_cull (reads exactly 48 B/ray): see DESIGN.md section 6.  WRITE_SIZE is used as is.
import json, re, sys
out = sys.argv[1]
def parse(path, counter):
    res, cur = {}, None
    for line in open(path):
        m = re.match(r"(?:void )?((?:k_|__amd_rocclr_fill)\w*(?:<[^>]*>)?)\s+launches=(\d+)", line)
        if m:
            cur = (m.group(1), int(m.group(2)))
            continue
        if line and not line.startswith(" "):
            cur = None          # another kernel's header (a torch kernel: not kept) -- its counters must not land on the previous name
            continue            # (round 3: k_tri_flat, 5.8 MB per launch, was listed with 422 MB this way)
        m = re.match(r"\s+%s\s+total=(\S+)" % counter, line)
        if m and cur:
            res[cur[0]] = (float(m.group(1)), cur[1])
    return res
try:
    f, w = parse(out + "/pmc_fetch.txt", "FETCH_SIZE"), parse(out + "/pmc_write.txt", "WRITE_SIZE")
    t = {}
    for k in f:
        fetch = 2.0 * f[k][0] * 1024 / f[k][1]
        write = w.get(k, (0.0, 1))[0] * 1024 / max(1, w.get(k, (0.0, 1))[1])
        t[k] = {"fetch_bytes_per_launch": round(fetch), "write_bytes_per_launch": round(write), "hbm_bytes_per_launch": round(fetch + write), "launches": f[k][1]}
    json.dump(t, open(out + "/traffic_raw.json", "w"), indent=1)
except Exception as e:
    print("traffic summary failed:", e)
PY
du -sh "$O"; grep -A30 "last 3 steps" "$O/kernel_summary.txt"
