#!/bin/bash
# A/B of one environment switch on ONE box: bench.py with VAR=0 and VAR=1 alternating, 2 rounds, full extras (per-stage numbers).
# usage (via gpurun): bash tools/ab_env.sh DRT_HIT_SEED [bench args]
export DRT_BENCH_REPEATS=${DRT_BENCH_REPEATS:-3}      # (bench.py without --repeats runs a >= 3 s sustained measurement: not what this script is after)
var=$1; shift
mkdir -p gpurun_out/ab
for r in 1 2; do
  for v in 0 1; do
    env $var=$v timeout 600 python bench.py --no-cpu-baseline "$@" > gpurun_out/ab/${var}_${v}_$r.json 2> gpurun_out/ab/${var}_${v}_$r.err
    echo "== $var=$v round $r"; python tools/benchsum.py gpurun_out/ab/${var}_${v}_$r.json | grep -E "Mrays|trace|raster|cull"
    python - <<PY
import json
d=json.loads(open("gpurun_out/ab/${var}_${v}_$r.json").read().strip().splitlines()[-1])
e=d.get("establish_mode") or {}
print("   repeats:", d.get("repeats",{}).get("ms_per_step"), " tight:", (d.get("tight_framing") or {}).get("ms_per_step"), " establish:", e.get("ms_per_step"), (e.get("stages_ms_per_step") or {}).get("cull"), " fused:", (d.get("fused_mode") or {}).get("ms_per_step"))
PY
  done
done
