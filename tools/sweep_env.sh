#!/bin/bash
# One environment variable over a list of values, bench.py (no extras, 3 repeats) per value, twice round-robin: tools/sweep_env.sh VAR "v1 v2 v3" [bench args]
export DRT_BENCH_REPEATS=${DRT_BENCH_REPEATS:-3}      # (bench.py without --repeats runs a >= 3 s sustained measurement: not what this script is after)
var=$1; vals=$2; shift 2
for r in 1 2; do
  for v in $vals; do
    out=$(env $var=$v DRT_BENCH_NOPROF=1 python bench.py --no-cpu-baseline --no-extras --repeats 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['repeats']['ms_per_step'])")
    echo "$var=$v round $r: $out"
  done
done
