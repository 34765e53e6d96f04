#!/bin/bash
# usage: tools/sweep.sh "ENV1=a ENV2=b" "ENV1=c" ...   -> one line per configuration: ms/step and M rays/s of bench.py (timed region only)
# extra bench arguments through BENCH_ARGS.  Run on the GPU box (gpurun -- 'tools/sweep.sh ...').
export DRT_BENCH_REPEATS=${DRT_BENCH_REPEATS:-3}      # (bench.py without --repeats runs a >= 3 s sustained measurement: not what this script is after)
for cfg in "$@"; do
  out=$(env $cfg python bench.py --no-cpu-baseline --no-extras --steps ${STEPS:-10} --warmup 3 $BENCH_ARGS 2>/dev/null | tail -1)
  echo "$cfg :: $(python -c "import json,sys; d=json.loads(sys.argv[1]); print(d['ms_per_step'], 'ms', d['value'], 'Mrays/s')" "$out" 2>/dev/null || echo FAILED)"
done
