#!/bin/bash
# usage: tools/abstage.sh <stage> variant... : ms/step and the named stage (alone, in the step) for each drt_amd/_ab/<variant>.so, two rounds
export DRT_BENCH_REPEATS=${DRT_BENCH_REPEATS:-3}      # (bench.py without --repeats runs a >= 3 s sustained measurement: not what this script is after)
st=$1; shift
for r in 1 2; do for v in "$@"; do
  DRT_HIP_LIB=$PWD/drt_amd/_ab/$v.so python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v', d['ms_per_step'], '$st in step', r['stages']['$st']['ms_per_step'], 'alone', (r.get('stages_alone_avg_launch_ms') or {}).get('$st'))"
done; done
