#!/bin/bash
# A/B on ONE box: bench.py with the baseline library (drt_amd/_ab/base.so, built from HEAD) and the working-tree one,
# alternating, 3 rounds.  usage (via gpurun): bash tools/ab.sh [bench args]
export DRT_BENCH_REPEATS=${DRT_BENCH_REPEATS:-3}      # (bench.py without --repeats runs a >= 3 s sustained measurement: not what this script is after)
for r in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export DRT_HIP_LIB=$PWD/drt_amd/_ab/base.so; else unset DRT_HIP_LIB; fi
    timeout 300 python bench.py --no-cpu-baseline --steps 5 --no-extras "$@" | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']['stages']
print('$v', d['ms_per_step'], {k:r[k]['ms_per_step'] for k in r})"
  done
done
