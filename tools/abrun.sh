#!/bin/bash
# usage: tools/abrun.sh <stage-regex> variant1 variant2 ...   (variants = names of drt_amd/_ab/<name>.so; "base" = the in-tree library)
export DRT_BENCH_REPEATS=${DRT_BENCH_REPEATS:-5}      # (bench.py without --repeats runs a >= 3 s sustained measurement: not what this script is after)
pat=$1; shift
for v in "$@" "$@"; do
  if [ "$v" = base ]; then unset DRT_HIP_LIB; else export DRT_HIP_LIB=$PWD/drt_amd/_ab/$v.so; fi
  python bench.py --no-cpu-baseline > gpurun_out/ab_$v.json 2>/dev/null
  echo "== $v"; python tools/benchsum.py gpurun_out/ab_$v.json | grep -E "Mrays|$pat"
done
