for r in 1 2; do for cfg in "DRT_HIT_SEED=0" "DRT_HIT_SEED=1 DRT_SEED_TILED=0" "DRT_HIT_SEED=1 DRT_SEED_TILED=1"; do
export DRT_BENCH_REPEATS=${DRT_BENCH_REPEATS:-3}      # (bench.py without --repeats runs a >= 3 s sustained measurement: not what this script is after)
  out=$(env $cfg DRT_BENCH_NOPROF=1 python bench.py --no-cpu-baseline --no-extras --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['repeats']['ms_per_step'])")
  t=$(env $cfg DRT_BENCH_NOPROF=1 python bench.py --no-cpu-baseline --no-extras --repeats 3 --steps 10 --distance-factor 1.1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
  echo "$cfg :: headline $out :: tight $t"
done; done
