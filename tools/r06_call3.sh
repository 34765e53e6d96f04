#!/bin/bash
O=gpurun_out/r06c; mkdir -p $O
python -m pytest tests/test_gpu_deterministic.py tests/test_gpu_bench_launch.py tests/test_gpu_dist.py -q --maxfail=8 -rf > $O/gputest.log 2>&1; tail -12 $O/gputest.log
export DRT_BENCH_REPEATS=5
for e in 0 1; do
  DRT_DETERMINISTIC=$e python bench.py --no-cpu-baseline --no-extras --steps 20 2> $O/err_det$e.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DRT_DETERMINISTIC=$e', d['ms_per_step'], d['repeats']['ms_per_step'], {k:v['ms_per_step'] for k,v in d['roofline']['stages'].items()})"
done 2>&1 | tee $O/det_cost.txt
DRT_DETERMINISTIC=1 python bench.py --distance-factor 1.1 --no-cpu-baseline --no-extras --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tight DRT_DETERMINISTIC=1', d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['roofline']['stages'].items()})" | tee -a $O/det_cost.txt
