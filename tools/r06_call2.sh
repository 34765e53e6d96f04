#!/bin/bash
# round 6, GPU call 2: GPU tests of the new paths first, then timings (iteration bench, binding vs heuristics, deterministic-mode cost)
O=gpurun_out/r06b; mkdir -p $O
rocm-smi --showclocks --showpower --showmaxpower 2>/dev/null | grep -E "sclk|mclk|Power|power" | head -8 > $O/smi_before.txt
python -m pytest tests/test_gpu_deterministic.py tests/test_gpu_lazy.py tests/test_gpu_recycle.py tests/test_gpu_trajectory.py tests/test_gpu_raster.py -q --maxfail=8 -rf > $O/gputest_new.log 2>&1
tail -25 $O/gputest_new.log
python tools/iter_bench.py > $O/iter_bench.txt 2>&1; grep -v amdgpu $O/iter_bench.txt
export DRT_BENCH_REPEATS=7
for v in "--bind 1" "--bind 0"; do
  python bench.py --no-cpu-baseline --no-extras --steps 20 $v 2> $O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['repeats']['ms_per_step'], d['config']['allocator_in_timed_region']['host_enqueue_ms_per_step'])"
done 2>&1 | tee $O/bind_ab.txt
DRT_DETERMINISTIC=1 python bench.py --no-cpu-baseline --no-extras --steps 20 2> $O/err_det.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DRT_DETERMINISTIC=1', d['ms_per_step'], d['repeats']['ms_per_step'], {k:v['ms_per_step'] for k,v in d['roofline']['stages'].items()})" 2>&1 | tee $O/det_cost.txt
tail -3 $O/err_det.txt
unset DRT_BENCH_REPEATS
python -m pytest tests -m gpu -q --maxfail=5 -rf --deselect tests/test_gpu_deterministic.py --deselect tests/test_gpu_lazy.py --deselect tests/test_gpu_recycle.py --deselect tests/test_gpu_trajectory.py --deselect tests/test_gpu_raster.py > $O/gputest_rest.log 2>&1
tail -15 $O/gputest_rest.log
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | head -4 > $O/smi_after.txt; cat $O/smi_before.txt $O/smi_after.txt
