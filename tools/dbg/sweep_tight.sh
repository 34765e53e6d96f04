export DRT_BENCH_REPEATS=3 DRT_BENCH_NOPROF=1
for e in "DRT_STREAMS=2 DRT_SUB_PER_STREAM=1" "DRT_STREAMS=2 DRT_SUB_PER_STREAM=2" "DRT_STREAMS=2 DRT_SUB_PER_STREAM=3" "DRT_STREAMS=2 DRT_SUB_PER_STREAM=4" "DRT_STREAMS=3 DRT_SUB_PER_STREAM=1" "DRT_STREAMS=3 DRT_SUB_PER_STREAM=2" "DRT_STREAMS=4 DRT_SUB_PER_STREAM=1"; do
  for a in "--distance-factor 1.1" ""; do
    echo "$e :: bench.py $a :: $(env $e python bench.py $a --no-cpu-baseline --no-extras --steps 10 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['repeats']['ms_per_step'])")"
  done
done
