#!/bin/bash
# Per-rank shares of the 72 views on ONE GPU, eager and as the whole-step hipGraph (what bench.py does for N > 1): profiles/rNN_scaling_proxy.txt
for v in 72 36 18 9; do
  DRT_BENCH_NOPROF=1 python bench.py --views $v --graph 0 --no-cpu-baseline --no-extras --steps 40 --warmup 5 --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('views $v eager', d['ms_per_step'], 'ms/step', d['repeats']['ms_per_step'], 'recycled' if d['config']['outputs_recycled'] else 'filled')"
done
for v in 72 36 18 9; do
  python bench.py --views $v --graph 1 --no-cpu-baseline --no-extras --steps 40 --warmup 5 --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('views $v whole-step hipGraph', d['ms_per_step'], 'ms/step', d['repeats']['ms_per_step'], 'hip_graph', d['config']['hip_graph'], 'recycled' if d['config']['outputs_recycled'] else 'filled')"
done
DRT_DIST_FORCE=1 python bench.py --views 9 --graph 1 --no-cpu-baseline --no-extras --steps 40 --warmup 5 --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('views 9 whole-step hipGraph with the all-reduce issued through RCCL (one rank)', d['ms_per_step'], 'ms/step', d.get('multi_gpu'))"
