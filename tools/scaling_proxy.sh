#!/bin/bash
# Per-rank shares of the 72 views on ONE GPU, eager and as the whole-step hipGraph (what bench.py does for N > 1): profiles/rNN_scaling_proxy.txt
export DRT_BENCH_REPEATS=${DRT_BENCH_REPEATS:-3}
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

# BASELINE.json config 5: monkey_vh.ply (184 090 triangles), 144 views on 8 GPUs = 18 views per rank; and the whole job on one GPU for the ratio
for v in 144 18; do
  for g in 0 1; do
    python bench.py --mesh monkey --views $v --graph $g --no-cpu-baseline --no-extras --steps 20 --warmup 5 --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 5 (monkey 184k): views $v', 'hipGraph' if d['config']['hip_graph'] else 'eager', d['ms_per_step'], 'ms/step', d['repeats']['ms_per_step'])"
  done
done
# weak scaling (SURVEY 8e's partition with the per-GPU work held fixed): every rank keeps 72 views of the horse -- the step of ONE rank is the headline
# step plus the all-reduce of 0.6 MB, so the projected weak-scaling efficiency is t(72 views) / (t(72 views) + t(all-reduce)); the all-reduce alone:
DRT_DIST_FORCE=1 python bench.py --views 72 --graph 0 --no-cpu-baseline --no-extras --steps 20 --warmup 5 --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d.get('multi_gpu') or {}
print('weak-scaling row: 72 views per rank', d['ms_per_step'], 'ms/step; all-reduce of grad[V,3] alone (RCCL, one rank)', m.get('allreduce_ms_per_rank'), 'ms ->', 'projected efficiency', round(d['ms_per_step'] / (d['ms_per_step'] + max(m.get('allreduce_ms_per_rank') or [0])), 4))"
