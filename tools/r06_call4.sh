#!/bin/bash
O=gpurun_out/r06d; mkdir -p $O
python -m pytest tests/test_gpu_remesh.py -q --maxfail=5 -rf > $O/gputest_remesh.log 2>&1; tail -8 $O/gputest_remesh.log
python tools/ubench/remesh_probe.py 0.9 2>&1 | grep -v amdgpu | tee $O/remesh_probe.txt
for k in 2 4 6; do DRT_REMESH_SUB_ROUNDS=$k python tools/ubench/remesh_probe.py 0.9 2>&1 | grep -v amdgpu | tail -2 | sed "s/^/SUB_ROUNDS=$k /"; done | tee $O/remesh_sub.txt
REMESH=gpu python tools/recon_trend.py 2>&1 | grep -v amdgpu | tail -6 | tee $O/recon_trend_gpu.txt
