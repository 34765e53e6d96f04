#!/bin/bash
O=gpurun_out/r06f; mkdir -p $O
python -m pytest tests/test_gpu_remesh.py tests/test_gpu_topology.py -q --maxfail=5 -rf > $O/gputest_remesh.log 2>&1; tail -8 $O/gputest_remesh.log
python tools/ubench/remesh_probe.py 0.9 2>&1 | grep -v amdgpu | tee $O/remesh_probe.txt
DRT_REMESH_KEEP_VERDICTS=0 python tools/ubench/remesh_probe.py 0.9 2>&1 | grep -v amdgpu | tail -2 | sed "s/^/KEEP_VERDICTS=0 /" | tee -a $O/remesh_probe.txt
for k in 2 4; do DRT_REMESH_SUB_ROUNDS=$k python tools/ubench/remesh_probe.py 0.9 2>&1 | grep -v amdgpu | tail -2 | sed "s/^/SUB_ROUNDS=$k /"; done | tee $O/remesh_sub.txt
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rmprof -o rm -- python tools/ubench/remesh_probe.py 0.9 > $O/probe_under_rocprof.txt 2>&1
f=$(find /tmp/rmprof -name '*kernel_stats.csv' | head -1)
python tools/kstats_top.py "$f" 22 | tee $O/remesh_kernels.txt
REMESH=gpu python tools/recon_trend.py 2>&1 | grep -v amdgpu | tail -3 | tee $O/recon_trend_gpu.txt
