#!/bin/bash
O=gpurun_out/r06g; mkdir -p $O
python -m pytest tests/test_gpu_remesh.py tests/test_gpu_topology.py -q --maxfail=5 -rf > $O/gputest_remesh.log 2>&1; grep -v "^  File" $O/gputest_remesh.log | tail -12
python tools/ubench/remesh_probe.py 0.9 2>&1 | grep -v amdgpu | tail -4 | tee $O/remesh_probe.txt
