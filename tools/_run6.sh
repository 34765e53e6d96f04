mkdir -p gpurun_out/r3n
timeout 2700 python -m pytest tests -m gpu -x -q > gpurun_out/r3n/gputests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3n/gputests.log
tail -4 gpurun_out/r3n/gputests.log
