mkdir -p gpurun_out/r3j
timeout 2700 python -m pytest tests -m gpu -x -q > gpurun_out/r3j/gputests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3j/gputests.log
tail -4 gpurun_out/r3j/gputests.log
python tools/iter_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r3j/iter_bench.txt; cat gpurun_out/r3j/iter_bench.txt
for V in 72 9; do
  timeout 600 python bench.py --views $V --no-cpu-baseline --steps 20 --warmup 5 2>gpurun_out/r3j/f$V.err | tail -1 > gpurun_out/r3j/f$V.json
  echo "== views $V"; python tools/benchsum.py gpurun_out/r3j/f$V.json | grep -E "Mrays|build|trace2|trace3"
done
python tools/build_trace.py 2>&1 | grep -v amdgpu | head -16
