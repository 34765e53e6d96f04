mkdir -p gpurun_out/r3a
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/gputests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3a/gputests.log
tail -5 gpurun_out/r3a/gputests.log
for v in 72 9; do
  DRT_BENCH_NOPROF=1 timeout 600 python bench.py --views $v --no-cpu-baseline --no-extras --steps 40 --warmup 5 2>gpurun_out/r3a/b$v.err | tail -1 > gpurun_out/r3a/b$v.json
  python -c "
import json;d=json.load(open('gpurun_out/r3a/b$v.json'));print('views $v', d['ms_per_step'],'ms/step', d['value'])"
done
timeout 600 python bench.py --views 9 --no-cpu-baseline --steps 20 --warmup 5 2>gpurun_out/r3a/b9full.err | tail -1 > gpurun_out/r3a/b9full.json
python tools/benchsum.py gpurun_out/r3a/b9full.json | head -40
