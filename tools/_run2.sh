mkdir -p gpurun_out/r3b
timeout 2700 python -m pytest tests -m gpu -x -q > gpurun_out/r3b/gputests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3b/gputests.log
tail -5 gpurun_out/r3b/gputests.log
( time timeout 900 python bench.py > gpurun_out/r3b/bench.json 2> gpurun_out/r3b/bench.err ) 2>&1 | tail -3
tail -3 gpurun_out/r3b/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3b/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'])
for k in ('paths','establish_mode','tight_framing','fused_mode'): print(k, d.get(k))
print(d['config'].get('loss_check'))
r=d['roofline']; print({k:r.get(k) for k in ('achieved','frac','frac_source','pmc_stale','valu_instr_per_launch','valu_instr_per_launch_est','wave_steps_per_launch','est')})
PY
