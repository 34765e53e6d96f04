#!/bin/bash
# round 6, GPU call 1: length-bin probe (upper bound), computed-direction probe of the projection pass (timing only), baseline bench, GPU tests
O=gpurun_out/r06a; mkdir -p $O
python tools/ubench/length_bin_probe.py > $O/length_bin_probe.txt 2>&1
PROBE_DISTANCE=1.1 python tools/ubench/length_bin_probe.py > $O/length_bin_probe_tight.txt 2>&1
export DRT_BENCH_REPEATS=5
for r in 1 2; do
  for v in base rcomp; do
    if [ $v = base ]; then unset DRT_HIP_LIB; else export DRT_HIP_LIB=$PWD/drt_amd/_ab/$v.so; fi
    python bench.py --no-cpu-baseline --steps 20 > $O/ab_${v}_$r.json 2> $O/ab_${v}_$r.err
    python - $O/ab_${v}_$r.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
st=r['stages']; al=r.get('stages_alone_avg_launch_ms') or {}
t=d.get('tight_framing',{})
print(sys.argv[2], 'step', d['ms_per_step'], 'raster in step', st['raster']['ms_per_step'], 'alone/launch', al.get('raster'), 'launches', st['raster']['launches'],
      '| tight step', t.get('ms_per_step'), 'raster', (t.get('stages_ms_per_step') or {}).get('raster'), 'alone', (t.get('stages_alone_ms_per_step') or {}).get('raster'),
      '| establish', d.get('establish_mode',{}).get('ms_per_step'))
PY
  done
done > $O/raster_ab.txt 2>&1
unset DRT_HIP_LIB DRT_BENCH_REPEATS
python bench.py --no-cpu-baseline > $O/bench_sustained.json 2> $O/bench_sustained.err
python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1
tail -3 $O/gputest.log; cat $O/length_bin_probe.txt $O/length_bin_probe_tight.txt $O/raster_ab.txt
python -c "
import json; d=json.loads(open('$O/bench_sustained.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['repeats']['first_repeat'], d['repeats']['by_half_second'], d['repeats']['n'])"
