for rm in 1 4 16 32; do
echo -n "refill=$rm: "; DRT_REFILL_MIN=$rm python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['roofline']['stages']
print(d['ms_per_step'], {k:(st[k]['ms_per_step'], st[k].get('lane_utilisation')) for k in ('trace1','trace2','trace3') if k in st})"
done
