for im in 1 8 16 32 48; do
echo -n "inner_min=$im: "; DRT_INNER_MIN=$im python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['roofline']['stages']
print(d['ms_per_step'], {k:(st[k]['ms_per_step'], st[k].get('lane_utilisation'), st[k].get('node_visits_per_ray')) for k in ('trace1','trace2','trace3') if k in st})"
done
