for cfg in "hand 0" "horse 0" "horse 1" "horse 2"; do set -- $cfg; DRT_STREAMS=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --mesh $1 --subdiv $2 --views 24 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['roofline']['stages']
print(d['config']['workload'][:60], d['ms_per_step'])
for k in ('trace1','trace2','trace3'):
    s=st[k]; rays=s['items_per_launch']*s['launches']/3; v=s.get('node_visits_per_ray',0)
    print('   ',k,'ms',s['ms_per_step'],'rays/step',int(rays),'visits/ray',v,'ps per lane-visit', round(1e9*s['ms_per_step']*1e-3/max(1,rays*v)*1e3,2), 'util', s.get('lane_utilisation'))"; done
