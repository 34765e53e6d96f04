for v in 9 18 36 72; do for g in 0 1; do echo -n "views=$v graph=$g: "; python bench.py --steps 5 --warmup 2 --no-cpu-baseline --views $v --graph $g 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); st=d['roofline']['stages']; print(d['value'], d['ms_per_step'], d['config']['final_loss'], 'kernels', round(sum(st[k]['ms_per_step'] for k in st),3))
except Exception as e: print('FAILED', e)"; done; done
