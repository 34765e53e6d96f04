for cfg in "2 1 24" "2 2 23" "2 3 22" "3 2 22" "2 4 21"; do set -- $cfg; echo -n "streams=$1 per=$2 minlog=$3: "; DRT_STREAMS=$1 DRT_SUB_PER_STREAM=$2 DRT_MIN_SUB_LOG2=$3 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
for v in 36 18 9; do echo -n "default views=$v: "; python bench.py --steps 5 --warmup 2 --no-cpu-baseline --views $v 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
