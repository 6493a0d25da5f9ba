for rep in 1 2; do for k in 2 3 4; do python bench.py --no-frames --no-cfg5 --no-cpu-baseline --steps 200 --warmup 20 --in-flight $k 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in-flight $k:', round(d['value'],1), round(d['ms_per_step']*1e3,2), 'us/step; single', round(d.get('value_single_stream',0),1))"; done; done
python bench.py --no-frames --no-cfg5 --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default 20 steps:', round(d['value'],1), round(d['ms_per_step']*1e3,2))"
