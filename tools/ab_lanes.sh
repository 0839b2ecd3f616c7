for l in 2 3 4 6; do
python bench.py --workload embed --legs "" --no-cpu-baseline --lanes $l 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes', $l, d['value'], d['ms_per_step'])
"
done
