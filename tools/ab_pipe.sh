for d in 1 2 3 4; do
python bench.py --workload pipeline --detectors $d --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('detectors', $d, d['value'], d['ms_per_step'])
"
done
