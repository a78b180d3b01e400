run() { echo -n "Q=${GPU_MAX_HW_QUEUES:-unset} PLAN=$KZG_PLAN -> "; timeout 120 python bench.py --no-cpu-baseline --no-paths --steps 6 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"; }
export GPU_MAX_HW_QUEUES=8
for p in 6,2 5,2 3,1 7,1 4,2 8,0 4,1 5,1; do KZG_PLAN=$p run; done
export GPU_MAX_HW_QUEUES=4
for p in 3,1 2,1 4,0 2,2; do KZG_PLAN=$p run; done
export GPU_MAX_HW_QUEUES=12
for p in 10,2 8,2 11,1 6,2; do KZG_PLAN=$p run; done
