#!/bin/bash
# batched throughput against the size of the HIP runtime's hardware-queue pool (same box, separate processes):
#   caller exports 24 / nothing exported (the library's load-time default) / the ROCm default of 4 / 8
run() { echo -n "GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-unset} $* -> "; KZG_DEBUG=1 timeout 120 python bench.py --no-cpu-baseline --no-paths --steps 6 --warmup 2 "$@" 2>/tmp/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['single_commit_latency_ms'], end=' ')"; grep "kzg:" /tmp/err.txt | sort -u | tr '\n' ';'; echo; }
for rep in 1 2; do
GPU_MAX_HW_QUEUES=24 run
run
GPU_MAX_HW_QUEUES=4 run
GPU_MAX_HW_QUEUES=8 run
done
