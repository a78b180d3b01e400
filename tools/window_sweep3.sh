#!/bin/bash
# after the two-level sort: is c = 17 now the better width below 2^17 too?  (same box; batch 64 throughput, single-commit ms)
for ln in 13 14 15 16 17; do
  for c in 10 13 16 17; do
    echo -n "log_n=$ln c=$c -> "
    timeout 120 python3 bench.py --log-n $ln --window-bits $c --no-cpu-baseline --no-paths --steps 5 --warmup 2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['single_commit_latency_ms'], d['config']['windows'])"
  done
done
