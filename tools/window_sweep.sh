#!/bin/bash
# batched throughput (batch 64) and single latency against the window width at sizes below 2^20 (same box)
for ln in 14 16 17 18 19; do
  for c in 10 11 12 13 14 15 16 17; do
    [ $c -lt $((ln-5)) ] && continue
    echo -n "log_n=$ln c=$c -> "
    timeout 120 python bench.py --log-n $ln --window-bits $c --no-cpu-baseline --no-paths --steps 5 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['single_commit_latency_ms'], d['config']['windows'])"
  done
done
