#!/bin/bash
# default window choice: batched throughput (batch 64) and single latency by size (same box)
for ln in 8 10 12 13 14 15 16 17 18 19 20; do
  echo -n "log_n=$ln -> "
  timeout 120 python bench.py --log-n $ln --no-cpu-baseline --no-paths --steps 5 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['single_commit_latency_ms'], d['config']['window_bits'], d['config']['windows'])"
done
