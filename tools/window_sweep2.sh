#!/bin/bash
for ln in 8 10 12 13 15; do
  for c in 6 8 9 10 11 12 13 16 17; do
    echo -n "log_n=$ln c=$c -> "
    timeout 120 python bench.py --log-n $ln --window-bits $c --no-cpu-baseline --no-paths --steps 5 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['single_commit_latency_ms'], d['config']['windows'])"
  done
done
for ln in 16 18; do for c in 12 13 14 17; do echo -n "u64 log_n=$ln c=$c -> "; timeout 120 python bench.py --u64 --log-n $ln --window-bits $c --no-cpu-baseline --no-paths --steps 5 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['single_commit_latency_ms'])"; done; done
