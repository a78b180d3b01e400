#!/bin/bash
# 2^22 / 2^24 points: do wider windows (the two-pass "wide" sort path) pay once n >> 2^c?
for ln in 22 24; do
  for c in 17 19 20; do
    echo -n "log_n=$ln c=$c -> "
    timeout 300 python3 bench.py --log-n $ln --batch $([ $ln = 24 ] && echo 2 || echo 8) --window-bits $c --no-cpu-baseline --no-paths --steps 3 --warmup 1 --check 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['single_commit_latency_ms'], d['config']['windows'], d.get('all_results_match_known_tau'))"
  done
done
