#!/bin/bash
# same-box sweep of bench.py argument sets: bash tools/ab_opt2.sh <rounds> "<args A>" "<args B>" ...
cd /root/repo
rounds=$1; shift
for r in $(seq $rounds); do
for a in "$@"; do
  echo -n "$r [$a] -> "
  timeout 300 python3 bench.py $a --no-cpu-baseline --no-paths --steps ${STEPS:-10} --warmup 3 2>/dev/null < /dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['single_commit_latency_ms'])"
done
done
