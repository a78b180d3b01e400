#!/bin/bash
# same-box A/B of engine options through bench.py (separate processes, interleaved):
#   tools/ab_opt.sh "" "--opt sort_single_pass=1" ...     (each argument = extra bench.py flags of one variant)
for round in 1 2 3; do
  for v in "$@"; do
    echo -n "[$v] "
    timeout 300 python3 bench.py $v --no-cpu-baseline --no-paths --steps 8 --warmup 2 --check 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['single_commit_latency_ms'], d['roofline']['alone']['avg_kernel_ms'], d.get('check'))"
  done
done
