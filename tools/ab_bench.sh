#!/bin/bash
# same-box A/B of several library builds through bench.py (separate processes, interleaved): tools/ab_bench.sh libA.so libB.so ...
for round in 1 2 3; do
  for lib in "$@"; do
    echo -n "$(basename $lib) "
    python tools/bench_with_lib.py $lib --no-cpu-baseline --no-paths --steps 8 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['single_commit_latency_ms'], d['roofline']['alone']['avg_kernel_ms'])"
  done
done
