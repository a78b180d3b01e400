#!/bin/bash
# same-box A/B of library variants: tools/ab_local.sh <rounds> <variant> ...   (base = the shipped library)
cd /root/repo
rounds=$1; shift
for r in $(seq $rounds); do
for v in "$@"; do
  if [ $v = base ]; then lib=kzg_amd/libkzg_mi355x.so; else lib=tools/bin/lib_$v.so; fi
  echo -n "$r $v value ms_per_step latency_ms: "
  timeout 300 python3 tools/bench_with_lib.py $lib --no-cpu-baseline --no-paths --steps ${STEPS:-10} --warmup 3 2>/dev/null < /dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['single_commit_latency_ms'])"
done
done
