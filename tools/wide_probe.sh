#!/bin/bash
# lone 2^20 MSMs at window width 17 (default) and 20: per-kernel durations (rocprofv3 kernel stats), to see what the accumulation
# saves and the tail costs apart from the sort
cd /root/repo
export TMPDIR=/tmp
for c in 0 20; do
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/wide_c$c -o s --output-format csv -- python3 bench.py --window-bits $c --batch 1 --no-cpu-baseline --no-paths --steps 10 --warmup 2 > gpurun_out/wide_c$c.json 2> gpurun_out/wide_c$c.log < /dev/null
  f=$(find gpurun_out/wide_c$c -name '*kernel_stats.csv' | head -1)
  echo "== c=$c"
  [ -n "$f" ] && head -30 "$f" | cut -d, -f1-5
done
