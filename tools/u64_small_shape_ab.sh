#!/bin/bash
# u64-valued coefficients at 2^20 (2^22 sorted entries): the small-MSM pipeline shape (fewer accumulation blocks on 4 streams) against the default
O=gpurun_out/r05; mkdir -p $O; rm -f $O/u64_small_shape.txt
run() { echo -n "$* -> " >> $O/u64_small_shape.txt; python bench.py --u64 --no-paths --no-cpu-baseline --steps 12 --warmup 3 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['timed_results_checked']['ok'])" >> $O/u64_small_shape.txt; }
for rep in 1 2; do
run
run --opt small_entries=4194304
run --opt small_entries=4194304 --opt accum_blocks_small=240
run --opt small_entries=4194304 --opt accum_blocks_small=320
run --opt accum_streams=4
done
cat $O/u64_small_shape.txt
