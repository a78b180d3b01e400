#!/bin/bash
# 112-byte vs 128-byte (one cache line) table rows on the final kernel: throughput A/B + FETCH_SIZE PMC
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r02_traffic
mkdir -p $O
timeout 600 tools/ab_bench.sh tools/bin/lib_r2c_row112.so tools/bin/lib_r2c_row128.so > $O/ab.txt 2>&1
for v in 112 128; do
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch_$v -o p --output-format csv -- python3 tools/bench_with_lib.py tools/bin/lib_r2c_row$v.so --steps 2 --warmup 1 --no-cpu-baseline --no-paths > $O/fetch_$v.json 2> $O/fetch_$v.log
  python3 tools/pmc_summary.py $O/fetch_$v k_accum_affine k_scan_a k_scatter > $O/fetch_$v.summary.json
done
cat $O/ab.txt; cat $O/fetch_112.summary.json $O/fetch_128.summary.json
