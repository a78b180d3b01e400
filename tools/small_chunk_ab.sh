# small sizes (2^16, 2^18): minimum equal-split chunk of k_accum_affine x number of accumulation streams, builds interleaved
for rep in 1 2; do for t in c8 c16 c32; do for as in 2 3 4; do for ln in 16 18; do
  python tools/bench_with_lib.py tools/bin/lib_$t.so --no-cpu-baseline --no-paths --steps 8 --log-n $ln --opt accum_streams=$as 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$t accum_streams=$as 2^$ln', d['value'], d['single_commit_latency_ms'], d['timed_results_checked']['ok'])"
done; done; done; done
