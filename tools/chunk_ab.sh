# minimum equal-split chunk of k_accum_affine (-DKZG_ACCUM_MIN_CHUNK): batched rate and lone latency, builds interleaved
for rep in 1 2; do for t in c8 c64 c96; do
  python tools/bench_with_lib.py tools/bin/lib_$t.so --no-cpu-baseline --no-paths --steps 8 --u64 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$t u64     2^20', d['value'], d['single_commit_latency_ms'], d['timed_results_checked']['ok'])"
  python tools/bench_with_lib.py tools/bin/lib_$t.so --no-cpu-baseline --no-paths --steps 8 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$t uniform 2^20', d['value'], d['single_commit_latency_ms'], d['timed_results_checked']['ok'])"
done; done
for t in c8 c64 c96; do
  python tools/bench_with_lib.py tools/bin/lib_$t.so --no-cpu-baseline --no-paths --steps 8 --log-n 16 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$t uniform 2^16', d['value'], d['single_commit_latency_ms'], d['timed_results_checked']['ok'])"
  python tools/bench_with_lib.py tools/bin/lib_$t.so --no-cpu-baseline --no-paths --steps 8 --log-n 18 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$t uniform 2^18', d['value'], d['single_commit_latency_ms'], d['timed_results_checked']['ok'])"
done
