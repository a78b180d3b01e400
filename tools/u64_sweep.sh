# u64-valued coefficients (the reference benches' distribution): batched rate by window width / accumulation grid
for wb in 0 16 15 14 13; do
  python bench.py --no-cpu-baseline --no-paths --u64 --steps 8 --window-bits $wb 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('u64 window_bits=$wb', d['value'], d['single_commit_latency_ms'], d['config']['windows'], d['timed_results_checked']['ok'])"
done
