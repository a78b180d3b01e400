# batched rate with / without deferred tails (option defer_tail), interleaved, uniform and u64-valued coefficients, and two smaller sizes
for rep in 1 2; do
for dt in 1 0; do
  python bench.py --no-cpu-baseline --no-paths --steps 8 --opt defer_tail=$dt 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('uniform 2^20 defer_tail=$dt', d['value'], d['timed_results_checked']['ok'])"
  python bench.py --no-cpu-baseline --no-paths --steps 8 --u64 --opt defer_tail=$dt 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('u64     2^20 defer_tail=$dt', d['value'], d['timed_results_checked']['ok'])"
done
done
for dt in 1 0; do
  python bench.py --no-cpu-baseline --no-paths --steps 8 --log-n 17 --opt defer_tail=$dt 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('uniform 2^17 defer_tail=$dt', d['value'], d['timed_results_checked']['ok'])"
  python bench.py --no-cpu-baseline --no-paths --steps 8 --log-n 18 --opt defer_tail=$dt 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('uniform 2^18 defer_tail=$dt', d['value'], d['timed_results_checked']['ok'])"
done
