for rep in 1 2; do for dt in 1 0; do for ln in 16 14; do
  python bench.py --no-cpu-baseline --no-paths --steps 10 --log-n $ln --opt defer_tail=$dt 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('uniform 2^$ln defer_tail=$dt', d['value'], d['timed_results_checked']['ok'])"
done; done; done
