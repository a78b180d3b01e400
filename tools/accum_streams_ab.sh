for as in 2 3 4; do
  python bench.py --no-cpu-baseline --no-paths --steps 8 --u64 --opt accum_streams=$as 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('u64 accum_streams=$as', d['value'], d['timed_results_checked']['ok'])"
  python bench.py --no-cpu-baseline --no-paths --steps 8 --opt accum_streams=$as 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('uniform accum_streams=$as', d['value'], d['timed_results_checked']['ok'])"
done
for as in 3; do for st in 14 12; do
  python bench.py --no-cpu-baseline --no-paths --steps 8 --u64 --opt accum_streams=$as --streams $st 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('u64 accum_streams=$as streams=$st', d['value'], d['timed_results_checked']['ok'])"
done; done
