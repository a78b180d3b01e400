for rep in 1 2; do for b in 64 128 256; do
  python bench.py --no-cpu-baseline --no-paths --steps 6 --batch $b 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('uniform batch=$b', d['value'], d['timed_results_checked']['ok'])"
done; done
