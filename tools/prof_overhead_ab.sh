for rep in 1 2 3; do
  python bench.py --no-cpu-baseline --no-paths --steps 10 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('events on the timed region ', d['value'])"
  KZG_BENCH_NO_PROF=1 python bench.py --no-cpu-baseline --no-paths --steps 10 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('no events                  ', d['value'])"
done
