for rep in 1 2; do for ab in 0 496 504 512 464; do
  python bench.py --no-cpu-baseline --no-paths --steps 8 --accum-blocks $ab 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('uniform accum_blocks_batch=$ab', d['value'], d['timed_results_checked']['ok'])"
done; done
