# 20-bit-window path (2^23 points and up; any size with --window-bits 20): accumulation kernel on a FIFO accumulation stream inside a
# pipeline (default) against on the lane's own stream (wide_in_lane=1, the old way).  Same box, interleaved.
run() { python bench.py --no-cpu-baseline --no-paths "$@" 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['single_commit_latency_ms'], d['timed_results_checked']['ok'])"; }
for rep in 1 2; do for o in 0 1; do
  echo "2^24 batch 4 wide_in_lane=$o $(run --log-n 24 --batch 4 --steps 3 --warmup 1 --opt wide_in_lane=$o)"
  echo "2^23 batch 8 wide_in_lane=$o $(run --log-n 23 --batch 8 --steps 3 --warmup 1 --opt wide_in_lane=$o)"
  echo "2^22 window 20 batch 16 wide_in_lane=$o $(run --log-n 22 --window-bits 20 --batch 16 --steps 3 --warmup 1 --opt wide_in_lane=$o)"
  echo "2^20 window 20 batch 64 wide_in_lane=$o $(run --log-n 20 --window-bits 20 --steps 6 --opt wide_in_lane=$o)"
done; done
