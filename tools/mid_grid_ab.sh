# 2^18 / 2^19 and u64-valued 2^20: accumulation grid x accumulation streams (same box, interleaved)
run() { python bench.py --no-cpu-baseline --no-paths --steps 8 "$@" 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['timed_results_checked']['ok'])"; }
for rep in 1 2; do for cfg in "0 2" "384 3" "320 3" "320 4" "240 4" "480 3"; do
  set -- $cfg
  echo "2^18 blocks=$1 streams=$2 $(run --log-n 18 --accum-blocks $1 --opt accum_streams=$2)"
  echo "2^19 blocks=$1 streams=$2 $(run --log-n 19 --accum-blocks $1 --opt accum_streams=$2)"
  echo "u64 2^20 blocks=$1 streams=$2 $(run --u64 --accum-blocks $1 --opt accum_streams=$2)"
done; done
