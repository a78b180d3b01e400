# Small MSMs in the pipeline: 160-block accumulation grid on 4 accumulation streams (the default since round 4, options
# accum_blocks_small / accum_streams_small / small_entries) against the old shape (accum_blocks_small=0: 480 blocks on 2 streams).
# Same box, interleaved.  The 2^18 / 2^19 lines with small_entries raised show where the rule stops paying.
run() { python bench.py --no-cpu-baseline --no-paths --steps 8 "$@" 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['timed_results_checked']['ok'])"; }
for rep in 1 2; do
  for ln in 12 14 15 16 17; do
    echo "2^$ln old     $(run --log-n $ln --opt accum_blocks_small=0)"
    echo "2^$ln default $(run --log-n $ln)"
  done
  for ln in 18 19; do
    echo "2^$ln default (not small) $(run --log-n $ln)"
    echo "2^$ln small_entries=2^23  $(run --log-n $ln --opt small_entries=8388608)"
    echo "2^$ln small_entries=2^23 blocks 240 $(run --log-n $ln --opt small_entries=8388608 --opt accum_blocks_small=240)"
  done
  echo "2^20 default $(run)"
  echo "2^20 old     $(run --opt accum_blocks_small=0)"
  echo "2^20 u64 default $(run --u64)"
  echo "2^20 u64 old     $(run --u64 --opt accum_blocks_small=0)"
done
