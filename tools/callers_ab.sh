# blocking callers (16 threads): plan with 4 accumulation streams (default) against 2 (accum_streams_small=2), same box, interleaved
run() { python bench.py --no-cpu-baseline --steps 4 --no-traffic "$@" 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); p=d['paths']; print(d['value'], p['blocking_callers_16_per_s'], p['blocking_callers_16_host_resident_per_s'], p['blocking_callers_16_witness_batched_k256_per_s'], p['witness_batched_k256_ms'])"; }
for rep in 1 2 3; do
  echo "default (16 + 4)          $(run)"
  echo "accum_streams_small=2     $(run --opt accum_streams_small=2)"
  echo "streams=14 (14 + 4)       $(run --streams 14)"
done
