for rep in 1 2 3; do for t in bim16 bim32 bim64; do
  python tools/bench_with_lib.py tools/bin/lib_$t.so --no-cpu-baseline --no-paths --callers --steps 6 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); p=d['paths']; print('$t', d['value'], p['blocking_callers_16_per_s'], p['blocking_callers_16_witness_batched_k256_per_s'], p['blocking_callers_16_witness_batched_vs_value'])"
done; done
