#!/bin/bash
# round 6: 13 lanes against 14 for the plain batched pipeline and for 16 blocking callers (the group's exchange stream needs the 14th lane's hardware queue)
for rep in 1 2; do for st in 14 13; do
python bench.py --no-cpu-baseline --no-paths --callers --steps 8 --warmup 2 --streams $st 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams=$st value', d['value'], 'lone commit ms', d['single_commit_latency_ms'], 'callers16', d['paths']['blocking_callers_16_per_s'], 'witness_batched callers', d['paths'].get('blocking_callers_16_witness_batched_k256_per_s'))"
done; done
