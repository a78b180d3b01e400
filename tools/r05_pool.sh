mkdir -p gpurun_out/r05
python tools/engine_and_group_ab.py 20 64 > gpurun_out/r05/engine_and_group3.txt 2>&1; cut -c1-330 gpurun_out/r05/engine_and_group3.txt
(time python -m pytest tests/ -x -q -m gpu --durations=6) > gpurun_out/r05/gpu_suite3.log 2>&1; tail -n 14 gpurun_out/r05/gpu_suite3.log | cut -c1-600
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > gpurun_out/r05/bench4_$i.json 2> gpurun_out/r05/bench4.err; python - <<P
import json
d=json.loads(open('gpurun_out/r05/bench4_$i.json').read().strip().splitlines()[-1])
p=d['paths']
print(d['value'], d['roofline']['frac'], {k:p[k] for k in p if ('callers' in k and 'per_s' in k) or k in ('commit_u64_per_s','witness_batched_k256_ms','commit_coeff_ms')}, p['commit_2e16']['commitments_per_s'], p['commit_2e24']['commitments_per_s'])
P
done
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-paths --streams 16 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams 16:', d['value'])"
