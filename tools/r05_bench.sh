mkdir -p gpurun_out/r05
(time python bench.py --gpus 1 --steps 20 --warmup 5) > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err
echo "rc=$?"; tail -n 4 gpurun_out/r05/bench_default.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r05/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
p=d['paths']
for k,v in p.items():
    if k!='ntt_roofline': print(k, v)
print(p['ntt_roofline']['mad_frac'], p['ntt_roofline']['kernel_ms'])
print(d['cpu_baseline'])
P
