mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_kzg.py tests/test_gpu_concurrent.py tests/test_gpu_fullsize.py tests/test_gpu_mgpu.py tests/test_gpu_golden.py tests/test_gpu_ntt_large.py -x -q -m gpu 2>&1 | tail -4
python tools/prof_witness.py 20 256 > gpurun_out/r05/prof_witness2.txt 2>&1; grep -A8 "witness_batched:" gpurun_out/r05/prof_witness2.txt | head -10; grep "commit:" gpurun_out/r05/prof_witness2.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/bench2.json 2> gpurun_out/r05/bench2.err; python - <<'P'
import json
d=json.loads(open('gpurun_out/r05/bench2.json').read().strip().splitlines()[-1])
p=d['paths']
print(d['value'], {k:p[k] for k in p if 'witness' in k or k in ('commit_coeff_ms','commit_u64_per_s','ntt_2e20_ms')}, p['checked_against_oracle'])
print(d['cpu_baseline'])
P
