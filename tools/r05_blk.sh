mkdir -p gpurun_out/r05
python bench.py --no-cpu-baseline --no-paths --sharded-block --steps 4 > gpurun_out/r05/blk.json 2> gpurun_out/r05/blk.err; echo "rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r05/blk.json').read().strip().splitlines()[-1])
sh=d['sharded']; print(d['value'], sh.get('formation'), {k:sh[k]['value'] for k in ('strong','config5') if k in sh}, sh.get('child'), sh.get('note'), sh.get('error'))
P
python -m pytest tests/test_gpu_bench_multi.py -x -q -m gpu 2>&1 | tail -4
