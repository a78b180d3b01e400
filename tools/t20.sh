cd /root/repo
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "wide_windows or heavy_bins" 2>&1 | tail -5
for ln in 22 24; do for c in 17 20; do
  echo -n "u64 log_n=$ln c=$c -> "
  timeout 400 python3 bench.py --u64 --log-n $ln --batch $([ $ln = 24 ] && echo 4 || echo 16) --window-bits $c --no-cpu-baseline --no-paths --steps 3 --warmup 1 --check 2>/dev/null < /dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['single_commit_latency_ms'], d['config']['windows'], d.get('all_results_match_known_tau'))"
done; done
for ln in 22 23; do for c in 17 20; do
  echo -n "full log_n=$ln c=$c -> "
  timeout 400 python3 bench.py --log-n $ln --batch 8 --window-bits $c --no-cpu-baseline --no-paths --steps 3 --warmup 1 --check 2>/dev/null < /dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['single_commit_latency_ms'], d['config']['windows'], d.get('all_results_match_known_tau'))"
done; done
