cd /root/repo
for r in 1 2; do for o in "" "--opt naf_window=18"; do
  echo -n "round $r [$o] -> "
  timeout 400 python3 bench.py $o --no-cpu-baseline --no-paths --steps 8 --warmup 2 --check 2>/dev/null < /dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['single_commit_latency_ms'], d.get('all_results_match_known_tau'))"
done; done
