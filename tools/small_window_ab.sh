# window width at small sizes with the small-MSM pipeline shape (same box, interleaved)
run() { python bench.py --no-cpu-baseline --no-paths --steps 8 "$@" 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['single_commit_latency_ms'], d['config']['windows'], d['timed_results_checked']['ok'])"; }
for rep in 1 2; do for ln in 14 15 16 17; do for wb in 0 12 13 14 15 16 17; do
  echo "2^$ln window_bits=$wb $(run --log-n $ln --window-bits $wb)"
done; done; done
