# device-group path at world 1 (RCCL all-gather forced on): streams of the context vs hardware queues left beside RCCL's
run() { MASTER_PORT=29533 python bench.py --no-cpu-baseline --no-paths --sharded --steps 6 "$@" 2>/tmp/gq_err.txt | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['timed_results_checked']['ok'], end=' ')"; grep "kzg: p" /tmp/gq_err.txt | sort -u | tr '\n' ';'; echo; }
for rep in 1 2; do
  echo "streams=16 (16 lanes + 4 planned)   $(KZG_DEBUG=1 run --streams 16)"
  echo "streams=16 accum_streams_small=2    $(KZG_DEBUG=1 run --streams 16 --opt accum_streams_small=2)"
  echo "group default (14 + 4)              $(KZG_DEBUG=1 run)"
  echo "streams=12 (12 + 4)                 $(KZG_DEBUG=1 run --streams 12)"
  echo "plain path, no group (16 + 4)       $(python bench.py --no-cpu-baseline --no-paths --steps 6 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'])")"
done
