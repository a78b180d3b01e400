# batched rate by pipeline depth (lanes); hardware queues exported by the shell so that every stream can have its own
export GPU_MAX_HW_QUEUES=32
for st in 16 20 24 28; do
  python bench.py --no-cpu-baseline --no-paths --u64 --steps 8 --streams $st 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('u64 streams=$st', d['value'], d['timed_results_checked']['ok'])"
done
for st in 16 20 24; do
  python bench.py --no-cpu-baseline --no-paths --steps 8 --streams $st 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('uniform streams=$st', d['value'], d['timed_results_checked']['ok'])"
done
for st in 16 24; do
  python bench.py --no-cpu-baseline --no-paths --steps 8 --streams $st --batch 96 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('uniform batch 96 streams=$st', d['value'], d['timed_results_checked']['ok'])"
done
