cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r03_ab_prio.txt
for round in 1 2 3; do
for v in prio0 prio2 prio3; do
  r=$(timeout 300 python tools/bench_with_lib.py tools/bin/lib_$v.so --no-cpu-baseline --no-paths --callers --steps 12 --warmup 3 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['avg_kernel_ms'], r['alone']['avg_kernel_ms'], d['single_commit_latency_ms'], d['paths']['blocking_callers_16_per_s'])")
  echo "$round $v value ms_per_step accum_insitu_ms accum_alone_ms latency_ms callers16: $r" | tee -a gpurun_out/r03_ab_prio.txt
done
done
