cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_kzg.py -m gpu -x -q > gpurun_out/pytest7.log 2>&1; echo pytest rc=$?; tail -3 gpurun_out/pytest7.log)
: > gpurun_out/r03_ab_stagger.txt
for round in 1 2 3; do
for v in unsig stagger; do
  r=$(timeout 300 python tools/bench_with_lib.py tools/bin/lib_$v.so --no-cpu-baseline --no-paths --steps 12 --warmup 3 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['avg_kernel_ms'], r['alone']['avg_kernel_ms'], d['single_commit_latency_ms'])")
  echo "$round $v value ms_per_step accum_insitu_ms accum_alone_ms latency_ms: $r" | tee -a gpurun_out/r03_ab_stagger.txt
done
done
