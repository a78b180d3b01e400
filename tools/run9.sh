cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_golden.py tests/test_gpu_naf.py tests/test_gpu_fullsize.py tests/test_gpu_concurrent.py -m gpu -x -q > gpurun_out/pytest5.log 2>&1; echo pytest rc=$?; tail -4 gpurun_out/pytest5.log)
: > gpurun_out/r03_ab_unsigned.txt
for round in 1 2 3; do
for v in merged unsig; do
  r=$(timeout 300 python tools/bench_with_lib.py tools/bin/lib_$v.so --no-cpu-baseline --no-paths --steps 12 --warmup 3 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['avg_kernel_ms'], r['alone']['avg_kernel_ms'], d['single_commit_latency_ms'], r['peak_measured_this_run'])")
  echo "$round $v value ms_per_step accum_insitu_ms accum_alone_ms latency_ms mad_peak: $r" | tee -a gpurun_out/r03_ab_unsigned.txt
done
done
