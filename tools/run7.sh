cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_naf.py tests/test_gpu_golden.py tests/test_gpu_msm.py -m gpu -x -q > gpurun_out/pytest4.log 2>&1; echo pytest rc=$?; tail -12 gpurun_out/pytest4.log)
: > gpurun_out/r03_ab_naf.txt
for round in 1 2; do
for v in "naf_window=0" "naf_window=-1"; do
  r=$(timeout 300 python bench.py --no-cpu-baseline --no-paths --steps 10 --warmup 2 --opt $v 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['avg_kernel_ms'], r['alone']['avg_kernel_ms'], d['single_commit_latency_ms'], r['frac'], r['digits_per_scalar'], r['alone']['kernel_ms_single_msm'])")
  echo "$round $v value ms_per_step accum_insitu_ms accum_alone_ms latency_ms frac digits kernels: $r" | tee -a gpurun_out/r03_ab_naf.txt
done
done
