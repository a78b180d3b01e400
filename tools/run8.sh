cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r03_ab_naf_sizes.txt
for ln in 17 18 19 20; do
for v in "naf_window=0" "naf_window=18"; do
  r=$(timeout 300 python bench.py --no-cpu-baseline --no-paths --steps 6 --warmup 2 --log-n $ln --opt $v 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; a=r['alone']; print(d['value'], r['avg_kernel_ms'], a['avg_kernel_ms'], d['single_commit_latency_ms'], r['digits_per_scalar'], round(a['avg_kernel_ms']*1e9/( (1<<$ln) * r['digits_per_scalar']),2), 'ps/entry')")
  echo "2^$ln $v value accum_insitu_ms accum_alone_ms latency_ms digits: $r" | tee -a gpurun_out/r03_ab_naf_sizes.txt
done
done
