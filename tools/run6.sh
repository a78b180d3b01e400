cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r03_batch_size.txt
for round in 1 2; do
for b in 64 128 256 512; do
  steps=$((640 / b)); [ $steps -lt 3 ] && steps=3
  r=$(timeout 300 python bench.py --no-cpu-baseline --no-paths --batch $b --steps $steps --warmup 1 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['avg_kernel_ms'], r['frac'])")
  echo "$round batch=$b steps=$steps value ms_per_step accum_insitu_ms frac: $r" | tee -a gpurun_out/r03_batch_size.txt
done
done
