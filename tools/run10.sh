cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r03_tune.txt
run() { r=$(timeout 300 python bench.py --no-cpu-baseline --no-paths --steps 10 --warmup 2 $1 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"); echo "[$1] $r" | tee -a gpurun_out/r03_tune.txt; }
for round in 1 2; do
run ""
run "--accum-blocks 448"
run "--accum-blocks 464"
run "--accum-blocks 496"
run "--accum-blocks 512"
run "--opt accum_streams=3"
run "--opt accum_streams=1"
run "--batch 128 --steps 5"
done
