set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 300 tools/bin/mad_issue > gpurun_out/r03_mad_issue.txt 2>&1; echo rc=$?)
(timeout 2400 python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/pytest1.log 2>&1; echo rc=$?)
tail -40 gpurun_out/pytest1.log
(timeout 900 python bench.py > gpurun_out/r03_bench_a.json 2> gpurun_out/r03_bench_a.err; echo rc=$?)
tail -5 gpurun_out/r03_bench_a.err
cat gpurun_out/r03_bench_a.json | head -c 3000
