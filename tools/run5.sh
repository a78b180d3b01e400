cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest3.log 2>&1; echo pytest rc=$?; tail -4 gpurun_out/pytest3.log)
(timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke3.log 2>&1; echo smoke rc=$?; tail -2 gpurun_out/smoke3.log)
bash tools/collect_profiles.sh a > gpurun_out/collect_a.log 2>&1
tail -30 gpurun_out/collect_a.log
cat gpurun_out/final_a/hw_queues.txt
head -c 1500 gpurun_out/final_a/bench.json
