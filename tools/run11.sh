cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/pytest6.log 2>&1; echo pytest rc=$?; tail -14 gpurun_out/pytest6.log)
(timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke6.log 2>&1; echo smoke rc=$?; tail -2 gpurun_out/smoke6.log)
bash tools/collect_profiles.sh b > gpurun_out/collect_b.log 2>&1
tail -3 gpurun_out/collect_b.log
cat gpurun_out/final_b/hw_queues.txt
head -c 600 gpurun_out/final_b/bench.json
