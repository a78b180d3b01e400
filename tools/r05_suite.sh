mkdir -p gpurun_out/r05
(time python -m pytest tests/ -x -q -m gpu --durations=25) > gpurun_out/r05/gpu_suite.log 2>&1
echo "suite rc=$?" >> gpurun_out/r05/gpu_suite.log
(time python -c "import __graft_entry__ as g; g.smoke()") > gpurun_out/r05/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/r05/smoke.log
tail -n 45 gpurun_out/r05/gpu_suite.log; tail -n 8 gpurun_out/r05/smoke.log
