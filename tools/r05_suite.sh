mkdir -p gpurun_out/r05
(time python -m pytest tests/ -x -q -m gpu --durations=8) > gpurun_out/r05/gpu_suite.log 2>&1
echo "suite rc=$?" >> gpurun_out/r05/gpu_suite.log
(time python -c "import __graft_entry__ as g; g.smoke()") > gpurun_out/r05/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/r05/smoke.log
tail -n 22 gpurun_out/r05/gpu_suite.log | cut -c1-600; tail -n 9 gpurun_out/r05/smoke.log | cut -c1-400
