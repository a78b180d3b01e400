mkdir -p gpurun_out/r05
(time python -m pytest tests/ -x -q -m gpu --durations=12) > gpurun_out/r05/gpu_suite.log 2>&1
echo "suite rc=$?" >> gpurun_out/r05/gpu_suite.log
(time python -c "import __graft_entry__ as g; g.smoke()") > gpurun_out/r05/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/r05/smoke.log
tail -n 30 gpurun_out/r05/gpu_suite.log | cut -c1-1200; tail -n 12 gpurun_out/r05/smoke.log
