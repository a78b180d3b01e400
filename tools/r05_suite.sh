mkdir -p gpurun_out/r05
(time python -m pytest tests/ -x -q -m gpu --durations=8) > gpurun_out/r05/gpu_suite.log 2>&1
echo "suite rc=$?" >> gpurun_out/r05/gpu_suite.log
tail -n 22 gpurun_out/r05/gpu_suite.log | cut -c1-600
