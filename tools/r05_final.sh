bash tools/collect_profiles.sh r05c > gpurun_out/collect_r05c.log 2>&1
mkdir -p gpurun_out/r05
(time python -m pytest tests/ -x -q -m gpu --durations=6) > gpurun_out/r05/gpu_suite_final.log 2>&1
echo "suite rc=$?" >> gpurun_out/r05/gpu_suite_final.log
(time NCCL_SOCKET_IFNAME=nonexistent0 python -m pytest tests/ -x -q -m gpu -rs) > gpurun_out/r05/suite_broken_rccl_final.log 2>&1
echo "rc=$?" >> gpurun_out/r05/suite_broken_rccl_final.log
(time python -c "import __graft_entry__ as g; g.smoke()") > gpurun_out/r05/smoke_final.log 2>&1
echo "smoke rc=$?" >> gpurun_out/r05/smoke_final.log
tail -n 5 gpurun_out/r05/gpu_suite_final.log; grep -E "passed|failed" gpurun_out/r05/suite_broken_rccl_final.log | tail -2; tail -n 4 gpurun_out/r05/smoke_final.log
