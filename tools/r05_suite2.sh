mkdir -p gpurun_out/r05
(time python -m pytest tests/test_gpu_zz_environment.py tests/test_gpu_mgpu.py tests/test_gpu_kzg.py -x -q -m gpu --durations=8) > gpurun_out/r05/zz.log 2>&1
echo "rc=$?" >> gpurun_out/r05/zz.log
# a box on which RCCL cannot bootstrap: the interface it is told to use does not exist
(time NCCL_SOCKET_IFNAME=nonexistent0 python -m pytest tests/ -x -q -m gpu -rs) > gpurun_out/r05/suite_broken_rccl.log 2>&1
echo "rc=$?" >> gpurun_out/r05/suite_broken_rccl.log
tail -n 30 gpurun_out/r05/zz.log | cut -c1-1500; tail -n 40 gpurun_out/r05/suite_broken_rccl.log | cut -c1-800
