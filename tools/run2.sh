set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_golden.py tests/test_gpu_kzg.py -m gpu -x -q > gpurun_out/pytest2.log 2>&1; echo rc=$?; tail -5 gpurun_out/pytest2.log)
(timeout 900 python tools/ab_libs.py tools/bin/lib_base.so tools/bin/lib_nonops.so tools/bin/lib_merged.so > gpurun_out/r03_ab_nops_merged.txt 2>&1; echo rc=$?; cat gpurun_out/r03_ab_nops_merged.txt)
