cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_naf.py tests/test_gpu_golden.py tests/test_gpu_msm.py -x -q -m gpu 2>&1 | tail -5
SKEW_PROF=1 python3 tools/skew_probe.py 20 2>&1 | tail -20
