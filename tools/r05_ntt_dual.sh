O=gpurun_out/r05; mkdir -p $O; rm -f $O/ab_ntt_dual2.txt
for l in 20 16 18 21 22 24; do python tools/ab_ntt.py tools/bin/lib_ntt_xcd.so tools/bin/lib_ntt_d1.so tools/bin/lib_ntt_d3.so $l >> $O/ab_ntt_dual2.txt 2>&1; done
cat $O/ab_ntt_dual2.txt
python -m pytest tests/test_gpu_ntt_poly.py tests/test_gpu_golden.py tests/test_gpu_ntt_large.py tests/test_gpu_kzg.py -x -q -m gpu 2>&1 | tail -3
