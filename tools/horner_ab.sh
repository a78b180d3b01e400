# same-box A/B of the Horner kernels.  HORNER_LIBS: the builds to compare (default: tools/bin/lib_r06head.so -- saturated Montgomery
# products, the state before this change -- against the tree's build)
python -m pytest tests/test_gpu_ntt_poly.py tests/test_gpu_kzg.py tests/test_gpu_golden.py -q -x 2>&1 | tail -3
LIBS=${HORNER_LIBS:-"tools/bin/lib_r06head.so kzg_amd/libkzg_mi355x.so"}
for log_n in 20 20 24; do
for lib in $LIBS; do
KZG_AMD_LIBRARY=$lib python tools/prof_witness_coeff.py $log_n 2>&1 | grep -E "library|horner|quotient_apply|wall|witness_coeff " | head -12
done
done
