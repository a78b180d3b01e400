cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "heavy or wide" 2>&1 | tail -5
bash tools/ab_local.sh 3 prev base
for v in prev base; do
  if [ $v = base ]; then lib=kzg_amd/libkzg_mi355x.so; else lib=tools/bin/lib_$v.so; fi
  echo "== $v"; SKEW_PROF=1 SKEW_ONLY=uniform python3 tools/run_with_lib.py $lib tools/skew_probe.py 20 2>&1 | tail -2
done
