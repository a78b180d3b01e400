#!/bin/bash
# round 6: the fused NTT kernels -- the NTT / polynomial / witness GPU tests, then PMC passes of a 2^20 loop (separate runs per counter group)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_ntt
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ntt_large.py tests/test_gpu_ntt_poly.py tests/test_gpu_golden.py tests/test_gpu_kzg.py -m gpu -x -q > $O/tests.txt 2>&1
tail -5 $O/tests.txt
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_ntt_sq -o p --output-format csv -- python3 tools/ntt_loop.py 20 10 > $O/pmc_ntt_sq.txt 2> $O/pmc_ntt_sq.log
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS -d $O/pmc_ntt_lds -o p --output-format csv -- python3 tools/ntt_loop.py 20 10 > $O/pmc_ntt_lds.txt 2> $O/pmc_ntt_lds.log
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_ntt_fetch -o p --output-format csv -- python3 tools/ntt_loop.py 20 10 > $O/pmc_ntt_fetch.txt 2> $O/pmc_ntt_fetch.log
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_ntt_write -o p --output-format csv -- python3 tools/ntt_loop.py 20 10 > $O/pmc_ntt_write.txt 2> $O/pmc_ntt_write.log
for d in pmc_ntt_sq pmc_ntt_lds pmc_ntt_fetch pmc_ntt_write; do python3 tools/pmc_summary.py $O/$d k_ntt > $O/$d.summary.json 2>/dev/null; rm -rf $O/$d; done
cat $O/pmc_ntt_sq.summary.json $O/pmc_ntt_lds.summary.json $O/pmc_ntt_fetch.summary.json $O/pmc_ntt_write.summary.json | python3 -c "
import sys,json,re
txt=sys.stdin.read()
for m in re.finditer(r'\"(k_ntt[^\"]*)\": \{(.*?)\n \}', txt, re.S):
    vals=dict((a,float(b)) for a,b in re.findall(r'\"([A-Z_]+)\": \{\s*\"avg\": ([0-9.e+]+)', m.group(2)))
    print(m.group(1), {k: round(v) for k,v in vals.items()})
"
