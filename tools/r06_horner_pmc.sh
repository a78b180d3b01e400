#!/bin/bash
# SQ counters of the three quotient kernels (k_horner_partials / k_horner_scan / k_quotient_apply) at 2^20 and 2^24:
# rocprofv3 --pmc over tools/prof_witness_coeff.py (kernels serialise under --pmc; counts per launch are what is read)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/horner_pmc
rm -rf $O; mkdir -p $O
for ln in 20 24; do
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d $O/sq$ln -o p --output-format csv -- python3 tools/prof_witness_coeff.py $ln > $O/sq$ln.txt 2> $O/sq$ln.log
python3 tools/pmc_summary.py $O/sq$ln k_horner k_quotient_apply > $O/sq${ln}.json 2>/dev/null
done
ls -la $O
