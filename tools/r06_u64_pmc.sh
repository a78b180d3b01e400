#!/bin/bash
# (hw_queues=24: under --pmc kernels are serialised, the queue probe would find ONE queue and the engine would fall back to the lone-MSM kernel shapes)
# round 6: where the vector-instruction work of a u64-valued 2^20 commitment goes (PMC pass of tools/u64_probe.py): wave-instructions per kernel and MSM
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_u64; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc -o p --output-format csv -- python3 tools/u64_probe.py hw_queues=24 > $O/pmc.txt 2> $O/pmc.log
python3 tools/pmc_summary.py $O/pmc k_ > $O/pmc_u64.summary.json; rm -rf $O/pmc
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r06_u64/pmc_u64.summary.json"))
rows=[]
for k,v in d.items():
    if "SQ_INSTS_VALU" in v:
        rows.append((v["SQ_INSTS_VALU"]["avg"]*v["SQ_INSTS_VALU"]["launches"], v["SQ_INSTS_VALU"]["launches"], v["SQ_INSTS_VALU"]["avg"], k))
tot=sum(r[0] for r in rows)
for t,l,a,k in sorted(rows,reverse=True)[:16]:
    print("%-40s launches %5d  VALU wave-instr per launch %12.0f  share of all %.3f" % (k[:40], l, a, t/tot))
PY
tail -3 $O/pmc.txt
# the same for full-width scalars (the headline)
PROBE_FULL_WIDTH=1 timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmcf -o p --output-format csv -- python3 tools/u64_probe.py hw_queues=24 > $O/pmc_full.txt 2> $O/pmc_full.log
python3 tools/pmc_summary.py $O/pmcf k_ > $O/pmc_full.summary.json; rm -rf $O/pmcf
python3 -c "
import json
d=json.load(open('gpurun_out/r06_u64/pmc_full.summary.json'))
for k,v in sorted(d.items(), key=lambda kv: -kv[1].get('SQ_INSTS_VALU',{}).get('avg',0))[:14]:
    if 'SQ_INSTS_VALU' in v: print('%-36s launches %5d  VALU wave-instr per launch %12.0f' % (k[:36], v['SQ_INSTS_VALU']['launches'], v['SQ_INSTS_VALU']['avg']))
"
