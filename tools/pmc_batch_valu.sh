#!/bin/bash
# VALU wave-instructions per kernel of the BATCHED pipeline (deep-batch tail variants), u64-valued and uniform coefficients:
#   bash tools/pmc_batch_valu.sh  -> gpurun_out/pmc_batch/{u64,uniform}.summary.json
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/pmc_batch
rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/u64 -o p --output-format csv -- python3 bench.py --u64 --steps 1 --warmup 1 --batch 32 --no-cpu-baseline --no-paths > $O/u64.json 2> $O/u64.log
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/uniform -o p --output-format csv -- python3 bench.py --steps 1 --warmup 1 --batch 32 --no-cpu-baseline --no-paths > $O/uniform.json 2> $O/uniform.log
for d in u64 uniform; do python3 tools/pmc_summary.py $O/$d > $O/$d.summary.json; done
python3 - <<'PY'
import json
for d in ("u64", "uniform"):
    s = json.load(open("gpurun_out/pmc_batch/%s.summary.json" % d))
    rows = sorted(((v["SQ_INSTS_VALU"]["avg"] * v["SQ_INSTS_VALU"]["launches"], k, v["SQ_INSTS_VALU"]["launches"], v["SQ_INSTS_VALU"]["avg"]) for k, v in s.items() if "SQ_INSTS_VALU" in v), reverse=True)
    tot = sum(r[0] for r in rows)
    print(d, "total VALU wave-instructions %.1f M" % (tot / 1e6))
    for t, k, l, a in rows[:16]:
        print("   %-28s launches %4d  avg %10.3f M  share %5.1f %%" % (k[:28], l, a / 1e6, 100 * t / tot))
PY
