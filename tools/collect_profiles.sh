#!/bin/bash
# Collects the round's measurements on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh   -> files under gpurun_out/final/, to be copied into profiles/ (named per round)
# Passes are separate processes: bench line, rocprofv3 kernel stats of the same command, PMC FETCH_SIZE and WRITE_SIZE
# (one pass each: they do not fit one pass on gfx950), SQ counters of a single MSM, size sweep, N>1 code path at world size 1.
# The CPU baseline's worker processes are never started under rocprofv3 (--no-cpu-baseline there).
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/final
rm -rf $O; mkdir -p $O
timeout 600 python3 bench.py --check > $O/bench.json 2> $O/bench.err
timeout 300 python3 bench.py --u64 --no-cpu-baseline --no-paths --check > $O/bench_u64.json 2>> $O/bench.err
MASTER_PORT=29533 timeout 300 python3 bench.py --sharded --no-cpu-baseline --check > $O/bench_sharded_world1.json 2>> $O/bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 bench.py --no-cpu-baseline --no-paths > $O/bench_under_rocprof.json 2> $O/stats.log
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats_paths -o s --output-format csv -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_paths_under_rocprof.json 2> $O/stats_paths.log
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-paths > $O/pmc_fetch.json 2> $O/pmc_fetch.log
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-paths > $O/pmc_write.json 2> $O/pmc_write.log
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_sq -o p --output-format csv -- python3 bench.py --batch 1 --steps 8 --warmup 1 --no-cpu-baseline --no-paths > $O/pmc_sq.json 2> $O/pmc_sq.log
for d in pmc_fetch pmc_write pmc_sq; do python3 tools/pmc_summary.py $O/$d > $O/$d.summary.json 2>/dev/null; done
timeout 600 python3 tools/sweep.py 16 18 20 22 24 > $O/sweep.jsonl 2> $O/sweep.err
ls -la $O
