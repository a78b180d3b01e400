#!/bin/bash
# Collects the round's measurements on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh   -> files under gpurun_out/final/, to be copied into profiles/
# Passes are separate processes: bench line, rocprofv3 kernel stats of the same command, PMC FETCH_SIZE and WRITE_SIZE
# (one pass each: they do not fit one pass on gfx950), SQ counters of a single MSM, per-path timings, size sweep.
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/final
mkdir -p $O
python3 bench.py --check --cpu-all-cores > $O/bench.json 2> $O/bench.err
python3 bench.py --u64 --no-cpu-baseline --check > $O/bench_u64.json 2>> $O/bench.err
MASTER_PORT=29533 python3 bench.py --sharded --no-cpu-baseline --check > $O/bench_sharded_world1.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.json 2> $O/pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write.json 2> $O/pmc_write.log
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_sq -o p --output-format csv -- python3 bench.py --batch 1 --steps 8 --warmup 1 --no-cpu-baseline > $O/pmc_sq.json 2> $O/pmc_sq.log
python3 tools/bench_paths.py 20 --check > $O/paths_2e20.json 2> $O/paths.err
python3 tools/sweep.py 16 18 20 22 24 > $O/sweep.jsonl 2> $O/sweep.err
./tools/bin/microbench > $O/microbench.txt 2>&1
ls -la $O
