#!/bin/bash
# Collects the round's measurements on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh [tag]   -> files under gpurun_out/final_<tag>/, to be copied into profiles/ (named per round)
# Passes are separate processes: bench line, rocprofv3 kernel stats of the same command, PMC FETCH_SIZE and WRITE_SIZE
# (one pass each: they do not fit one pass on gfx950), SQ counters of a single MSM and of the NTT kernels, hardware-queue
# matrix, size sweep, N>1 code path at world size 1, fuzz.  The CPU baseline's worker processes are never started under rocprofv3.
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/final_${1:-a}
rm -rf $O; mkdir -p $O
timeout 900 python3 bench.py --check > $O/bench.json 2> $O/bench.err
timeout 300 python3 bench.py --u64 --no-cpu-baseline --no-paths --check > $O/bench_u64.json 2>> $O/bench.err
MASTER_PORT=29533 timeout 300 python3 bench.py --sharded --no-cpu-baseline --check > $O/bench_sharded_world1.json 2>> $O/bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 bench.py --no-cpu-baseline --no-paths > $O/bench_under_rocprof.json 2> $O/stats.log
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats_paths -o s --output-format csv -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_paths_under_rocprof.json 2> $O/stats_paths.log
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-paths > $O/pmc_fetch.json 2> $O/pmc_fetch.log
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-paths > $O/pmc_write.json 2> $O/pmc_write.log
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_sq -o p --output-format csv -- python3 bench.py --batch 1 --steps 8 --warmup 1 --no-cpu-baseline --no-paths > $O/pmc_sq.json 2> $O/pmc_sq.log
# the NTT kernels: SQ counters, LDS conflicts, and HBM bytes (three passes of the same loop)
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_ntt_sq -o p --output-format csv -- python3 tools/ntt_loop.py 20 10 > $O/pmc_ntt_sq.txt 2> $O/pmc_ntt_sq.log
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS -d $O/pmc_ntt_lds -o p --output-format csv -- python3 tools/ntt_loop.py 20 10 > $O/pmc_ntt_lds.txt 2> $O/pmc_ntt_lds.log
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_ntt_fetch -o p --output-format csv -- python3 tools/ntt_loop.py 20 10 > $O/pmc_ntt_fetch.txt 2> $O/pmc_ntt_fetch.log
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_ntt_write -o p --output-format csv -- python3 tools/ntt_loop.py 20 10 > $O/pmc_ntt_write.txt 2> $O/pmc_ntt_write.log
for d in pmc_fetch pmc_write pmc_sq; do python3 tools/pmc_summary.py $O/$d > $O/$d.summary.json 2>/dev/null; done
for d in pmc_ntt_sq pmc_ntt_lds pmc_ntt_fetch pmc_ntt_write; do python3 tools/pmc_summary.py $O/$d k_ntt > $O/$d.summary.json 2>/dev/null; done
# hardware queues: who asks for them (separate processes, interleaved, two rounds)
hw() { echo -n "$1 -> "; shift; env "$@" KZG_DEBUG=1 timeout 200 python3 bench.py --no-cpu-baseline --no-paths --callers --steps 6 --warmup 2 2>/tmp/hwq_err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['single_commit_latency_ms'], d['paths']['blocking_callers_16_per_s'], end=' ')"; grep "kzg:" /tmp/hwq_err.txt | sort -u | tr '\n' ';'; echo; }
for rep in 1 2; do
hw "host calls kzg_init_hw_queues(0) before its first HIP call (kzg_amd.load default)" KZG_X=1
hw "host exports GPU_MAX_HW_QUEUES=24 itself, library sets nothing" KZG_HW_QUEUES=0 GPU_MAX_HW_QUEUES=24
hw "nobody asks: the runtime's default pool (4 queues), engine narrows the pipeline" KZG_HW_QUEUES=0
hw "KZG_SET_HW_QUEUES=1: the library's load-time constructor opts in" KZG_HW_QUEUES=0 KZG_SET_HW_QUEUES=1
done > $O/hw_queues.txt 2>&1
timeout 900 python3 tools/sweep.py 14 16 17 18 19 20 21 22 23 24 > $O/sweep.jsonl 2> $O/sweep.err
# round 4: the concurrent-callers mix (every leased call, oracle-checked), the NTT probe, N > 1 block at world size 1
timeout 300 python3 tools/stress_callers.py 60 18 16 > $O/stress_callers.txt 2>&1
timeout 200 python3 tools/ntt_probe.py 16 18 20 22 24 > $O/ntt_probe.txt 2>&1
timeout 400 python3 bench.py --no-cpu-baseline --no-paths --sharded-block --steps 4 > $O/bench_sharded_block_world1.json 2>> $O/bench.err
timeout 200 python3 tools/fuzz_gpu.py 90 > $O/fuzz.txt 2>&1
SKEW_PROF=1 timeout 200 python3 tools/skew_probe.py 20 > $O/skew.txt 2>&1
ls -la $O
