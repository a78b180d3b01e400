#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X: degree-2^20 coeff-form KZG commitments/sec
(G1 Pippenger MSM, full-width uniform Fr scalars, SRS and scalars already resident in HBM).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N = 1 : workload = BASELINE configs[1] "degree-2^20 coeff_form commit (G1 Pippenger MSM) on 1xMI355X".
        One step = one batch of `--batch` independent commitments pipelined on the engine's HIP streams
        (kzg_msm_g1_batch); value = commitments / second.
N > 1 : one process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher; torch itself is never imported: the
        ranks' barriers and the max-over-ranks time travel over tools/benchlib/control.py's TCP star).  Commitments are
        independent objects and a degree-2^20 SRS is 2 GiB, so the metric shards by commitment: every rank holds the full SRS and
        commits its own polynomials, NO data-path collective ("scaling": "weak").  That is `value`.  The same run also measures
        the sharded-SRS + RCCL design through the library's device group and reports it beside `value` as `sharded.strong` /
        `sharded.config5` -- by the same ranks, in a fresh child process each, BEFORE this process touches the GPU.
        --config5 / --strong / --weak make the device group the timed region itself (kzg_commit_coeff_sharded_batch: SRS sharded
        contiguously, one 144-byte partial per rank and polynomial, ONE ncclAllGather inside the library, local sums).

The JSON line carries `roofline` for the dominant kernel (k_accum_affine; HIP-event times measured on the engine's streams
inside this process), `paths` (the other BASELINE configs, timed after the timed region) and `cpu_baseline` (the oracle's C
restatement on the host cores, rank 0, N = 1 only; worker processes are started BEFORE the GPU is initialised).
The oracle is never on the measured path.  The parts live in tools/benchlib/: headline (timed region + line), paths, traffic (live PMC
passes), sharded, control (ranks), checks and cpu_pool (the only two that touch oracle/).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from tools.benchlib.common import LOG_N  # noqa: E402


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="commitments per step (pipelined on the engine's HIP streams)")
    ap.add_argument("--streams", type=int, default=0,
                    help="lanes the engine pipelines a batch over (0 = the engine's default: 13 + 4 accumulation streams from the process' shared "
                         "stream pool, which leaves an RCCL communicator its hardware queues)")
    ap.add_argument("--accum-blocks", type=int, default=0, help="engine option accum_blocks_batch (0 = default)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="extra engine option (kzg_ctx_set_option), repeatable")
    ap.add_argument("--log-n", type=int, default=LOG_N)
    ap.add_argument("--window-bits", type=int, default=0, help="engine option window_bits (0 = engine default)")
    ap.add_argument("--u64", action="store_true", help="u64-valued coefficients (the reference benches' distribution)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sizes", default="10,16,20", help="log2 sizes of the CPU baseline's MSM legs (the 2^20 leg also carries the NTT and create_witness legs)")
    ap.add_argument("--no-paths", action="store_true", help="skip the `paths` measurements after the timed region")
    ap.add_argument("--callers", action="store_true", help="with --no-paths: still measure paths.blocking_callers_16_per_s")
    ap.add_argument("--strong", action="store_true", help="N>1: one degree-2^log_n commitment sharded N ways (device group, RCCL all-gather)")
    ap.add_argument("--config5", action="store_true", help="N>1: BASELINE configs[4], 2^21 terms per rank (degree 2^24 at N = 8)")
    ap.add_argument("--weak", action="store_true", help="N>1: 2^log_n terms per rank (degree N * 2^log_n)")
    ap.add_argument("--sharded", action="store_true",
                    help="use the device-group code path (sharded SRS, RCCL all-gather inside the library) even at world size 1")
    ap.add_argument("--check", action="store_true", help="verify EVERY commitment of the last step against [p(tau)]G (default: a sample of 4)")
    ap.add_argument("--replicas", action="store_true",
                    help="N>1 (the default there): data-parallel replicas (full SRS on every GPU, different polynomials per GPU, no collective)")
    ap.add_argument("--sharded-block", action="store_true",
                    help="measure the `sharded` block (strong + config5 through the device group) after the timed region even at world "
                         "size 1 (RCCL all-gather forced on); at N > 1 the block is part of the default run")
    ap.add_argument("--no-sharded-block", action="store_true", help="N>1: skip the `sharded` block")
    ap.add_argument("--sharded-batch", type=int, default=64, help="commitments per step of the sharded block (as --batch)")
    ap.add_argument("--sharded-steps", type=int, default=3)
    ap.add_argument("--pmc-kind", default="msm", choices=["msm", "ntt"], help=argparse.SUPPRESS)
    ap.add_argument("--pmc-child", type=int, default=0, help=argparse.SUPPRESS)   # internal: the workload of measure_traffic_pmc's rocprofv3 passes
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc passes that fill roofline.traffic")
    ap.add_argument("--sharded-timeout", type=int, default=90, help="seconds the sharded block's child process may take before it is killed and the line printed without it")
    ap.add_argument("--sharded-child", action="store_true", help=argparse.SUPPRESS)   # internal: the block alone (run_sharded_block_in_children)
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if args.pmc_child:            # internal: the workload of the live rocprofv3 --pmc passes
        from tools.benchlib import traffic
        traffic.pmc_child(args.pmc_child, args.pmc_kind)
        return
    if args.sharded_child:        # internal: the sharded block alone, one fresh process per rank
        from tools.benchlib import sharded
        sharded.sharded_child_main(args)
        return
    from tools.benchlib import headline
    headline.run(args)


if __name__ == "__main__":
    main()
