#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X: degree-2^20 coeff-form KZG commitments/sec
(G1 Pippenger MSM, full-width uniform Fr scalars, SRS and scalars already resident in HBM).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N = 1 : workload = BASELINE configs[1] "degree-2^20 coeff_form commit (G1 Pippenger MSM) on 1xMI355X".
        One step = one batch of `--batch` independent commitments pipelined on the engine's HIP
        streams (kzg_msm_g1_batch); value = commitments / second.
N > 1 : the SRS is sharded contiguously, 2^20 terms per rank (polynomial of N*2^20 coefficients, the
        shape of configs[4]); each step every rank reduces its shard on its GPU, the 96-byte partials
        are all-gathered over RCCL/xGMI and summed locally.  value = (terms processed by all ranks /
        2^20) per second, i.e. degree-2^20-equivalent commitments/s (weak scaling, no work skipped).

The JSON line also carries `roofline` for the dominant kernel (k_accum_affine; HIP-event time measured
on the engine's stream inside this process) and `cpu_baseline` (the oracle's single-threaded C
Pippenger on a bounded sample, rank 0, N = 1 only).  The oracle is never on the measured path.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# One hardware queue per engine stream: 16 MSM lanes + 2 accumulation streams (the ROCm default multiplexes all streams
# onto 4 in-order queues, which makes independent MSM lanes wait for each other's tail kernels); must be in the
# environment before HIP initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

LOG_N = 20
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BYTES_PER_TERM = 128           # 32 B scalar + 96 B affine point per MSM term
# The resource that actually binds k_accum_affine: VALU issue, dominated by v_mad_i64_i32.  One bucket
# addition executes 6 mul30 (338 mads) + 2 sqr30 (260) + one fused double product with a single reduction (507) = 3055
# mads per lane (static count from the ISA; 4379 VALU instructions in all by SQ_INSTS_VALU).  tools/microbench.hip
# measured the chip's 64-bit multiply-add rate: 31.5 T lane-op/s at 8 waves/SIMD, 23.7 T at the 2 waves/SIMD a
# 215-VGPR kernel holds (profiles/r01_microbench.txt).
MADS_PER_ADD = 6 * 338 + 2 * 260 + 507
MAD_PEAK_TLANE_S = 31.51
MAD_PEAK_OCC2_TLANE_S = 23.70
TAU = 0x5EED5EED5EED5EED       # known secret for the synthetic SRS (setup(s, n), src/lib.rs:38)


def g1_adds_per_msm(n, c, W):
    """algorithmic G1 additions, n*W bucket accumulations + bucket reduction."""
    return n * W + 2 * (1 << (c - 1))


def cpu_baseline(engine, params, scal, log_n):
    """Oracle C Pippenger (single thread, like the reference's multi_exp) on one whole polynomial of the timed
    workload: the first min(2^log_n, 2^20) coefficients of batch entry 0, same SRS (about 20 s of CPU at 2^20)."""
    from oracle import c_oracle as C
    n = 1 << min(log_n, 20)
    pts = params.gs.download(0, n)
    sc = scal.download(n)
    if scal.sfmt != 1:  # the oracle takes canonical little-endian scalars
        raise RuntimeError("cpu_baseline expects canonical scalars")
    t0 = time.perf_counter()
    out = C.msm_g1_raw(pts, sc, n)
    dt = time.perf_counter() - t0
    ok = engine.msm(params.gs, scal, n=n) == out   # the baseline run doubles as a parity check
    terms_per_s = n / dt
    return {"value": terms_per_s / (1 << log_n), "unit": "commitments/s",
            "cores": 1, "kind": "port",
            "sample": f"one 2^{min(log_n, 20)}-term MSM = polynomial 0 of the timed batch, same SRS, {dt:.2f} s, "
                      f"{terms_per_s:.0f} terms/s; matches GPU result: {ok}"}


def _cpu_chunk(args):
    pts, sc, n = args
    from oracle import c_oracle as C
    return C.msm_g1_raw(pts, sc, n)


def cpu_baseline_all_cores(engine, params, scal, log_n):
    """The same oracle MSM split into one contiguous chunk per host core (processes), partial points added at the end."""
    import multiprocessing as mp
    from oracle import c_oracle as C
    n = 1 << min(log_n, 20)
    # chunks below ~2^15 terms make the bucket method inefficient (measured: 256 chunks of 4096 take 2.75 s, 5x the
    # per-term cost), so at most n / 2^15 worker processes are used
    cores = max(1, min(os.cpu_count() or 1, n >> 15))
    pts = params.gs.download(0, n)
    sc = scal.download(n)
    per = (n + cores - 1) // cores
    chunks = [(pts[96 * i:96 * min(i + per, n)], sc[32 * i:32 * min(i + per, n)], min(i + per, n) - i) for i in range(0, n, per)]
    with mp.get_context("fork").Pool(cores) as pool:
        pool.map(_cpu_chunk, chunks[:1])  # warm the workers (library load)
        t0 = time.perf_counter()
        parts = pool.map(_cpu_chunk, chunks)
        acc = parts[0]
        for p_ in parts[1:]:
            acc = C.g1_add(acc, p_)
        dt = time.perf_counter() - t0
    ok = engine.msm(params.gs, scal, n=n) == acc
    return {"value": n / dt / (1 << log_n), "unit": "commitments/s", "cores": cores, "kind": "port",
            "sample": f"one 2^{min(log_n, 20)}-term MSM in {len(chunks)} chunks over {cores} processes, {dt:.2f} s; "
                      f"matches GPU result: {ok}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="commitments per step (pipelined on the engine's HIP streams)")
    ap.add_argument("--streams", type=int, default=16, help="HIP streams the engine pipelines a batch over (0 = engine default, 8)")
    ap.add_argument("--accum-blocks", type=int, default=0, help="engine option accum_blocks (0 = default)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="extra engine option (kzg_ctx_set_option), repeatable")
    ap.add_argument("--log-n", type=int, default=LOG_N)
    ap.add_argument("--window-bits", type=int, default=0, help="engine option window_bits (0 = engine default)")
    ap.add_argument("--u64", action="store_true", help="u64-valued coefficients (the reference benches' distribution)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sharded", action="store_true",
                    help="use the N>1 code path (process group, sharded SRS, RCCL all_gather) even at world size 1")
    ap.add_argument("--check", action="store_true", help="verify the timed result against [p(tau)]G (known-tau identity)")
    ap.add_argument("--replicas", action="store_true",
                    help="N>1 only: data-parallel replicas (full SRS on every GPU, different polynomials per GPU, no "
                         "collective in the data path) instead of the sharded-SRS mode (SURVEY 8e, throughput alternative)")
    ap.add_argument("--cpu-all-cores", action="store_true",
                    help="also time the oracle MSM split over all host cores (extra field cpu_baseline_all_cores)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n = 1 << args.log_n

    import torch
    import kzg_amd
    from kzg_amd import _lib as L

    dist = None
    sharded = (world > 1 and not args.replicas) or args.sharded
    if sharded or world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), rank=rank, world_size=world)
    engine = kzg_amd.Engine(local_rank)
    if args.window_bits:
        engine.set_option("window_bits", args.window_bits)
    if args.streams:
        engine.set_option("streams", args.streams)
    if args.accum_blocks:
        engine.set_option("accum_blocks_batch", args.accum_blocks)
    for kv in args.opt:
        key, val = kv.split("=")
        engine.set_option(key, int(val))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- inputs, resident in HBM before the timed region -----------------------------------------
    if not sharded:
        params = kzg_amd.setup(engine, TAU, n, g2_len=0)                       # gs[i] = [tau^i]G
        srs = params.gs
        scal = engine.alloc_scalars(n * args.batch).fill_random(1 + 1000 * rank, u64_valued=args.u64)
    else:
        # rank r holds the contiguous shard gs[r*n .. (r+1)*n) = [tau^(r*n + i)]G of setup(tau, world*n)
        params = kzg_amd.KZGParams(kzg_amd.setup_shard(engine, TAU, rank * n, n))
        srs = params.gs
        scal = engine.alloc_scalars(n * args.batch).fill_random(1 + 1000 * rank, u64_valued=args.u64)
    c, W = srs.window_info()
    out = ctypes.create_string_buffer(96 * max(args.batch, 1))

    last = {}
    if not sharded:
        def step():
            rc = engine.lib.kzg_msm_g1_batch(engine.ctx, srs.handle, 0, scal.ptr, n, args.batch, scal.sfmt,
                                             L.IN_DEVICE, out, L.G1_AFFINE_MONT)
            if rc:
                raise RuntimeError(engine.last_error())
        units_per_step = world * args.batch   # replicas: every rank commits its own batch
    else:
        from kzg_amd.distributed import ShardedCommitter
        committer = ShardedCommitter.for_engine(engine, srs, dist, rank, world, max_batch=args.batch, always_gather=True)

        def step():
            last["commitments"] = committer.commit_batch(scal, args.batch)
        units_per_step = world * args.batch   # world * n terms per polynomial = `world` degree-2^20 equivalents each

    for _ in range(args.warmup):
        step()
    barrier()
    if rank == 0:
        engine.prof_enable(True)     # HIP events on the engine's streams around every kernel of the timed region
        engine.prof_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    check = None
    if args.check:
        # known-tau identity: commitment == [p(tau)]G.  p(tau) = sum_r tau^(r*n) * p_r(tau) from per-rank GPU Horner
        # evaluations; the single scalar multiplication of G is done by the oracle (checker only, outside the timing).
        from oracle import c_oracle as C
        R = kzg_amd.api.R_MODULUS
        mine = pow(TAU, rank * n, R) * engine.poly_eval(scal, TAU, n=n) % R if sharded else None  # polynomial 0 of the batch
        if sharded:
            vals = [None] * world
            if world > 1:
                dist.all_gather_object(vals, mine)
            else:
                vals = [mine]
            want = C.g1_mul(C.g1_generator(), sum(vals) % R)
            check = bool(last.get("commitments", [None])[0] == want)
        else:
            want = C.g1_mul(C.g1_generator(), engine.poly_eval(scal, TAU, n=n))
            check = bool(out.raw[:96] == want)

    # ---- roofline of the dominant kernel: HIP events recorded on the engine's streams over the timed region ----
    roofline = None
    latency_ms = None
    if rank == 0:
        prof = engine.prof_all()
        engine.prof_enable(False)
        launches, total_ms = prof.get("k_accum_affine", (0, 0.0))
        if launches:
            avg_s = total_ms / launches / 1e3
            achieved = BYTES_PER_TERM * n / avg_s / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                try:
                    traffic = json.load(open(tpath)).get("k_accum_affine_bytes_per_launch")
                except Exception:
                    traffic = None
            per_msm = launches  # one k_accum_affine launch per MSM
            roofline = {"bound": "hbm", "kernel": "k_accum_affine", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                        "algorithmic_bytes_per_launch": BYTES_PER_TERM * n, "launches": launches,
                        "avg_kernel_ms": round(avg_s * 1e3, 4),
                        "note": "binding resource is integer VALU issue (~4.9e4 32-bit multiply-adds per term), not HBM; "
                                "see DESIGN.md 3.2",
                        "kernel_ms_per_msm": {k: round(v[1] / per_msm, 4) for k, v in sorted(prof.items())}}
            # informational: the same kernel against the integer-multiply issue rate, over the whole timed region
            # (launches overlap on 8 streams, so the aggregate rate is the meaningful one)
            mads = float(launches) * n * (4 if args.u64 else W) * MADS_PER_ADD  # u64-valued scalars: 4 non-zero 16-bit windows
            t_mad = mads / dt / 1e12
            roofline["valu"] = {"resource": "v_mad_i64_i32 issue", "achieved": round(t_mad, 2), "unit": "T lane-mad/s",
                                "peak": MAD_PEAK_TLANE_S, "frac": round(t_mad / MAD_PEAK_TLANE_S, 4),
                                "peak_at_2_waves_per_simd": MAD_PEAK_OCC2_TLANE_S,
                                "frac_of_occupancy_2_peak": round(t_mad / MAD_PEAK_OCC2_TLANE_S, 4),
                                "mads_per_bucket_add": MADS_PER_ADD}
        # single-commit latency (one MSM alone on the GPU), outside the timed region
        one = ctypes.create_string_buffer(96)
        reps = 2
        t1 = time.perf_counter()
        for _ in range(reps):
            rc = engine.lib.kzg_msm_g1(engine.ctx, srs.handle, 0, scal.ptr, n, scal.sfmt, L.IN_DEVICE, one, L.G1_AFFINE_MONT)
            if rc:
                raise RuntimeError(engine.last_error())
        latency_ms = (time.perf_counter() - t1) / reps * 1e3
        # the same kernel with nothing else on the GPU (the batched figure above divides by a duration that is stretched
        # by the other MSMs in flight): two more single MSMs with the engine's event profiling on
        if roofline is not None:
            engine.prof_enable(True)
            engine.prof_reset()
            for _ in range(reps):
                engine.lib.kzg_msm_g1(engine.ctx, srs.handle, 0, scal.ptr, n, scal.sfmt, L.IN_DEVICE, one, L.G1_AFFINE_MONT)
            l2, ms2 = engine.prof_all().get("k_accum_affine", (0, 0.0))
            engine.prof_enable(False)
            if l2:
                a2 = BYTES_PER_TERM * n / (ms2 / l2 / 1e3) / 1e9
                roofline["alone"] = {"avg_kernel_ms": round(ms2 / l2, 4), "achieved": round(a2, 2), "unit": "GB/s",
                                     "frac": round(a2 / HBM_PEAK_GBS, 5)}

    if rank == 0:
        value = units_per_step * args.steps / dt
        res = {
            "metric": "commitments/sec + MSM G1-adds/sec at degree 2^20, 1/2/4/8 MI355X",
            "value": round(value, 3),
            "unit": "commitments/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 limbs (Fq 381-bit / Fr 255-bit Montgomery integer arithmetic)",
            "data": "synthetic",
            "config": {
                "workload": (("degree-2^%d coeff_form commit (G1 Pippenger MSM) on 1xMI355X, batch of %d per step"
                              % (args.log_n, args.batch)) if world == 1 else
                             ("degree-2^%d coeff_form commit, %d data-parallel replicas (full SRS per GPU, batch of %d per "
                              "rank and step, no data-path collective)" % (args.log_n, world, args.batch))) if not sharded else
                            ("degree-%d*2^%d coeff_form commit, SRS sharded 2^%d terms per rank over %d GPUs, batch of %d "
                             "per step, one RCCL all_gather of the 144-B Jacobian partials + local sums"
                             % (world, args.log_n, args.log_n, world, args.batch)),
                "scalars": "u64-valued Fr" if args.u64 else "uniform full-width Fr (SplitMix64 counter stream, seed 1)",
                "terms_per_rank": n, "window_bits": c, "windows": W, "srs": "setup(tau, n) generated on the GPU",
                "inputs_resident_in_hbm": True,
            },
            "g1_adds_per_sec": round(value * g1_adds_per_msm(n, c, W), 1),
            "msm_terms_per_sec": round(value * n, 1),
            "single_commit_latency_ms": None if latency_ms is None else round(latency_ms, 4),
        }
        if check is not None:
            res["result_matches_known_tau"] = check
        if world > 1:
            res["unit_note"] = "N>1: value = (terms processed by all ranks / 2^20) per second (degree-2^20 equivalents)"
        if roofline:
            res["roofline"] = roofline
        if world == 1 and not sharded and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(engine, params, scal, args.log_n)
            except Exception as e:  # the baseline must never take the bench line down
                res["cpu_baseline"] = {"value": None, "unit": "commitments/s", "cores": 1, "kind": "port",
                                       "sample": f"failed: {e}"}
            if args.cpu_all_cores:
                try:
                    res["cpu_baseline_all_cores"] = cpu_baseline_all_cores(engine, params, scal, args.log_n)
                except Exception as e:
                    res["cpu_baseline_all_cores"] = {"value": None, "sample": f"failed: {e}"}
        line = json.dumps(res)
    # RCCL writes a version banner through C stdio, which is block-buffered on a pipe and would surface after Python's own
    # output when a process exits: every rank flushes it before the last barrier, so that rank 0's JSON line ends the output
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    engine.close()
    if rank == 0:
        print(line, flush=True)


if __name__ == "__main__":
    main()
