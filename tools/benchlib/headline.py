"""The timed region of bench.py and the JSON line built around it: BASELINE configs[1] (degree-2^20 coeff-form commit, batches of
kzg_msm_g1_batch on scalars and an SRS already resident in HBM), the roofline of its dominant kernel from HIP events on the engine's
streams, and the other readings (`paths`, `cpu_baseline`, `sharded`) attached after the timer has stopped.

Nothing in this module imports oracle/: the timed results are checked, after the timed region, through tools.benchlib.checks, and the
CPU baseline runs in tools.benchlib.cpu_pool's worker processes."""
import ctypes
import json
import os
import sys
import time

from . import checks
from .common import *  # noqa: F401,F403
from .control import Job
from .cpu_pool import CpuBaseline
from .paths import measure_blocking_callers, measure_paths, measure_spots, measure_u64
from .sharded import run_sharded_block_in_children
from .traffic import measure_traffic_pmc


def gpu_side_of_cpu_legs(kzg_amd, L, engine, params, scal, n_s, cpu_sizes):
    """What the GPU produces for the inputs the CPU baseline's legs run on (tools.benchlib.cpu_pool compares bytes): the commitment
    of the first 2^k coefficients of polynomial 0 for every k in --cpu-sizes, the forward NTT of its first n_s coefficients (sha256 of
    the canonical outputs) and one create_witness at a point on the polynomial."""
    import hashlib
    top = n_s.bit_length() - 1
    gpu = {"msm": {}, "ntt_sha256": None, "witness": None}
    for k in sorted({top} | {int(v) for v in cpu_sizes.split(",") if v.strip()}):
        if 0 < k <= top:
            gpu["msm"][k] = engine.msm(params.gs, view(kzg_amd, scal, 0, 1 << k), n=1 << k)
    coeffs = view(kzg_amd, scal, 0, n_s)
    ev = engine.alloc_scalars(n_s)
    ev.upload(coeffs.download())
    engine.ntt(ev, top)
    gpu["ntt_sha256"] = hashlib.sha256(ev.download()).hexdigest()
    ev.free()
    R = kzg_amd.api.R_MODULUS
    x = kzg_amd.splitmix_scalar(99, 0)
    y = engine.poly_eval(coeffs, x)
    out = ctypes.create_string_buffer(96)
    rc = engine.lib.kzg_witness_coeff(engine.ctx, params.gs.handle, coeffs.ptr, n_s, (x % R).to_bytes(32, "little"), (y % R).to_bytes(32, "little"),
                                      coeffs.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
    if rc == 0:
        gpu["witness"] = (x, y, out.raw)
    return gpu


def run(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    sharded = args.sharded or (world > 1 and (args.strong or args.config5 or args.weak) and not args.replicas)
    want_block = (args.sharded_block or (world > 1 and not args.no_sharded_block)) and not sharded

    # stdout carries exactly ONE line, the JSON record.  RCCL prints a banner (ROCm version / hostname / library path) through C
    # stdio when a communicator is created, and that buffer is flushed at process exit -- after anything Python printed.  So
    # file descriptor 1 is pointed at stderr for the whole run (this process, its native libraries, its children) and the
    # record goes to the original stdout at the very end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    # The sharded-SRS + RCCL modes (north_star's design; at N > 1 part of the default run), measured by the ranks' CHILD processes
    # before this process initialises HIP (run_sharded_block_in_children); the result joins the line at the end.
    sharded_res = run_sharded_block_in_children(args, rank, local_rank, world) if want_block else None

    cpu = None
    if rank == 0 and world == 1 and not sharded and not args.no_cpu_baseline:
        try:
            cpu = CpuBaseline()
            cpu.start()          # before HIP: the workers are spawned from a process that has not touched the GPU
        except Exception as e:   # the baseline must never take the bench line down
            cpu = None
            cpu_err = str(e)

    # No torch at any world size: the library is the only thing that loads a HIP runtime (the system's), the ranks' control plane is
    # tools.benchlib.control's TCP star.
    job = Job(rank, local_rank, world)
    local_rank = job.local_rank
    import kzg_amd
    from kzg_amd import _lib as L

    # ---- the device group first (it decides the mode: if the group cannot be formed on this node the run degrades to
    # data-parallel replicas and says so, instead of producing no number at all)
    group, group_note = None, None
    if sharded:
        from kzg_amd.api import DeviceGroup
        from kzg_amd.distributed import shard_range
        ok = 1
        try:
            uid = job.broadcast_object(DeviceGroup.unique_id)      # rank 0 draws the RCCL unique id, the control plane carries the 128 bytes
            group = DeviceGroup.for_rank(local_rank, rank, world, uid)
        except Exception as e:  # noqa: BLE001
            ok, group_note = 0, f"device group could not be formed ({e}); fell back to replicas"
        ok = job.all_agree(ok)  # all ranks agree on the outcome
        if not ok:
            if group is not None:
                group.close()
            group, sharded = None, False
            group_note = group_note or "device group could not be formed on another rank; fell back to replicas"

    # ---- which polynomial, which slice of it this rank holds -------------------------------------------------
    if not sharded:
        mode = "single" if world == 1 else "replicas"
        n_poly = 1 << args.log_n
        lo, hi = 0, n_poly
    else:
        if args.config5:
            mode, n_poly = "config5", world << 21
        elif args.weak:
            mode, n_poly = "weak", world << args.log_n
        else:
            mode, n_poly = "strong", 1 << args.log_n
        lo, hi = shard_range(n_poly, rank, world)
    n_local = hi - lo

    if sharded:
        if world == 1:
            group.set_option("always_gather", 1)     # --sharded at N = 1 exercises the RCCL exchange
        if os.environ.get("KZG_GATHER_TIMEOUT_MS"):  # experiments: the exchange wait's deadline (0 = plain hipStreamSynchronize)
            group.set_option("gather_timeout_ms", int(os.environ["KZG_GATHER_TIMEOUT_MS"]))
        engine = group.engine(0)
    else:
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

    job.engines.append(engine)
    barrier = job.barrier

    # ---- inputs, resident in HBM before the timed region -----------------------------------------
    # polynomial b of the batch = elements of the counter stream seeded SEED + 1000 b (replicas: + 10^6 rank); a rank holds
    # coefficients [lo, hi) of each, laid out [batch][hi - lo]
    def poly_seed(b):
        return SEED + 1000 * b + (1_000_000 * rank if mode == "replicas" else 0)

    scal = engine.alloc_scalars(max(n_local, 1) * args.batch)
    for b in range(args.batch):
        view(kzg_amd, scal, b * n_local, n_local).fill_random(poly_seed(b) + 4 * lo, u64_valued=args.u64)
    if sharded:
        msrs = group.setup(TAU, n_poly)                          # rank r generates gs[lo_r, hi_r) on its GPU
        srs, first = msrs.shard(0)
        assert first == lo and len(srs) == n_local
        params = None
    else:
        params = kzg_amd.setup(engine, TAU, n_poly, g2_len=0)    # gs[i] = [tau^i]G
        srs = params.gs
    c, W = srs.window_info()
    digits = float(W)              # sorted entries per scalar
    if c == 18:                    # positional tables: the NAF digit count depends on the scalars -- measured on a sample of the input
        digits = naf18_avg_digits(view(kzg_amd, scal, 0, min(n_local, 4096)).download())
    elif args.u64:
        digits = 4.0               # u64-valued scalars: 4 non-zero 16/17-bit windows
    out = ctypes.create_string_buffer(96 * max(args.batch, 1))

    if not sharded:
        def step():
            rc = engine.lib.kzg_msm_g1_batch(engine.ctx, srs.handle, 0, scal.ptr, n_poly, args.batch, scal.sfmt,
                                             L.IN_DEVICE, out, L.G1_AFFINE_MONT)
            if rc:
                raise RuntimeError(engine.last_error())
        units_per_step = world * args.batch   # replicas: every rank commits its own batch
    else:
        ptrs = (ctypes.c_void_p * 1)(scal.ptr.value)

        def step():
            rc = group.lib.kzg_commit_coeff_sharded_batch(group.handle, msrs.handle, ptrs, n_poly, args.batch, scal.sfmt,
                                                          L.IN_DEVICE, out, L.G1_AFFINE_MONT)
            if rc:
                raise RuntimeError(group.last_error())
        units_per_step = args.batch           # every commitment involves all ranks

    for _ in range(args.warmup):
        step()
    barrier()
    if rank == 0 and not os.environ.get("KZG_BENCH_NO_PROF"):   # (the variable: what do the events themselves cost? profiles/r04_prof_overhead.txt)
        # HIP events on the engine's streams over the timed region, around the dominant kernel only: events around all ~14 kernels
        # of every MSM cost 1.4 % of `value` (profiles/r04_prof_overhead.txt); the other kernels' durations come from one more,
        # untimed, fully instrumented step below
        engine.prof_enable(2)
        engine.prof_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = job.max_over_ranks(time.perf_counter() - t0)

    # known-tau identity for the commitments of the last timed step: C_b == [p_b(tau)]G, p_b(tau) = sum_r tau^(lo_r) p_{b,r}(tau).
    # Checker = the oracle throughout: each rank downloads its coefficient slices, the oracle's Horner loop evaluates them, the
    # oracle multiplies G.  No HIP kernel on the right-hand side.  Always a sample of four (first, last, two in between: ~0.5 s
    # at 2^20); --check: every commitment of the step.  Outside the timed region.
    which = list(range(args.batch)) if args.check else sorted({0, args.batch // 3, 2 * args.batch // 3, args.batch - 1})
    try:
        checked_ok = checks.check_known_tau(kzg_amd, job, scal, n_local, lo, out.raw, which, spans_ranks=sharded)
    except Exception as e:  # noqa: BLE001
        checked_ok = "check failed to run: %s" % e
    check = checked_ok if args.check else None

    # ---- roofline of the dominant kernel: HIP events recorded on the engine's streams over the timed region ----
    roofline = None
    latency_ms = None
    mad_peak = MAD_PEAK_TLANE_S
    if rank == 0:
        prof = engine.prof_all()
        if prof.get("k_accum_affine", (0, 0.0))[0] and not sharded:   # (a sharded step is collective: rank 0 cannot take one alone)
            engine.prof_enable(True)     # one untimed step with events around every kernel (kernel_ms_per_msm of the pipeline)
            engine.prof_reset()
            step()
            prof_all_kernels = engine.prof_all()
        else:
            prof_all_kernels = prof
        engine.prof_enable(False)
        # the roofline peak, measured on THIS device in this run (~30 ms mad-issue loop, 8 waves per SIMD; and at the 2 waves
        # per SIMD the accumulation kernel holds), right after the timed region
        pk, pk2 = ctypes.c_double(), ctypes.c_double()
        peak_measured = peak2_measured = None
        if engine.lib.kzg_measure_mad_issue_rate(engine.ctx, 8, ctypes.byref(pk)) == 0 and pk.value > 0:
            peak_measured = mad_peak = pk.value
        if engine.lib.kzg_measure_mad_issue_rate(engine.ctx, 2, ctypes.byref(pk2)) == 0 and pk2.value > 0:
            peak2_measured = pk2.value
        launches, total_ms = prof.get("k_accum_affine", (0, 0.0))
        if launches:
            avg_s = total_ms / launches / 1e3
            adds_per_launch = n_local * digits
            mads_per_launch = float(adds_per_launch) * MADS_PER_ADD
            # HBM traffic needs PMC counters (rocprofv3 --pmc passes, tools/collect_profiles.sh); nothing in this process can
            # measure it, so the line carries null and names the profile that holds the collected figure
            traffic = None
            traffic_profile = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath) and n_local == (1 << 20) and not args.u64:
                try:
                    traffic_profile = {"bytes_per_launch": json.load(open(tpath)).get("k_accum_affine_bytes_per_launch"),
                                       "source": "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, collected separately; "
                                                 "not measured in this run)"}
                except Exception:
                    traffic_profile = None
            t_mad = launches * mads_per_launch / dt / 1e12       # over the whole timed region (launches overlap on streams)
            hbm_achieved = BYTES_PER_TERM * n_local / avg_s / 1e9
            roofline = {
                "bound": "valu", "kernel": "k_accum_affine", "resource": "v_mad_i64_i32 issue (integer VALU)",
                "achieved": round(t_mad, 2), "peak": round(mad_peak, 2), "unit": "T lane-mad/s", "frac": round(t_mad / mad_peak, 4),
                # what `frac` is a fraction OF: the integer multiply-add issue rate of this device measured in this run -- not HBM
                # (the HBM reading of the same kernel is `hbm` below: 128 B per term / kernel duration / 8 TB/s)
                "frac_kind": "valu_mad_issue_measured",
                # launches x avg_kernel_ms <= steps x ms_per_step x overlap: `overlap` accumulation kernels run side by side on the
                # engine's FIFO streams, so avg_kernel_ms is an IN-SITU duration; exclusive_kernel_ms (filled below) is the same
                # kernel alone on the GPU
                "overlap": 2, "exclusive_kernel_ms": None,
                "peak_measured_this_run": None if peak_measured is None else round(peak_measured, 2),
                "peak_reference": MAD_PEAK_TLANE_S, "frac_of_peak_reference": round(t_mad / MAD_PEAK_TLANE_S, 4),
                # the measured peak moves +-7 % with the box and its thermal state; the nominal issue rate does not
                "peak_nominal": round(MAD_NOMINAL_TLANE_S, 2), "frac_of_nominal": round(t_mad / MAD_NOMINAL_TLANE_S, 4),
                "peak_at_2_waves_per_simd_measured_this_run": None if peak2_measured is None else round(peak2_measured, 2),
                "traffic": traffic, "traffic_profiled": traffic_profile,
                "digits_per_scalar": round(digits, 3),
                "derivation": "launches x terms x digits per scalar x %d mads per bucket addition (6 mul30 x 338 + 2 sqr30 x 260 + 1 fused "
                              "muladd 507; 13 x 30-bit signed limbs) / wall time of the timed region; peak = the v_mad_i64_i32 issue "
                              "rate of THIS device measured in this run (kzg_measure_mad_issue_rate: 8 chains per lane, 8 waves per "
                              "SIMD, ~30 ms); peak_reference = round 1's figure from another box" % MADS_PER_ADD,
                "peak_at_2_waves_per_simd": MAD_PEAK_OCC2_TLANE_S, "frac_of_occupancy_2_peak": round(t_mad / MAD_PEAK_OCC2_TLANE_S, 4),
                "mads_per_bucket_add": MADS_PER_ADD, "launches": launches, "avg_kernel_ms": round(avg_s * 1e3, 4),
                "hbm": {"bound": "hbm", "achieved": round(hbm_achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(hbm_achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                        "algorithmic_bytes_per_launch": BYTES_PER_TERM * n_local,
                        "note": "128 B per term / in-situ kernel duration (two accumulation kernels share the GPU in the batched "
                                "pipeline); the kernel is VALU-bound, see DESIGN.md 3.2"},
                "kernel_ms_per_msm": {k: round(v[1] / max(prof_all_kernels.get("k_accum_affine", (1, 0))[0], 1), 4) for k, v in sorted(prof_all_kernels.items())},
                "kernel_ms_per_msm_note": "one untimed step with HIP events around every kernel; avg_kernel_ms / launches above: the timed region"}
        # single-commit latency (one MSM alone on the GPU = what a blocking KZGProver::commit call sees), outside the timed region
        one = ctypes.create_string_buffer(96)
        if not sharded:
            def single():
                rc = engine.lib.kzg_msm_g1(engine.ctx, srs.handle, 0, scal.ptr, n_poly, scal.sfmt, L.IN_DEVICE, one, L.G1_AFFINE_MONT)
                if rc:
                    raise RuntimeError(engine.last_error())
            latency_ms = timeit(single, reps=5, warm=4)
            if roofline is not None:
                engine.prof_enable(True)
                engine.prof_reset()
                for _ in range(3):
                    single()
                pa = engine.prof_all()
                engine.prof_enable(False)
                l2, ms2 = pa.get("k_accum_affine", (0, 0.0))
                if l2:
                    k_s = ms2 / l2 / 1e3
                    roofline["exclusive_kernel_ms"] = round(ms2 / l2, 4)
                    roofline["alone"] = {"avg_kernel_ms": round(ms2 / l2, 4),
                                         "valu_frac": round(mads_per_launch / k_s / 1e12 / mad_peak, 4),
                                         "hbm_gbs": round(BYTES_PER_TERM * n_local / k_s / 1e9, 2),
                                         "hbm_frac": round(BYTES_PER_TERM * n_local / k_s / 1e9 / HBM_PEAK_GBS, 5),
                                         "kernel_ms_single_msm": {k: round(v[1] / l2, 4) for k, v in sorted(pa.items())}}

    if rank == 0:
        value = units_per_step * args.steps / dt
        workloads = {
            "single": "degree-2^%d coeff_form commit (G1 Pippenger MSM) on 1xMI355X, batch of %d per step" % (args.log_n, args.batch),
            "replicas": "degree-2^%d coeff_form commit, %d data-parallel replicas (full SRS per GPU, batch of %d per rank and step, "
                        "no data-path collective)" % (args.log_n, world, args.batch),
            "strong": "degree-2^%d coeff_form commit, every commitment's terms and the SRS sharded %d ways (%d terms per rank), batch "
                      "of %d per step, one RCCL all-gather of the 144-B partials inside the library + local sums (strong scaling)"
                      % (args.log_n, world, n_local, args.batch),
            "config5": "degree-%d (= %d x 2^21) coeff_form commit, SRS sharded 2^21 terms per rank over %d GPUs (BASELINE configs[4] is "
                       "N = 8: degree 2^24), batch of %d per step, RCCL all-gather of the partials inside the library"
                       % (n_poly, world, world, args.batch),
            "weak": "degree-%d (= %d x 2^%d) coeff_form commit, SRS sharded 2^%d terms per rank, batch of %d per step, RCCL "
                    "all-gather of the partials inside the library" % (n_poly, world, args.log_n, args.log_n, args.batch),
        }
        res = {
            "metric": "commitments/sec + MSM G1-adds/sec at degree 2^20, 1/2/4/8 MI355X",
            "value": round(value, 3),
            "unit": "commitments/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong" if mode == "strong" else "weak",
            "vs_baseline": None,
            "dtype": "u32 limbs (Fq 381-bit / Fr 255-bit Montgomery integer arithmetic)",
            "data": "synthetic",
            "config": {
                "workload": workloads[mode], "mode": mode, "polynomial_coefficients": n_poly,
                "scalars": "u64-valued Fr" if args.u64 else "uniform full-width Fr (SplitMix64 counter stream)",
                "terms_per_rank": n_local, "window_bits": c, "windows": W, "srs": "setup(tau, n) generated on the GPU",
                "table": ("positional: 255 rows 2^j P per point, width-18 NAF digits (%.2f per scalar)" % digits) if c == 18 else
                         "%d window rows 2^(%d w) P per point" % (srs.table_rows() if hasattr(srs, "table_rows") else W, c),
                "inputs_resident_in_hbm": True,
            },
            **({"note": group_note} if group_note else {}),
            "g1_adds_per_sec": round(value * (world if mode != "replicas" else 1) * g1_adds_per_msm(n_local, c, digits), 1),
            "msm_terms_per_sec": round(value * n_poly, 1),
            "parity_pin": "fr-literal+known-tau+published-points",   # G1 layer: no literal vector in the reference
            "single_commit_latency_ms": None if latency_ms is None else round(latency_ms, 4),
            "blocking_commit_per_s": None if latency_ms is None else round(1e3 / latency_ms, 2),
        }
        if check is not None:
            res["all_results_match_known_tau"] = check
        res["timed_results_checked"] = {"against": "[p(tau)]G, p(tau) by the oracle's Horner loop on the downloaded coefficients",
                                        "commitments_of_last_step": which if len(which) <= 8 else "all %d" % len(which),
                                        "every_rank": world > 1, "ok": checked_ok}
        res["hip_runtime"] = job.runtime(L)
        if roofline:
            res["roofline"] = roofline
        t_extra = time.perf_counter()
        if mode == "single" and not args.no_paths:
            try:
                res["paths"] = measure_paths(kzg_amd, L, engine, params, scal, n_poly, args.log_n, mad_peak=mad_peak)
                per_s, same = measure_blocking_callers(kzg_amd, L, engine, srs, scal, n_poly, args.batch)
                res["paths"]["blocking_callers_16_per_s"] = round(per_s, 2)
                res["paths"]["blocking_callers_16_vs_value"] = round(per_s / value, 4)
                res["paths"]["blocking_callers_16_match_batch_results"] = same
                per_s, same = measure_blocking_callers(kzg_amd, L, engine, srs, scal, n_poly, args.batch, calls=8, host_resident=True)
                res["paths"]["blocking_callers_16_host_resident_per_s"] = round(per_s, 2)   # 32 MiB over PCIe per call (pageable memory)
                res["paths"]["blocking_callers_16_host_resident_match"] = same
                kb = 256 if n_poly > 512 else 4
                per_s, same = measure_blocking_callers(kzg_amd, L, engine, srs, scal, n_poly, args.batch, calls=8, op="witness_batched", k=kb)
                res["paths"]["blocking_callers_16_witness_batched_k%d_per_s" % kb] = round(per_s, 2)   # configs[3], primary reading
                res["paths"]["blocking_callers_16_witness_batched_vs_value"] = round(per_s / value, 4)
                res["paths"]["blocking_callers_16_witness_batched_match_lone_calls"] = same
                if not args.u64:
                    res["paths"].update(measure_u64(kzg_amd, L, engine, srs, n_poly, args.batch))
                if args.log_n == 20 and not args.u64:
                    res["paths"].update(measure_spots(kzg_amd, L, engine, lambda: time.perf_counter() - t_extra < 45.0))
            except Exception as e:
                res["paths"] = {"error": str(e)}
        elif mode == "single" and args.callers:
            per_s, same = measure_blocking_callers(kzg_amd, L, engine, srs, scal, n_poly, args.batch)
            res["paths"] = {"blocking_callers_16_per_s": round(per_s, 2), "blocking_callers_16_vs_value": round(per_s / value, 4),
                            "blocking_callers_16_match_batch_results": same}
            per_s, same = measure_blocking_callers(kzg_amd, L, engine, srs, scal, n_poly, args.batch, calls=8, host_resident=True)
            res["paths"]["blocking_callers_16_host_resident_per_s"] = round(per_s, 2)
            res["paths"]["blocking_callers_16_host_resident_match"] = same
            kb = 256 if n_poly > 512 else 4
            per_s, same = measure_blocking_callers(kzg_amd, L, engine, srs, scal, n_poly, args.batch, calls=8, op="witness_batched", k=kb)
            res["paths"]["blocking_callers_16_witness_batched_k%d_per_s" % kb] = round(per_s, 2)
            res["paths"]["blocking_callers_16_witness_batched_vs_value"] = round(per_s / value, 4)
            res["paths"]["blocking_callers_16_witness_batched_match_lone_calls"] = same
        if mode == "single" and roofline and not args.no_paths and not args.no_traffic and not args.u64:
            tr, note = measure_traffic_pmc(args.log_n)
            if tr is not None:
                alg = BYTES_PER_TERM * n_local
                tr["ratio_to_algorithmic"] = round(tr["bytes_per_launch"] / alg, 2)
                res["roofline"]["traffic"] = tr["bytes_per_launch"]
                res["roofline"]["hbm"]["traffic"] = tr["bytes_per_launch"]
                res["roofline"]["traffic_measured"] = tr
                alone = res["roofline"].get("alone")
                if alone:
                    alone["hbm_real_gbs"] = round(tr["bytes_per_launch"] / (alone["avg_kernel_ms"] / 1e3) / 1e9, 1)
                    alone["hbm_real_frac"] = round(tr["bytes_per_launch"] / (alone["avg_kernel_ms"] / 1e3) / 1e9 / HBM_PEAK_GBS, 4)
            else:
                res["roofline"]["traffic_note"] = note
            if isinstance(res.get("paths"), dict) and "ntt_roofline" in res["paths"]:   # the NTT's traffic, the same way
                trn, note = measure_traffic_pmc(args.log_n, kind="ntt")
                nr = res["paths"]["ntt_roofline"]
                if trn is not None:
                    nr["traffic"] = trn["bytes_per_transform"]
                    nr["traffic_measured"] = trn
                    nr["traffic_ratio_to_algorithmic"] = round(trn["bytes_per_transform"] / (NTT_BYTES_PER_ELEM * n_local), 2)
                    nr["hbm"]["traffic"] = trn["bytes_per_transform"]
                    nr["hbm"]["real_gbs"] = round(trn["bytes_per_transform"] / (nr["kernel_ms"] / 1e3) / 1e9, 1)
                    nr["hbm"]["real_frac"] = round(trn["bytes_per_transform"] / (nr["kernel_ms"] / 1e3) / 1e9 / HBM_PEAK_GBS, 4)
                else:
                    nr["traffic"] = None
                    nr["traffic_note"] = note
        if mode == "single" and not args.no_cpu_baseline:
            if cpu is not None:
                try:
                    n_s = 1 << min(args.log_n, 20)
                    pts = params.gs.download(0, n_s)
                    sc = view(kzg_amd, scal, 0, n_s).download()
                    res["cpu_baseline"], res["cpu_baseline_all_cores"] = cpu.run(pts, sc, n_s, args.log_n,
                                                                                 gpu_side_of_cpu_legs(kzg_amd, L, engine, params, scal, n_s, args.cpu_sizes))
                except Exception as e:
                    res["cpu_baseline"] = {"value": None, "unit": "commitments/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}
            else:
                res["cpu_baseline"] = {"value": None, "unit": "commitments/s", "cores": 1, "kind": "port",
                                       "sample": "failed to start the worker pool: " + locals().get("cpu_err", "?")}
        line = json.dumps(res)
    if cpu is not None:
        cpu.close()
    main_closed = False
    if want_block and rank == 0:
        res["sharded"] = sharded_res
        res["headline"] = ("value = data-parallel replicas (the throughput answer for independent degree-2^20 commitments: a 2 GiB SRS "
                           "fits every GPU); sharded.strong / sharded.config5 = the sharded-SRS + RCCL design north_star names, "
                           "measured by the same ranks in fresh child processes before the timed region")
        line = json.dumps(res)
    # RCCL writes a version banner through C stdio, which is block-buffered on a pipe and would surface after Python's own
    # output when a process exits: every rank flushes it before the last barrier, so that rank 0's JSON line ends the output
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if job.star is not None:
        job.star.all_gather(None)
    if group is not None:
        scal.free()
        msrs.free()
        group.close()
    elif not main_closed:
        scal.free()
        engine.close()
    job.close()
    if rank == 0:
        os.write(real_stdout, (line + "\n").encode())
    os.close(real_stdout)


