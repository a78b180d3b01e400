#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X: degree-2^20 coeff-form KZG commitments/sec
(G1 Pippenger MSM, full-width uniform Fr scalars, SRS and scalars already resident in HBM).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N = 1 : workload = BASELINE configs[1] "degree-2^20 coeff_form commit (G1 Pippenger MSM) on 1xMI355X".
        One step = one batch of `--batch` independent commitments pipelined on the engine's HIP streams
        (kzg_msm_g1_batch); value = commitments / second.
N > 1 : one process per GPU.  Commitments are independent objects and a degree-2^20 SRS is 2 GiB, so the metric shards by
        commitment: every rank holds the full SRS and commits its own polynomials, NO data-path collective ("scaling": "weak";
        torch.distributed -- a gloo process group: control plane only -- carries the barriers and the max-over-ranks time).  That
        is `value`.  The same run also measures the sharded-SRS + RCCL design through the library's device group and reports it
        beside `value` as `sharded.strong` / `sharded.config5` (measure_sharded_block) -- by the same ranks, in a fresh child
        process each, BEFORE this process touches the GPU (run_sharded_block_in_children: killable, and a process that has used
        the GPU slows every other process on it).  Where one commitment
        does not fit or its latency matters, the C ABI's device group shards the commitment itself (kzg_mctx_create_rank /
        kzg_commit_coeff_sharded_batch, kzg_amd/csrc/mgpu.hip: SRS sharded contiguously, one partial point per rank and
        polynomial, ONE ncclAllGather of the 144-byte partials inside the library, local sums):
          --config5            : BASELINE configs[4] -- 2^21 terms per rank (degree 2^24 at N = 8); value = commitments/s of
                                 that N * 2^21-coefficient polynomial, msm_terms_per_sec beside it; "scaling": "weak".
          --strong             : degree-2^20 commitments/s with each commitment's 2^20 terms sharded N ways (2^20 / N per
                                 rank); "scaling": "strong" (bounded by the per-MSM sort and bucket reduction: DESIGN.md section 4).
          --weak               : 2^log_n terms per rank.

The JSON line carries `roofline` for the dominant kernel (k_accum_affine; HIP-event times measured on the engine's streams
inside this process), `paths` (the other BASELINE configs, timed after the timed region) and `cpu_baseline` (the oracle's
single-threaded C Pippenger, rank 0, N = 1 only; worker processes are started BEFORE the GPU is initialised).
The oracle is never on the measured path.
"""
import argparse
import ctypes
import json
import os
import statistics
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Hardware queues: this process is the host.  Like a Rust host following INTEGRATION.md section 6 it asks for the queues of the
# pipelined paths before its first HIP call -- kzg_amd.load() calls kzg_init_hw_queues(0) (KZG_HW_QUEUES=0 in the environment
# skips that; the engine then measures the runtime's default pool and narrows its pipeline: profiles/r03_hw_queues.txt).

LOG_N = 20
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BYTES_PER_TERM = 128           # 32 B scalar + 96 B affine point per MSM term
NTT_BYTES_PER_ELEM = 64        # one read + one write of a 32-byte Fr
# The resource that binds k_accum_affine: VALU issue, dominated by v_mad_i64_i32.  One bucket addition
# executes 6 mul30 (338 mads) + 2 sqr30 (260) + one fused double product with a single reduction (507) = 3055 mads per lane
# (static count from the ISA).  tools/microbench.hip measured the chip's 64-bit multiply-add issue rate: 31.5 T lane-op/s at
# 8 waves/SIMD, 23.7 T at the 2 waves/SIMD a 200-VGPR kernel holds (profiles/r01_microbench.txt).
MADS_PER_ADD = 6 * 338 + 2 * 260 + 507
MAD_PEAK_TLANE_S = 31.51       # reference value (round-1 microbench on another box); the line's `peak` is measured in this run
MAD_PEAK_OCC2_TLANE_S = 23.70
MAD_NOMINAL_TLANE_S = 256 * 4 * 64 / 4 * 2.4e9 / 1e12   # 39.3: 256 CUs x 4 SIMDs x 64 lanes, one wave-instruction per 4 cycles, 2.4 GHz nominal
FR_MUL_PEAK_G_S = 111.0        # measured Fr (9 x 29-bit) multiplies per second of the NTT's multiply
MADS_PER_FR29_MUL = 162        # 9 x 9 products + 9 x 9 reduction products of one Fr29 Montgomery multiply (fr29.h): the NOMINAL price of a
                               # butterfly multiplication, kept so that mad_frac stays comparable with earlier rounds
MADS_PER_SHOUP_MUL = 143       # what the stage twiddles cost since round 4: 53 (quotient columns) + 45 + 45 multiply-adds (fr29.h)
TAU = 0x5EED5EED5EED5EED       # known secret for the synthetic SRS (setup(s, n), src/lib.rs:38)
SEED = 1


def g1_adds_per_msm(n, c, W):
    """algorithmic G1 additions, n*W bucket accumulations + bucket reduction (W: digits per scalar; positional
    tables, c = 18: the measured average number of NAF digits, 2^16 buckets)."""
    return n * W + 2 * (1 << ((17 if c == 18 else c) - 1))


def naf18_avg_digits(blob):
    """Average number of width-18 NAF digits (kzg_amd/csrc/naf.h) of the canonical 32-byte scalars in `blob`: what one scalar
    contributes to the sorted entry list when the SRS uses positional tables."""
    R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    tot, cnt = 0, 0
    for o in range(0, len(blob), 32):
        k = int.from_bytes(blob[o:o + 32], "little") % R
        if k >> 254:
            k = R - k
        while k:
            if k & 1:
                d = k & 0x3ffff
                k -= d - (1 << 18) if d >= (1 << 17) else d
                tot += 1
            k >>= 1
        cnt += 1
    return tot / max(cnt, 1)


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline: worker processes (spawned before HIP initialises; they never touch the GPU)
# ---------------------------------------------------------------------------------------------------------------------
def _cpu_worker_init(native_path):
    from oracle import c_oracle as C
    if native_path:
        try:
            C.use_library(native_path)
        except Exception:
            pass
    C.lib()


def _cpu_worker_msm(args):
    """One CPU Pippenger over terms [lo, hi) of the shared sample file -- the oracle library's TIMING leg (orc_msm_g1_fast: signed
    16-bit windows, batch-affine bucket accumulation, unrolled Montgomery multiplication; checked against the plain Pippenger in
    tests/test_oracle_c.py and, by the caller, against the GPU's result); returns (seconds, 96-byte result)."""
    path, n, lo, hi = args
    from oracle import c_oracle as C
    with open(path, "rb") as f:
        f.seek(96 * lo)
        pts = f.read(96 * (hi - lo))
        f.seek(96 * n + 32 * lo)
        sc = f.read(32 * (hi - lo))
    t0 = time.perf_counter()
    out = C.msm_g1_fast_raw(pts, sc, hi - lo)
    return time.perf_counter() - t0, out


def _cpu_worker_ping(_):
    return os.getpid()


class CpuBaseline:
    """Pool of oracle workers.  start() must run before anything initialises HIP (no fork of a GPU process)."""

    def __init__(self):
        self.pool = None
        self.workers = 0
        self.native = None

    def start(self):
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor
        from oracle import c_oracle as C
        C.build()
        self.native = C.build_native()
        self.native_build = C.NATIVE_BUILD
        try:
            ncpu = len(os.sched_getaffinity(0))
        except Exception:
            ncpu = os.cpu_count() or 1
        self.workers = max(1, min(ncpu, 64))
        self.pool = ProcessPoolExecutor(self.workers, mp_context=mp.get_context("spawn"), initializer=_cpu_worker_init,
                                        initargs=(self.native,))
        list(self.pool.map(_cpu_worker_ping, range(self.workers)))  # all workers up (and the library loaded) before HIP

    def run(self, pts, sc, n, log_n, gpu_result):
        """pts / sc: the first 2^min(log_n, 20) SRS points and coefficients of polynomial 0 of the timed batch."""
        fd, path = tempfile.mkstemp(prefix="kzg_cpu_sample_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        try:
            with os.fdopen(fd, "wb") as f:
                f.write(pts)
                f.write(sc)
            scale = n / float(1 << log_n)
            # (i) single core, like the reference's multi_exp: three samples of the whole MSM on three cores at once
            r1 = list(self.pool.map(_cpu_worker_msm, [(path, n, 0, n)] * 3))
            t_med = statistics.median(t for t, _ in r1)
            ok1 = all(o == gpu_result for _, o in r1)
            single = {"value": round(scale / t_med, 5), "unit": "commitments/s", "cores": 1, "kind": "port",
                      "build": (self.native_build or "native") if self.native else "gcc -O2 (portable)",
                      "algorithm": "Pippenger, signed 16-bit windows, batch-affine bucket accumulation (one inversion per 1024 additions), "
                                   "64-bit no-carry CIOS Montgomery multiplication in C (oracle/kzg_oracle.c: orc_msm_g1_fast); no precomputed tables",
                      "samples_s": [round(t, 2) for t, _ in r1],
                      "sample": f"one whole 2^{n.bit_length() - 1}-term MSM = polynomial 0 of the timed batch, same SRS; median of 3 "
                                f"single-threaded runs ({t_med:.2f} s, {n / t_med:.0f} terms/s); matches GPU result: {ok1}"}
            # (ii) all cores.  The box may grant far fewer cores than os.cpu_count() reports (cgroup quota), so the usable
            # parallelism is measured first: every worker runs a 2^15-term slice, once alone and once all together.
            w = self.workers
            cal_n = min(n, 1 << 15)
            t_alone = list(self.pool.map(_cpu_worker_msm, [(path, n, 0, cal_n)]))[0][0]
            t0 = time.perf_counter()
            list(self.pool.map(_cpu_worker_msm, [(path, n, 0, cal_n)] * w))
            p_eff = max(1.0, min(float(w), w * t_alone / (time.perf_counter() - t0)))
            use = max(1, min(w, int(p_eff + 0.999)))
            # one commitment per core at a time is how a host would use a single-threaded multi_exp: `use` whole MSMs at once
            t0 = time.perf_counter()
            r2 = list(self.pool.map(_cpu_worker_msm, [(path, n, 0, n)] * use))
            wall = time.perf_counter() - t0
            ok2 = all(o == gpu_result for _, o in r2)
            allc = {"value": round(use * scale / wall, 4), "unit": "commitments/s", "cores": use, "kind": "port",
                    "sample": f"{use} concurrent whole 2^{n.bit_length() - 1}-term MSMs, one per worker process; os.cpu_count() = "
                              f"{os.cpu_count()}, usable parallelism measured with {w} workers on 2^15-term slices: {p_eff:.1f} cores; "
                              f"wall {wall:.2f} s, slowest worker {max(t for t, _ in r2):.2f} s; all match the GPU result: {ok2}"}
            return single, allc
        finally:
            try:
                os.unlink(path)
            except OSError:
                pass

    def close(self):
        if self.pool:
            self.pool.shutdown(wait=False, cancel_futures=True)
            self.pool = None


# ---------------------------------------------------------------------------------------------------------------------
def view(kzg_amd, buf, first, n):
    v = kzg_amd.DeviceBuffer.__new__(kzg_amd.DeviceBuffer)
    v.engine, v.n, v.sfmt, v.ptr = buf.engine, n, buf.sfmt, ctypes.c_void_p(buf.ptr.value + 32 * first)
    return v


def timeit(f, reps=3, warm=1):
    for _ in range(warm):
        f()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    return (time.perf_counter() - t0) / reps * 1e3


def measure_blocking_callers(kzg_amd, L, engine, srs, scal, n, n_polys, threads=16, calls=12, host_resident=False, op="commit", k=256):
    """The reference's call shape: `threads` host threads, each looping a BLOCKING prover call on ONE context and one resident SRS
    (thread t works on polynomial t of the timed batch).  op = "commit": kzg_commit_coeff (KZGProver::commit,
    src/coeff_form.rs:59-64); op = "witness_batched": kzg_witness_coeff_batched with k opening points (create_witness_batched,
    src/coeff_form.rs:83-111 -- BASELINE configs[3], primary reading).  Coefficients device-resident, or -- host_resident -- in
    the caller's pageable host memory as a Rust `Polynomial` would be (every call then carries its 32 MiB over PCIe).  Returns
    calls per second over all threads, and whether every result matched the same call made alone beforehand."""
    import threading
    lib, ctx = engine.lib, engine.ctx
    R = kzg_amd.api.R_MODULUS
    want, pts = {}, {}
    ref = ctypes.create_string_buffer(96)
    rbuf0, rlen0 = ctypes.create_string_buffer(32 * max(k, 2)), ctypes.c_size_t()
    for t in range(min(threads, n_polys)):
        v = view(kzg_amd, scal, t * n, n)
        if op == "commit":
            assert lib.kzg_commit_coeff(ctx, srs.handle, v.ptr, n, v.sfmt, L.IN_DEVICE, ref, L.G1_AFFINE_MONT) == 0, engine.last_error()
            want[t] = ref.raw
        else:
            xs = [kzg_amd.splitmix_scalar(700 + t, i) for i in range(k)]
            ys = [engine.poly_eval(v, x) for x in xs]
            pts[t] = (kzg_amd.pack_scalars(xs), kzg_amd.pack_scalars(ys))
            rc = lib.kzg_witness_coeff_batched(ctx, srs.handle, v.ptr, n, pts[t][0], pts[t][1], k, v.sfmt, L.IN_DEVICE, ref, L.G1_AFFINE_MONT,
                                               rbuf0, ctypes.byref(rlen0))
            assert rc == 0, engine.last_error()
            want[t] = ref.raw + rbuf0.raw[:32 * rlen0.value]
    host = {}
    if host_resident:
        for t in range(threads):
            host[t] = ctypes.create_string_buffer(view(kzg_amd, scal, (t % n_polys) * n, n).download(), 32 * n)
    ok = [True] * threads
    start = threading.Barrier(threads + 1)

    def work(t):
        v = view(kzg_amd, scal, (t % n_polys) * n, n)
        src, flags = (host[t], 0) if host_resident else (v.ptr, L.IN_DEVICE)
        out = ctypes.create_string_buffer(96)
        rbuf, rlen = ctypes.create_string_buffer(32 * max(k, 2)), ctypes.c_size_t()
        start.wait()
        for _ in range(calls):
            if op == "commit":
                rc = lib.kzg_commit_coeff(ctx, srs.handle, src, n, v.sfmt, flags, out, L.G1_AFFINE_MONT)
                got = out.raw
            else:
                xb, yb = pts[t % n_polys]
                rc = lib.kzg_witness_coeff_batched(ctx, srs.handle, src, n, xb, yb, k, v.sfmt, flags, out, L.G1_AFFINE_MONT, rbuf,
                                                   ctypes.byref(rlen))
                got = out.raw + rbuf.raw[:32 * rlen.value]
            if rc != 0 or got != want[t % n_polys]:
                ok[t] = False

    th = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    for x in th:
        x.start()
    # one untimed round first (lanes, arenas and the queue plan come into being), then the timed one
    start.wait()
    for x in th:
        x.join()
    start = threading.Barrier(threads + 1)
    th = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    for x in th:
        x.start()
    start.wait()
    t0 = time.perf_counter()
    for x in th:
        x.join()
    dt = time.perf_counter() - t0
    return threads * calls / dt, all(ok)


def measure_paths(kzg_amd, L, engine, params, scal, n, log_n, budget_s=60.0, mad_peak=MAD_PEAK_TLANE_S):
    """The other BASELINE configs at degree 2^log_n, inputs resident in HBM, each result checked by an identity that needs no
    oracle (eval-form == coeff-form, witness_eval == witness_coeff at omega^m); outside the timed region."""
    t_start = time.perf_counter()
    lib, ctx, srs = engine.lib, engine.ctx, params.gs
    R = kzg_amd.api.R_MODULUS
    res = {"log_n": log_n}
    out = ctypes.create_string_buffer(96)
    coeffs = view(kzg_amd, scal, 0, n)          # polynomial 0 of the timed batch

    def b32(v):
        return (v % R).to_bytes(32, "little")

    def commit():
        assert lib.kzg_commit_coeff(ctx, srs.handle, coeffs.ptr, n, coeffs.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT) == 0, engine.last_error()
    res["commit_coeff_ms"] = round(timeit(commit), 3)
    commitment = out.raw
    host_coeffs = coeffs.download()
    # Every reading below is CHECKED against the oracle (never on the measured path: it runs after each timing, on downloaded
    # data): p(tau) by its Horner loop over the downloaded coefficients, the expected point by its scalar multiplication of G.
    from oracle import c_oracle as C
    G = C.g1_generator()
    ptau = C.poly_eval_bytes(host_coeffs, n, TAU)
    checks = {}
    res["checked_against_oracle"] = checks
    checks["commit_coeff"] = bool(commitment == C.g1_mul(G, ptau))

    def commit_host():
        assert lib.kzg_commit_coeff(ctx, srs.handle, host_coeffs, n, coeffs.sfmt, 0, out, L.G1_AFFINE_MONT) == 0, engine.last_error()
    res["commit_host_resident_ms"] = round(timeit(commit_host), 3)       # + one 32 n-byte PCIe copy; never `value`
    assert out.raw == commitment
    # config 3: NTT then Lagrange-SRS MSM
    lag = kzg_amd.setup_lagrange(engine, TAU, n)
    ev = engine.alloc_scalars(n)
    ev.upload(host_coeffs)
    def ntt():
        assert lib.kzg_ntt_fr(ctx, ev.ptr, log_n, 0, L.IN_DEVICE) == 0, engine.last_error()
    reps = 5
    ntt_ms = timeit(ntt, reps=20, warm=2)       # wall time of the blocking call, profiling off
    engine.prof_enable(True)                    # kernel times: HIP events on the engine's stream (their recording costs wall time)
    engine.prof_reset()
    timeit(ntt, reps=reps, warm=1)
    prof = engine.prof_all()
    engine.prof_enable(False)
    kern_ms = sum(v[1] for k, v in prof.items() if k.startswith("k_ntt")) / (reps + 1)
    res["ntt_2e%d_ms" % log_n] = round(ntt_ms, 4)
    nbytes = NTT_BYTES_PER_ELEM * n
    fr_muls = (n // 2) * log_n
    res["ntt_roofline"] = {
        "bound": "valu", "kernels": {k: round(v[1] / (reps + 1), 4) for k, v in sorted(prof.items()) if k.startswith("k_ntt")},
        "kernel_ms": round(kern_ms, 4), "achieved": round(fr_muls / (kern_ms / 1e3) / 1e9, 2), "peak": FR_MUL_PEAK_G_S,
        "unit": "G Fr-mul/s ((n/2) log n butterflies)", "frac": round(fr_muls / (kern_ms / 1e3) / 1e9 / FR_MUL_PEAK_G_S, 4),
        # the same work in the unit the MSM is priced in: lane multiply-adds against the mad-issue peak measured in this run
        "mad_achieved": round(fr_muls * MADS_PER_FR29_MUL / (kern_ms / 1e3) / 1e12, 3), "mad_peak": round(mad_peak, 2),
        "mad_unit": "T lane-mad/s (%d per Fr29 multiply)" % MADS_PER_FR29_MUL,
        "mad_frac": round(fr_muls * MADS_PER_FR29_MUL / (kern_ms / 1e3) / 1e12 / mad_peak, 4),
        # the multiply-adds the kernels really execute: twiddle products are Shoup products (143), the one inter-pass product per
        # element (n <= 2^21) a Montgomery product (163)
        "mad_executed": round((fr_muls - n / 2) * MADS_PER_SHOUP_MUL / (kern_ms / 1e3) / 1e12 + n * 163 / (kern_ms / 1e3) / 1e12, 3),
        "mad_frac_executed": round(((fr_muls - n / 2) * MADS_PER_SHOUP_MUL + n * 163) / (kern_ms / 1e3) / 1e12 / mad_peak, 4),
        "hbm": {"bound": "hbm", "achieved": round(nbytes / (kern_ms / 1e3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(nbytes / (kern_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 5), "algorithmic_bytes": nbytes}}
    ev.upload(host_coeffs)
    ntt()

    def commit_eval():
        assert lib.kzg_commit_eval(ctx, lag.handle, ev.ptr, n, ev.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT) == 0, engine.last_error()
    res["commit_eval_ms"] = round(timeit(commit_eval), 3)
    res["commit_eval_equals_commit_coeff"] = bool(out.raw == commitment)
    # configs[2]: the evaluations came from the GPU's NTT of the coefficients; the Lagrange-SRS MSM of them must be [p(tau)]G
    checks["commit_eval"] = bool(out.raw == C.g1_mul(G, ptau))
    _, _, omega_n = kzg_amd.compute_omega(n)
    ev_head = ev.download(2, offset=n - 2)       # ... and two of the NTT's outputs against direct Horner evaluation by the oracle
    checks["ntt_outputs_sampled"] = all(int.from_bytes(ev_head[32 * i:32 * i + 32], "little") == C.poly_eval_bytes(host_coeffs, n, pow(omega_n, n - 2 + i, R))
                                        for i in range(2))
    # config 4, single opening and batched k = 256
    x = kzg_amd.splitmix_scalar(99, 0)
    y = engine.poly_eval(coeffs, x)

    def witness():
        rc = lib.kzg_witness_coeff(ctx, srs.handle, coeffs.ptr, n, b32(x), b32(y), coeffs.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
        assert rc == 0, engine.last_error()
    res["witness_coeff_ms"] = round(timeit(witness), 3)
    checks["witness_coeff"] = bool(y == C.poly_eval_bytes(host_coeffs, n, x) and out.raw == C.g1_mul(G, (ptau - y) * pow(TAU - x, -1, R) % R))
    m = 12345 % n
    xm = pow(kzg_amd.compute_omega(n)[2], m, R)
    ym = engine.poly_eval(coeffs, xm)
    rc = lib.kzg_witness_coeff(ctx, srs.handle, coeffs.ptr, n, b32(xm), b32(ym), coeffs.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
    w_coeff = out.raw

    def witness_eval():
        assert lib.kzg_witness_eval(ctx, lag.handle, ev.ptr, n, m, ev.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT) == 0, engine.last_error()
    res["witness_eval_ms"] = round(timeit(witness_eval), 3)
    res["witness_eval_equals_witness_coeff"] = bool(rc == 0 and out.raw == w_coeff)
    checks["witness_eval"] = bool(out.raw == C.g1_mul(G, (ptau - C.poly_eval_bytes(host_coeffs, n, xm)) * pow(TAU - xm, -1, R) % R))
    k = 256 if n > 512 else 4
    xs = [kzg_amd.splitmix_scalar(7, i) for i in range(k)]
    ys = [engine.poly_eval(coeffs, v) for v in xs]
    xb, yb = kzg_amd.pack_scalars(xs), kzg_amd.pack_scalars(ys)
    rbuf, rlen = ctypes.create_string_buffer(32 * k), ctypes.c_size_t()

    def batched():
        rc = lib.kzg_witness_coeff_batched(ctx, srs.handle, coeffs.ptr, n, xb, yb, k, coeffs.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT,
                                           rbuf, ctypes.byref(rlen))
        assert rc == 0, engine.last_error()
    res["witness_batched_k%d_ms" % k] = round(timeit(batched, reps=2), 3)
    # configs[3]: w == [(p(tau) - I(tau)) / Z(tau)]G with I = the returned interpolant (it must pass through the k points, two of
    # whose values the oracle recomputes from the coefficients) and Z = prod (tau - x_i): all oracle / integer arithmetic
    Icoef = kzg_amd.unpack_scalars(rbuf.raw[:32 * rlen.value])
    Ztau = 1
    for v in xs:
        Ztau = Ztau * (TAU - v) % R
    checks["witness_batched_k%d" % k] = bool(
        rlen.value == (k if k > 1 else 2) and all(C.poly_eval(Icoef, xs[i]) == ys[i] for i in range(0, k, max(1, k // 8)))
        and all(C.poly_eval_bytes(host_coeffs, n, xs[i]) == ys[i] for i in (0, k - 1))
        and out.raw == C.g1_mul(G, (ptau - C.poly_eval(Icoef, TAU)) * pow(Ztau, -1, R) % R))
    if time.perf_counter() - t_start < budget_s * 0.5:
        outs = ctypes.create_string_buffer(96 * k)
        st = (ctypes.c_int * k)()

        def witness_many():
            rc = lib.kzg_witness_coeff_many(ctx, srs.handle, coeffs.ptr, n, xb, yb, k, coeffs.sfmt, L.IN_DEVICE, outs, L.G1_AFFINE_MONT, st)
            assert rc == 0, engine.last_error()
        t_many = timeit(witness_many, reps=1, warm=1)
        res["witness_many_k%d_per_s" % k] = round(k / t_many * 1e3, 1)
        res["witness_many_all_on_poly"] = all(v == 0 for v in st)
    # config 3 at pipeline speed: 16 host threads, each taking coefficient vectors to evaluation form (EvaluationDomain::fft, in
    # place on its own device buffer) and committing them against the Lagrange-basis SRS (KZGProverEvalForm::commit) -- every
    # commitment must equal the coefficient-form commitment of the same polynomial
    if time.perf_counter() - t_start < budget_s * 0.7:
        try:
            import threading
            threads, calls = 16, 4
            bufs = [[engine.alloc_scalars(n) for _ in range(calls + 1)] for _ in range(threads)]
            for t in range(threads):
                for b in bufs[t]:
                    b.upload(host_coeffs)
            ok = [True] * threads

            def work(t, which, barrier):
                o = ctypes.create_string_buffer(96)
                barrier.wait()
                for b in which(bufs[t]):
                    rc = lib.kzg_ntt_fr(ctx, b.ptr, log_n, 0, L.IN_DEVICE)
                    rc = rc or lib.kzg_commit_eval(ctx, lag.handle, b.ptr, n, b.sfmt, L.IN_DEVICE, o, L.G1_AFFINE_MONT)
                    if rc != 0 or o.raw != commitment:
                        ok[t] = False

            def round_(which):
                bar = threading.Barrier(threads + 1)
                th = [threading.Thread(target=work, args=(t, which, bar)) for t in range(threads)]
                for x in th:
                    x.start()
                bar.wait()
                t0 = time.perf_counter()
                for x in th:
                    x.join()
                return time.perf_counter() - t0
            round_(lambda bs: bs[:1])          # untimed: lanes, plans, arenas
            dt = round_(lambda bs: bs[1:])
            res["blocking_callers_16_fft_commit_eval_per_s"] = round(threads * calls / dt, 2)   # configs[2] from many threads
            res["blocking_callers_16_fft_commit_eval_match_commit_coeff"] = all(ok)
            for bl in bufs:
                for b in bl:
                    b.free()
        except Exception as e:  # noqa: BLE001
            res["blocking_callers_16_fft_commit_eval_note"] = str(e)[:200]
    ev.free()
    lag.free()
    return res


def measure_spots(kzg_amd, L, engine, budget_ok):
    """2^16 and 2^24 spot values of the same metric (sweep 2^16 - 2^24), full-width scalars."""
    res = {}
    for log_m, batch in ((16, 64), (24, 2)):
        if not budget_ok():
            break
        m = 1 << log_m
        p = kzg_amd.setup(engine, TAU, m, g2_len=0)
        sc = engine.alloc_scalars(m * batch).fill_random(SEED + 77)
        out = ctypes.create_string_buffer(96 * batch)

        def step():
            rc = engine.lib.kzg_msm_g1_batch(engine.ctx, p.gs.handle, 0, sc.ptr, m, batch, sc.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
            assert rc == 0, engine.last_error()
        ms = timeit(step, reps=3 if log_m == 16 else 2)
        c, W = p.gs.window_info()
        res["commit_2e%d" % log_m] = {"commitments_per_s": round(batch / ms * 1e3, 2), "batch": batch, "window_bits": c, "windows": W,
                                      "hbm_frac": round(BYTES_PER_TERM * m * batch / (ms / 1e3) / 1e9 / HBM_PEAK_GBS, 5)}
        one = ctypes.create_string_buffer(96)

        def single():
            rc = engine.lib.kzg_msm_g1(engine.ctx, p.gs.handle, 0, sc.ptr, m, sc.sfmt, L.IN_DEVICE, one, L.G1_AFFINE_MONT)
            assert rc == 0, engine.last_error()
        res["commit_2e%d" % log_m]["single_commit_latency_ms"] = round(timeit(single, reps=2), 3)
        # checked: commitments of the last batch step against [p(tau)]G by the oracle (2^24: the last one -- half a GiB of
        # coefficients through its Horner loop; BASELINE configs[4]'s polynomial size on one GPU)
        try:
            from oracle import c_oracle as C
            G = C.g1_generator()
            which = [batch - 1] if log_m == 24 else sorted({0, batch // 2, batch - 1})
            res["commit_2e%d" % log_m]["checked_against_oracle"] = all(
                out.raw[96 * b:96 * b + 96] == C.g1_mul(G, C.poly_eval_bytes(view_of(sc, b * m, m).download(), m, TAU)) for b in which)
            res["commit_2e%d" % log_m]["checked_commitments"] = which
        except Exception as e:  # noqa: BLE001
            res["commit_2e%d" % log_m]["checked_against_oracle"] = "check failed to run: %s" % e
        sc.free()
        p.gs.free()
    return res


def view_of(buf, first, n):
    import kzg_amd
    return view(kzg_amd, buf, first, n)


def measure_u64(kzg_amd, L, engine, srs, n, batch, steps=3):
    """The reference benches' own distribution (benches/commit_coeff_form.rs:16-21: coefficients are u64 values): the same batched
    commit on u64-valued scalars resident in HBM -- SURVEY 8(d)'s secondary reading of the headline metric.  Three commitments of
    the last step are checked against [p(tau)]G by the oracle."""
    sc = engine.alloc_scalars(n * batch)
    for b in range(batch):
        view(kzg_amd, sc, b * n, n).fill_random(SEED + 31000 + 1000 * b, u64_valued=True)
    out = ctypes.create_string_buffer(96 * batch)

    def step():
        rc = engine.lib.kzg_msm_g1_batch(engine.ctx, srs.handle, 0, sc.ptr, n, batch, sc.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
        assert rc == 0, engine.last_error()
    ms = timeit(step, reps=steps, warm=1)
    res = {"commit_u64_per_s": round(batch / ms * 1e3, 2), "commit_u64_batch": batch,
           "commit_u64_scalars": "u64-valued Fr (benches/commit_coeff_form.rs:16-21), 4 non-zero 16/17-bit windows per scalar"}
    try:
        from oracle import c_oracle as C
        G = C.g1_generator()
        which = sorted({0, batch // 2, batch - 1})
        res["commit_u64_checked_against_oracle"] = all(
            out.raw[96 * b:96 * b + 96] == C.g1_mul(G, C.poly_eval_bytes(view(kzg_amd, sc, b * n, n).download(), n, TAU)) for b in which)
    except Exception as e:  # noqa: BLE001
        res["commit_u64_checked_against_oracle"] = "check failed to run: %s" % e
    sc.free()
    return res


def pmc_child(log_n):
    """The workload a `rocprofv3 --kernel-trace --pmc <counter> -- python3 bench.py --pmc-child <log_n>` pass of THIS script
    profiles (measure_traffic_pmc below starts it as a child process): a few lone degree-2^log_n commitments on uniform scalars
    resident in HBM -- k_accum_affine launches of exactly the shape the timed region runs."""
    import kzg_amd
    from kzg_amd import _lib as L
    e = kzg_amd.Engine(0)
    n = 1 << log_n
    params = kzg_amd.setup(e, TAU, n, g2_len=0)
    sc = e.alloc_scalars(n).fill_random(SEED)
    out = ctypes.create_string_buffer(96)
    for _ in range(4):
        assert e.lib.kzg_msm_g1(e.ctx, params.gs.handle, 0, sc.ptr, n, sc.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT) == 0, e.last_error()
    sc.free()
    params.gs.free()
    e.close()


def measure_traffic_pmc(log_n, timeout_s=150):
    """HBM bytes per k_accum_affine launch from the PMC counters, collected the way MI355X_MICROARCH.md prescribes: two separate
    rocprofv3 passes (--kernel-trace --pmc FETCH_SIZE, then WRITE_SIZE: they do not fit one pass on gfx950) of a child process
    running pmc_child, FETCH_SIZE doubled (gfx950 tallies the 128-byte requests of wide loads at 64 bytes), units of KB.  Returns
    (dict or None, note)."""
    import csv
    import glob
    import shutil
    import subprocess
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not rp:
        return None, "rocprofv3 not found"
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="kzg_pmc_", dir="/tmp")
        try:
            cmd = [rp, "--kernel-trace", "--pmc", ctr, "-d", d, "-o", "p", "--output-format", "csv", "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--pmc-child", str(log_n)]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout_s)
            tot, cnt = 0.0, 0
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "k_accum_affine" in row.get("Kernel_Name", "") and row.get("Counter_Name") == ctr:
                        tot += float(row["Counter_Value"])
                        cnt += 1
            if not cnt:
                return None, "%s pass produced no k_accum_affine rows (rc %d): %s" % (ctr, r.returncode, (r.stderr or "")[-200:])
            vals[ctr] = (tot / cnt, cnt)
        except Exception as e:  # noqa: BLE001
            return None, "%s pass failed: %s" % (ctr, e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    f_kb, w_kb = vals["FETCH_SIZE"][0], vals["WRITE_SIZE"][0]
    return {"bytes_per_launch": int(round((2 * f_kb + w_kb) * 1024)), "raw_fetch_kb": round(f_kb, 1), "raw_write_kb": round(w_kb, 1),
            "launches_sampled": vals["FETCH_SIZE"][1],
            "method": "two rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE, WRITE_SIZE) of a child process running lone 2^%d commitments, "
                      "collected during this bench run; bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB (gfx950 correction of "
                      "MI355X_MICROARCH.md)" % log_n}, None


class Job:
    """The ranks of one bench run.  World 1 needs no torch at all ("PyTorch is not needed"): barrier = the
    engine's own device synchronisation.  World > 1: torch.distributed carries the barriers, the max-over-ranks time and the small
    host objects of the checks; the data-path collective is inside the library (kzg_mctx, RCCL)."""

    def __init__(self, rank, local_rank, world, use_torch, backend="gloo"):
        self.rank, self.local_rank, self.world = rank, local_rank, world
        self.torch = self.dist = None
        self.red_dev = "cpu"
        self.engines = []
        if not use_torch:
            return
        import torch
        self.torch = torch
        # order matters: torch first (its wheel bundles a HIP runtime under the system's SONAME and must bring it in), then the
        # library (binds to that runtime and asks for the hardware queues: kzg_init_hw_queues sets GPU_MAX_HW_QUEUES), and only then
        # the first HIP call of the process (set_device below), which is when the runtime sizes its queue pool
        import kzg_amd
        kzg_amd.load()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist
        self.dist = dist
        # The process group is the CONTROL plane only (barriers, the max-over-ranks time, the small host objects of the checks): gloo
        # on CPU tensors.  The data-path collective is the library's own RCCL communicator (kzg_mctx), and a second, idle NCCL
        # communicator of torch's on the same GPU costs it dearly: with torch's nccl process group alive the device-group path
        # measured 361 commitments/s against 454 without (one GPU, RCCL all-gather forced on; profiles/r04_torch_pg_interference.txt),
        # while the plain path is unaffected.  --torch-backend nccl restores the old control plane.
        shared = bool(os.environ.get("KZG_BENCH_SHARED_GPU"))   # test mode for a one-GPU box: every rank on device 0
        if shared:
            self.local_rank = 0
        torch.cuda.set_device(self.local_rank)
        if backend == "nccl" and not shared:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", self.local_rank), rank=rank, world_size=world)
            self.red_dev = "cuda"   # where the timing / agreement reductions live
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
            self.torch.cuda.synchronize()
        for e in self.engines:      # the engine's own streams (non-blocking streams: torch's device sync covers them too)
            e.sync()

    def max_over_ranks(self, x):
        if self.dist is None or self.world == 1:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.red_dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_agree(self, ok):
        if self.dist is None or self.world == 1:
            return bool(ok)
        t = self.torch.tensor([1 if ok else 0], dtype=self.torch.int32, device=self.red_dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def gather_objects(self, obj):
        if self.dist is None or self.world == 1:
            return [obj]
        allv = [None] * self.world
        self.dist.all_gather_object(allv, obj)
        return allv

    def broadcast_object(self, make):
        if self.dist is None or self.world == 1:
            return make()
        box = [make() if self.rank == 0 else None]
        self.dist.broadcast_object_list(box, src=0)
        return box[0]

    def runtime(self, kzg_amd, L):
        buf = ctypes.create_string_buffer(512)
        L.load().kzg_runtime_info(buf, 512)
        r = {"library": buf.value.decode(), "torch_imported": self.torch is not None}
        if self.torch is not None:
            r["torch"] = "%s (hip %s)" % (self.torch.__version__, getattr(self.torch.version, "hip", None))
        return r

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def known_tau_partials(kzg_amd, scal, n_local, lo, which):
    """tau^lo * p_slice(tau) for the polynomials `which` of a [batch][n_local] device array, by the ORACLE (coefficients
    downloaded, its Horner loop): this rank's share of p_b(tau)."""
    from oracle import c_oracle as C
    R = kzg_amd.api.R_MODULUS
    return [pow(TAU, lo, R) * C.poly_eval_bytes(view(kzg_amd, scal, b * n_local, n_local).download(), n_local, TAU) % R if n_local else 0
            for b in which]


def check_known_tau(kzg_amd, job, scal, n_local, lo, out_raw, which, spans_ranks):
    """out[b] == [p_b(tau)]G for b in `which`; p_b(tau) = sum over ranks of the slices' shares when a commitment spans the ranks.
    The right-hand side is the oracle's alone; every rank checks, all must agree."""
    from oracle import c_oracle as C
    R = kzg_amd.api.R_MODULUS
    mine = known_tau_partials(kzg_amd, scal, n_local, lo, which)
    if spans_ranks and job.world > 1:
        allv = job.gather_objects(mine)
        mine = [sum(v[i] for v in allv) % R for i in range(len(which))]
    G = C.g1_generator()
    ok = all(out_raw[96 * b: 96 * b + 96] == C.g1_mul(G, mine[i]) for i, b in enumerate(which))
    return job.all_agree(ok)


def measure_sharded_block(kzg_amd, L, job, args, force_gather):
    """What north_star names, measured in the default multi-GPU run next to the replicas: the SRS sharded over the ranks, one
    partial point per rank and polynomial, ONE RCCL all-gather of the 144-byte partials inside the library, local sums
    (kzg_commit_coeff_sharded_batch, kzg_amd/csrc/mgpu.hip).  (i) strong: every degree-2^log_n commitment sharded N ways;
    (ii) config5: BASELINE configs[4], 2^21 terms per rank (degree 2^24 at N = 8).  Every commitment of the last step of each is
    checked against [p(tau)]G by the oracle.  A group that cannot form degrades to a note."""
    from kzg_amd.api import DeviceGroup
    from kzg_amd.distributed import shard_range
    rank, world = job.rank, job.world
    res = {}
    group, err = None, None
    try:
        uid = job.broadcast_object(DeviceGroup.unique_id)
        group = DeviceGroup.for_rank(job.local_rank, rank, world, uid)
        if force_gather:
            group.set_option("always_gather", 1)
    except Exception as e:  # noqa: BLE001
        err = str(e)
    if not job.all_agree(group is not None):
        if group is not None:
            group.close()
        return {"note": "device group could not be formed (%s): sharded-SRS + RCCL modes not measured in this run" % (err or "failure on another rank")}
    try:
        res["rccl"] = group.info()
        res["rccl_ranks"] = group.world
        eng = group.engine(0)
        if args.streams:
            eng.set_option("streams", args.streams)
        job.engines.append(eng)
        batch, steps = args.sharded_batch, args.sharded_steps
        for mode, n_poly in (("strong", 1 << args.log_n), ("config5", world << 21)):
            lo, hi = shard_range(n_poly, rank, world)
            n_local = hi - lo
            scal = eng.alloc_scalars(max(n_local, 1) * batch)
            for b in range(batch):
                view(kzg_amd, scal, b * n_local, n_local).fill_random(SEED + 5000 + 1000 * b + 4 * lo)
            msrs = group.setup(TAU, n_poly)
            srs, first = msrs.shard(0)
            assert first == lo and len(srs) == n_local
            c, W = srs.window_info()
            out = ctypes.create_string_buffer(96 * batch)
            ptrs = (ctypes.c_void_p * 1)(scal.ptr.value)

            def step():
                rc = group.lib.kzg_commit_coeff_sharded_batch(group.handle, msrs.handle, ptrs, n_poly, batch, scal.sfmt, L.IN_DEVICE, out,
                                                              L.G1_AFFINE_MONT)
                if rc:
                    raise RuntimeError(group.last_error())
            step()
            if "formation" not in res:      # after the first exchange: what forming the communicator cost on this rank (ms per phase)
                res["formation"] = group.formation()
                res["rccl"] = group.info()
            job.barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            job.barrier()
            dt = job.max_over_ranks(time.perf_counter() - t0)
            ok = check_known_tau(kzg_amd, job, scal, n_local, lo, out.raw, list(range(batch)), spans_ranks=True)
            v = batch * steps / dt
            res[mode] = {"value": round(v, 3), "unit": "commitments/s", "scaling": "strong" if mode == "strong" else "weak",
                         "polynomial_coefficients": n_poly, "terms_per_rank": n_local, "batch": batch, "steps": steps,
                         "ms_per_step": round(dt / steps * 1e3, 4), "window_bits": c, "windows": W,
                         "msm_terms_per_sec": round(v * n_poly, 1), "g1_adds_per_sec": round(v * world * g1_adds_per_msm(n_local, c, W), 1),
                         "hbm_frac_algorithmic": round(BYTES_PER_TERM * n_poly * v / 1e9 / (HBM_PEAK_GBS * world), 6),
                         "collective": "one ncclAllGather of (batch + 1) x 144 B per rank and step, inside the library",
                         "all_results_match_known_tau": ok}
            scal.free()
            msrs.free()
        job.engines.remove(eng)
    except Exception as e:  # noqa: BLE001
        res["error"] = str(e)
    finally:
        group.close()
    return res


def sharded_child_main(args):
    """`bench.py --sharded-child`: the sharded block alone, in a fresh process per rank (started by run_sharded_block_in_children).
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the parent; rank 0 prints the block as one JSON line."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)       # RCCL's banner and anything else native goes to stderr: stdout carries the JSON line only
    job = Job(rank, local_rank, world, use_torch=(world > 1), backend="gloo")
    import kzg_amd
    from kzg_amd import _lib as L
    res = measure_sharded_block(kzg_amd, L, job, args, force_gather=(world == 1))
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if job.dist is not None:
        job.dist.barrier()
    job.close()
    if rank == 0:
        os.write(real_stdout, (json.dumps(res) + "\n").encode())
    os.close(real_stdout)


def _tail(path_or_text, n=1500, is_path=False):
    try:
        t = open(path_or_text, errors="replace").read() if is_path else (path_or_text or "")
    except OSError:
        return ""
    return t[-n:]


def run_sharded_block_in_children(args, rank, local_rank, world):
    """The `sharded` block in a FRESH child process per rank, run BEFORE this process touches the GPU.  Whatever goes wrong while a
    device group forms over RCCL -- a bootstrap that stalls for minutes on a hostile network stack (round 4's driver box), a crash
    inside the communicator, a dead peer -- happens in a process that can be killed; this process' line and exit status stay
    truthful, and the block says what happened: the library's per-phase formation times (KZG_DEBUG), the child's exit code, and the
    tail of RCCL's own log (NCCL_DEBUG=INFO into NCCL_DEBUG_FILE from the start).  Before, not after: a process that has used the
    GPU slows every OTHER process on it by its mere presence (its hardware queues stay mapped; measured from a parent that had run
    one batch and closed its engine: 43 instead of 460 commitments/s in the child, and hipDeviceReset does not give them back)."""
    import glob
    import signal
    import subprocess
    # the children's own rendezvous: a port every rank can derive without talking (this runs before the ranks have a process group)
    base = int(os.environ.get("MASTER_PORT", "29531"))
    port = base + 29 if base + 29 < 65536 else base - 29
    log_prefix = os.path.join(tempfile.gettempdir(), "kzg_rccl_%d_r%d" % (os.getpid(), rank))
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}   # (the agent-store flag would make the child look
    #                                                                                     for torchrun's store on the new port)
    env.update(RANK=str(rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               KZG_DEBUG="1")
    if env.get("KZG_RCCL_SINGLE_NODE_ENV", "1") != "0":     # this process is the host: the one-node RCCL knobs (kzg_amd/distributed.py)
        from kzg_amd.distributed import SINGLE_NODE_RCCL_ENV
        for k, v in SINGLE_NODE_RCCL_ENV.items():
            env.setdefault(k, v)
    if env.get("NCCL_DEBUG", "VERSION").upper() in ("VERSION", "WARN"):
        env["NCCL_DEBUG"] = "INFO"
        env.setdefault("NCCL_DEBUG_SUBSYS", "INIT,BOOTSTRAP,NET,ENV")
    env.setdefault("NCCL_DEBUG_FILE", log_prefix + ".%p.log")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--sharded-child", "--gpus", str(world), "--log-n", str(args.log_n),
           "--sharded-batch", str(args.sharded_batch), "--sharded-steps", str(args.sharded_steps), "--streams", str(args.streams)]
    t0 = time.perf_counter()
    p = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    timed_out = False
    try:
        out, err = p.communicate(timeout=args.sharded_timeout)
    except subprocess.TimeoutExpired:
        timed_out = True
        try:
            os.killpg(p.pid, signal.SIGKILL)     # exactly the process group this call started
        except OSError:
            pass
        out, err = p.communicate()
    wall = time.perf_counter() - t0
    block = None
    for ln in (out or "").splitlines():
        if ln.startswith("{"):
            try:
                block = json.loads(ln)
            except ValueError:
                pass
    child = {"rc": p.returncode, "wall_s": round(wall, 2), "timed_out": timed_out, "process": "fresh child per rank"}
    healthy = not timed_out and p.returncode == 0     # (rank 0's child has agreed every check with the other ranks' children)
    if rank != 0:
        for f in glob.glob(log_prefix + "*"):
            try:
                os.unlink(f)
            except OSError:
                pass
        return None
    if block is None:
        block = {"note": ("the sharded block did not finish within %d s and its process was killed" % args.sharded_timeout) if timed_out
                 else "the sharded block's process ended with code %s and no result" % p.returncode}
    block["child"] = child
    slow = isinstance(block.get("formation"), dict) and block["formation"].get("formation_ms", 0) > 10000
    if not healthy or slow or "error" in block or "note" in block:
        logs = sorted(glob.glob(log_prefix + "*"))
        block["diagnostics"] = {"stderr_tail": _tail(err, 2500), "rccl_log_tail": _tail(logs[0], 2500, is_path=True) if logs else "",
                                "env": {k: v for k, v in env.items() if k.startswith(("NCCL_", "RCCL_", "KZG_", "GPU_MAX"))}}
    for f in glob.glob(log_prefix + "*"):
        try:
            os.unlink(f)
        except OSError:
            pass
    return block


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="commitments per step (pipelined on the engine's HIP streams)")
    ap.add_argument("--streams", type=int, default=0,
                    help="lanes the engine pipelines a batch over (0 = the engine's default: 14 + 4 accumulation streams from the process' shared "
                         "stream pool, which leaves an RCCL communicator its hardware queues)")
    ap.add_argument("--accum-blocks", type=int, default=0, help="engine option accum_blocks_batch (0 = default)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="extra engine option (kzg_ctx_set_option), repeatable")
    ap.add_argument("--log-n", type=int, default=LOG_N)
    ap.add_argument("--window-bits", type=int, default=0, help="engine option window_bits (0 = engine default)")
    ap.add_argument("--u64", action="store_true", help="u64-valued coefficients (the reference benches' distribution)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    ap.add_argument("--pmc-child", type=int, default=0, help=argparse.SUPPRESS)   # internal: the workload of measure_traffic_pmc's rocprofv3 passes
    ap.add_argument("--torch-backend", default="gloo", choices=["gloo", "nccl"],
                    help="N>1: backend of the torch process group that carries barriers and timing reductions (the data-path collective is the "
                         "library's own RCCL communicator either way)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc passes that fill roofline.traffic")
    ap.add_argument("--sharded-timeout", type=int, default=240, help="seconds the sharded block's child process may take before it is killed and the line printed without it")
    ap.add_argument("--sharded-child", action="store_true", help=argparse.SUPPRESS)   # internal: the block alone (run_sharded_block_in_children)
    args = ap.parse_args()
    if args.pmc_child:
        pmc_child(args.pmc_child)
        return
    if args.sharded_child:
        sharded_child_main(args)
        return

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
            cpu.start()          # before torch / HIP: the workers are spawned from a process that has not touched the GPU
        except Exception as e:   # the baseline must never take the bench line down
            cpu = None
            cpu_err = str(e)

    # torch only where there is more than one rank (process group) or the torch-carried group hand-off is the thing under test
    # (--sharded).  At N = 1 the library is the only thing that loads a HIP runtime: the system's.
    job = Job(rank, local_rank, world, use_torch=(world > 1 or sharded), backend=args.torch_backend)
    local_rank = job.local_rank
    dist = job.dist
    import kzg_amd
    from kzg_amd import _lib as L

    # ---- the device group first (it decides the mode: if the group cannot be formed on this node the run degrades to
    # data-parallel replicas and says so, instead of producing no number at all)
    group, group_note = None, None
    if sharded:
        from kzg_amd.distributed import group_from_torch, shard_range
        ok = 1
        try:
            group = group_from_torch(dist, local_rank, rank, world)
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
        checked_ok = check_known_tau(kzg_amd, job, scal, n_local, lo, out.raw, which, spans_ranks=sharded)
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
            latency_ms = timeit(single, reps=4, warm=1)
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
        res["hip_runtime"] = job.runtime(kzg_amd, L)
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
        if mode == "single" and not args.no_cpu_baseline:
            if cpu is not None:
                try:
                    n_s = 1 << min(args.log_n, 20)
                    pts = params.gs.download(0, n_s)
                    sc = view(kzg_amd, scal, 0, n_s).download()
                    gpu_res = engine.msm(params.gs, view(kzg_amd, scal, 0, n_s), n=n_s)
                    res["cpu_baseline"], res["cpu_baseline_all_cores"] = cpu.run(pts, sc, n_s, args.log_n, gpu_res)
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
    if dist is not None:
        dist.barrier()
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


if __name__ == "__main__":
    main()
