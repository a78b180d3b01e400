"""cpu_baseline: the oracle (oracle/kzg_oracle.c, a C restatement of the reference's CPU path -- "kind": "port") timed on the host
cores by worker processes that are started BEFORE HIP initialises and never touch the GPU.  One of the two bench modules that may
import oracle/ (the other: checks).  What is timed mirrors the reference's own benches:
  msm_2e10 / msm_2e16 / msm_2e20   KZGProver::commit           benches/commit_coeff_form.rs:24-39   (orc_msm_g1_fast)
  ntt_2e20                         EvaluationDomain::fft       benches/fft.rs:20-34                 (orc_fft: serial_fft, src/ft.rs:291-333)
  witness_2e20                     KZGProver::create_witness   benches/create_witness_coeff_form.rs:28-31 (orc_witness_quotient + orc_msm_g1_fast)
Every leg's output is compared with the bytes the GPU produced for the same input."""
import hashlib
import os
import statistics
import tempfile
import time


def _cpu_worker_init(native_path):
    from oracle import c_oracle as C
    if native_path:
        try:
            C.use_library(native_path)
        except Exception:
            pass
    C.lib()


def _read_sample(path, n, lo, hi):
    with open(path, "rb") as f:
        f.seek(96 * lo)
        pts = f.read(96 * (hi - lo))
        f.seek(96 * n + 32 * lo)
        sc = f.read(32 * (hi - lo))
    return pts, sc


def _cpu_worker_msm(args):
    """One CPU Pippenger over terms [lo, hi) of the shared sample file -- the oracle library's TIMING leg (orc_msm_g1_fast: signed
    16-bit windows, batch-affine bucket accumulation, unrolled Montgomery multiplication; checked against the plain Pippenger in
    tests/test_oracle_c.py and, by the caller, against the GPU's result); returns (seconds, 96-byte result)."""
    path, n, lo, hi = args[:4]
    plain = len(args) > 4 and args[4] == "plain"    # the textbook Pippenger (orc_msm_g1: window by size): faster below ~2^15 terms
    from oracle import c_oracle as C
    pts, sc = _read_sample(path, n, lo, hi)
    t0 = time.perf_counter()
    out = C.msm_g1_raw(pts, sc, hi - lo) if plain else C.msm_g1_fast_raw(pts, sc, hi - lo)
    return time.perf_counter() - t0, out


def _cpu_worker_ntt(args):
    """EvaluationDomain::fft of the first 2^log_m coefficients of the sample (orc_fft = serial_fft, src/ft.rs:291-333; the reference
    without --features parallel runs exactly this on one core); returns (seconds, sha256 of the 32-byte little-endian outputs)."""
    path, n, log_m = args
    from oracle import c_oracle as C
    _, sc = _read_sample(path, n, 0, 1 << log_m)
    t0 = time.perf_counter()
    out = C.fft_bytes(sc, log_m)
    dt = time.perf_counter() - t0
    return dt, hashlib.sha256(out).hexdigest()


def _cpu_worker_witness(args):
    """KZGProver::create_witness (src/coeff_form.rs:66-81): long division of p - y by X - x (orc_witness_quotient), then the MSM of the
    n - 1 quotient coefficients; returns (seconds, 96-byte witness, remainder-is-nonzero flag)."""
    path, n, x, y = args
    from oracle import c_oracle as C
    pts, sc = _read_sample(path, n, 0, n)
    t0 = time.perf_counter()
    q, nz = C.witness_quotient_bytes(sc, n, x, y)
    out = C.msm_g1_fast_raw(pts[:96 * (n - 1)], q, n - 1)
    return time.perf_counter() - t0, out, nz


def _cpu_worker_ping(_):
    return os.getpid()


MSM_ALGORITHM = ("Pippenger, signed 16-bit windows, batch-affine bucket accumulation (one inversion per 1024 additions), 64-bit no-carry "
                 "CIOS Montgomery multiplication in C (oracle/kzg_oracle.c: orc_msm_g1_fast); no precomputed tables")


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

    def run(self, pts, sc, n, log_n, gpu):
        """pts / sc: the first n = 2^min(log_n, 20) SRS points and coefficients of polynomial 0 of the timed batch.  gpu: what the
        GPU produced for the same inputs -- {"msm": {log_m: 96 bytes}, "ntt_sha256": hex or None, "witness": (x, y, 96 bytes) or None}.
        Returns (cpu_baseline, cpu_baseline_all_cores)."""
        fd, path = tempfile.mkstemp(prefix="kzg_cpu_sample_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        build = (self.native_build or "native") if self.native else "gcc -O2 (portable)"
        log_top = n.bit_length() - 1
        try:
            with os.fdopen(fd, "wb") as f:
                f.write(pts)
                f.write(sc)
            scale = n / float(1 << log_n)
            # (i) single core, like the reference's multi_exp: three samples of the whole MSM on three cores at once; the NTT and
            # create_witness legs (one core each) run beside them
            f_msm = [self.pool.submit(_cpu_worker_msm, (path, n, 0, n)) for _ in range(3)]
            f_ntt = [self.pool.submit(_cpu_worker_ntt, (path, n, log_top)) for _ in range(3)] if gpu.get("ntt_sha256") else []
            f_wit = self.pool.submit(_cpu_worker_witness, (path, n) + tuple(gpu["witness"][:2])) if gpu.get("witness") else None
            r1 = [f.result() for f in f_msm]
            t_med = statistics.median(t for t, _ in r1)
            ok1 = all(o == gpu["msm"][log_top] for _, o in r1)
            single = {"value": round(scale / t_med, 5), "unit": "commitments/s", "cores": 1, "kind": "port", "build": build,
                      "algorithm": MSM_ALGORITHM, "samples_s": [round(t, 2) for t, _ in r1], "matches_gpu": ok1,
                      "sample": f"one whole 2^{log_top}-term MSM = polynomial 0 of the timed batch, same SRS; median of 3 "
                                f"single-threaded runs ({t_med:.2f} s, {n / t_med:.0f} terms/s); matches GPU result: {ok1}"}
            legs = {"msm_2e%d" % log_top: {"value": round(1 / t_med, 5), "unit": "commitments/s", "seconds": round(t_med, 4), "cores": 1,
                                           "terms_per_s": round(n / t_med, 1), "matches_gpu": ok1}}
            if f_ntt:
                rn = [f.result() for f in f_ntt]
                tn = statistics.median(t for t, _ in rn)
                legs["ntt_2e%d" % log_top] = {"value": round(n / tn, 1), "unit": "elements/s", "seconds": round(tn, 4), "cores": 1,
                                              "transforms_per_s": round(1 / tn, 4), "samples_s": [round(t, 3) for t, _ in rn],
                                              "what": "EvaluationDomain::fft = serial_fft (src/ft.rs:291-333; benches/fft.rs:20-34), orc_fft",
                                              "matches_gpu": all(h == gpu["ntt_sha256"] for _, h in rn)}
            if f_wit is not None:
                tw, ow, nz = f_wit.result()
                legs["witness_2e%d" % log_top] = {"value": round(1 / tw, 5), "unit": "witnesses/s", "seconds": round(tw, 4), "cores": 1,
                                                  "what": "KZGProver::create_witness: long division by X - x, then the (n - 1)-term MSM "
                                                          "(src/coeff_form.rs:66-81; benches/create_witness_coeff_form.rs:28-31)",
                                                  "matches_gpu": bool(not nz and ow == gpu["witness"][2])}
            for log_m in sorted(k for k in gpu["msm"] if k != log_top):     # the smaller sizes BASELINE.md lists: prefixes of the same sample
                m = 1 << log_m
                # both of the oracle's Pippengers, the faster one is the baseline: the 16-bit windows of orc_msm_g1_fast cost 2^20 bucket
                # operations whatever n is, the textbook loop picks its window by size
                rs = list(self.pool.map(_cpu_worker_msm, [(path, n, 0, m)] * 3 + [(path, n, 0, m, "plain")] * 3))
                tf, tp = statistics.median(t for t, _ in rs[:3]), statistics.median(t for t, _ in rs[3:])
                tm = min(tf, tp)
                legs["msm_2e%d" % log_m] = {"value": round(1 / tm, 4), "unit": "commitments/s", "seconds": round(tm, 5), "cores": 1,
                                            "terms_per_s": round(m / tm, 1), "matches_gpu": all(o == gpu["msm"][log_m] for _, o in rs),
                                            "algorithm": "orc_msm_g1_fast" if tf <= tp else "orc_msm_g1 (textbook Pippenger, window by size)",
                                            "seconds_by_algorithm": {"orc_msm_g1_fast": round(tf, 5), "orc_msm_g1": round(tp, 5)}}
            single.update(legs)
            single["all_match_gpu"] = all(v["matches_gpu"] for v in legs.values())
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
            ok2 = all(o == gpu["msm"][log_top] for _, o in r2)
            allc = {"value": round(use * scale / wall, 4), "unit": "commitments/s", "cores": use, "kind": "port", "matches_gpu": ok2,
                    "sample": f"{use} concurrent whole 2^{log_top}-term MSMs, one per worker process; os.cpu_count() = "
                              f"{os.cpu_count()}, usable parallelism measured with {w} workers on 2^15-term slices: {p_eff:.1f} cores; "
                              f"wall {wall:.2f} s, slowest worker {max(t for t, _ in r2):.2f} s; all match the GPU result: {ok2}"}
            if f_ntt:
                t0 = time.perf_counter()
                r3 = list(self.pool.map(_cpu_worker_ntt, [(path, n, log_top)] * use))
                wall3 = time.perf_counter() - t0
                allc["ntt_2e%d" % log_top] = {"value": round(use * n / wall3, 1), "unit": "elements/s", "cores": use, "wall_s": round(wall3, 3),
                                              "sample": f"{use} concurrent whole transforms, one per worker process",
                                              "matches_gpu": all(h == gpu["ntt_sha256"] for _, h in r3)}
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
