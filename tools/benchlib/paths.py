"""The other BASELINE readings of the line (`paths`): configs[2] / [3], the reference benches' u64 distribution, 2^16 / 2^24 spots,
the blocking-caller shapes.  Timed AFTER the headline region; every reading is checked through tools.benchlib.checks."""
import ctypes
import time

from . import checks
from .common import *  # noqa: F401,F403


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
    G = checks.g1_generator()
    ptau = checks.poly_eval_bytes(host_coeffs, n, TAU)
    chk = {}
    res["checked_against_oracle"] = chk
    chk["commit_coeff"] = bool(commitment == checks.g1_mul(G, ptau))

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
    reps = 20
    ntt_ms = timeit(ntt, reps=20, warm=20)       # wall time of the blocking call, profiling off
    engine.prof_enable(True)                    # kernel times: HIP events on the engine's stream (their recording costs wall time)
    engine.prof_reset()
    for _ in range(reps + 1):                   # exactly reps + 1 profiled calls (timeit warms up by time)
        ntt()
    prof = engine.prof_all()
    engine.prof_enable(False)
    kern_ms = sum(v[1] for k, v in prof.items() if k.startswith("k_ntt")) / (reps + 1)
    res["ntt_2e%d_ms" % log_n] = round(ntt_ms, 4)
    nbytes = NTT_BYTES_PER_ELEM * n
    fr_muls = (n // 2) * log_n
    res["ntt_roofline"] = {
        "bound": "valu", "frac_kind": "fr_multiplies_per_s_against_the_library_multiply_rate (see mad_frac for the measured multiply-add issue rate)", "kernels": {k: round(v[1] / (reps + 1), 4) for k, v in sorted(prof.items()) if k.startswith("k_ntt")},
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
    chk["commit_eval"] = bool(out.raw == checks.g1_mul(G, ptau))
    _, _, omega_n = kzg_amd.compute_omega(n)
    ev_head = ev.download(2, offset=n - 2)       # ... and two of the NTT's outputs against direct Horner evaluation by the oracle
    chk["ntt_outputs_sampled"] = all(int.from_bytes(ev_head[32 * i:32 * i + 32], "little") == checks.poly_eval_bytes(host_coeffs, n, pow(omega_n, n - 2 + i, R))
                                        for i in range(2))
    # config 4, single opening and batched k = 256
    x = kzg_amd.splitmix_scalar(99, 0)
    y = engine.poly_eval(coeffs, x)

    def witness():
        rc = lib.kzg_witness_coeff(ctx, srs.handle, coeffs.ptr, n, b32(x), b32(y), coeffs.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
        assert rc == 0, engine.last_error()
    res["witness_coeff_ms"] = round(timeit(witness), 3)
    chk["witness_coeff"] = bool(y == checks.poly_eval_bytes(host_coeffs, n, x) and out.raw == checks.g1_mul(G, (ptau - y) * pow(TAU - x, -1, R) % R))
    m = 12345 % n
    xm = pow(kzg_amd.compute_omega(n)[2], m, R)
    ym = engine.poly_eval(coeffs, xm)
    rc = lib.kzg_witness_coeff(ctx, srs.handle, coeffs.ptr, n, b32(xm), b32(ym), coeffs.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
    w_coeff = out.raw

    def witness_eval():
        assert lib.kzg_witness_eval(ctx, lag.handle, ev.ptr, n, m, ev.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT) == 0, engine.last_error()
    res["witness_eval_ms"] = round(timeit(witness_eval), 3)
    res["witness_eval_equals_witness_coeff"] = bool(rc == 0 and out.raw == w_coeff)
    chk["witness_eval"] = bool(out.raw == checks.g1_mul(G, (ptau - checks.poly_eval_bytes(host_coeffs, n, xm)) * pow(TAU - xm, -1, R) % R))
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
    chk["witness_batched_k%d" % k] = bool(
        rlen.value == (k if k > 1 else 2) and all(checks.poly_eval(Icoef, xs[i]) == ys[i] for i in range(0, k, max(1, k // 8)))
        and all(checks.poly_eval_bytes(host_coeffs, n, xs[i]) == ys[i] for i in (0, k - 1))
        and out.raw == checks.g1_mul(G, (ptau - checks.poly_eval(Icoef, TAU)) * pow(Ztau, -1, R) % R))
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
            G = checks.g1_generator()
            which = [batch - 1] if log_m == 24 else sorted({0, batch // 2, batch - 1})
            res["commit_2e%d" % log_m]["checked_against_oracle"] = all(
                out.raw[96 * b:96 * b + 96] == checks.g1_mul(G, checks.poly_eval_bytes(view_of(sc, b * m, m).download(), m, TAU)) for b in which)
            res["commit_2e%d" % log_m]["checked_commitments"] = which
        except Exception as e:  # noqa: BLE001
            res["commit_2e%d" % log_m]["checked_against_oracle"] = "check failed to run: %s" % e
        sc.free()
        p.gs.free()
    return res


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
        G = checks.g1_generator()
        which = sorted({0, batch // 2, batch - 1})
        res["commit_u64_checked_against_oracle"] = all(
            out.raw[96 * b:96 * b + 96] == checks.g1_mul(G, checks.poly_eval_bytes(view(kzg_amd, sc, b * n, n).download(), n, TAU)) for b in which)
    except Exception as e:  # noqa: BLE001
        res["commit_u64_checked_against_oracle"] = "check failed to run: %s" % e
    sc.free()
    return res


