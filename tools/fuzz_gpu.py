#!/usr/bin/env python3
"""Randomised differential run of the HIP engine against the C oracle (checker only): MSM with random sizes, offsets,
window sizes, scalar shapes and batch counts; NTT round trips and values; single-point witnesses.  Time-boxed:
    python tools/fuzz_gpu.py [seconds] [seed]
Prints one line per failure and a summary; exit code 1 if anything differed."""
import ctypes
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kzg_amd  # noqa: E402
from kzg_amd import _lib as L  # noqa: E402
from oracle import c_oracle as C, kzg_model as M  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
R = M.R
e = kzg_amd.Engine(0)
TAU = rng.getrandbits(64)
t_end = time.time() + budget
fails = 0
cases = {"msm": 0, "msm_batch": 0, "ntt": 0, "witness": 0, "witness_batched": 0, "eval_form": 0, "group": 0, "lagrange_intt": 0, "poly_mul": 0,
         "group_witness_eval": 0}
group = kzg_amd.DeviceGroup([0])
group.set_option("always_gather", 1)


def rand_scalar(kind):
    if kind == 0:
        return rng.randrange(R)
    if kind == 1:
        return rng.getrandbits(64)
    if kind == 2:
        return rng.choice([0, 1, R - 1, R - 2, 1 << 255 if (1 << 255) < R else R - 3, 0x8000, 0x7fff, 0xffff, 0x10000,
                           int("8000" * 16, 16) % R, int("7fff" * 16, 16) % R, (1 << 128) - 1])
    if kind == 4:
        return rng.getrandbits(rng.choice([1, 1, 8]))      # bits / bytes: a handful of buckets hold everything
    return rng.getrandbits(rng.randrange(1, 255))


while time.time() < t_end:
    wb = rng.choice([0, 0, 0, 4, 7, 9, 12, 13, 16, 17, 17, 17, 18, 19, 20])
    e.set_option("window_bits", wb)
    e.set_option("sort_single_pass", rng.choice([0, 0, 1]))  # c = 17: two-level sort (default) / single-pass sort
    rows = rng.choice([0, 0, 0, 1, 2, 3, 5, 11])            # low-memory SRS: multi-pass MSM
    e.set_option("window_rows", rows)
    naf = 18 if (wb == 0 and rows == 0 and rng.random() < 0.3) else 0   # positional tables + width-18 NAF digits (opt-in layout)
    e.set_option("naf_window", naf)
    e.set_option("heavy_bins", rng.choice([0, 0, 1, 2]))       # oversized sort bins in slices: adaptive / always / never
    e.set_option("tail_quads", rng.randrange(2))             # latency-mode tail kernels on / off
    e.set_option("hw_queues", rng.choice([0, 0, 1, 3, 4, 6]))  # pipeline plans
    e.set_option("streams", rng.choice([1, 2, 4, 8]))
    e.set_option("defer_tail", rng.choice([1, 1, 0]))           # batched MSMs: a lane's tail behind the sort of its next MSM / right away
    nmax = rng.choice([1, 2, 5, 33, 257, 1000, 4097, 20000, 70000, 70000, 300000])
    params = kzg_amd.setup(e, TAU, nmax, g2_len=0)
    srs_blob = params.gs.download()
    assert srs_blob == C.setup_g1(TAU, nmax), "setup mismatch"
    for _ in range(4):
        n = rng.randrange(0, nmax + 1)
        off = rng.randrange(0, nmax - n + 1)
        kind = rng.randrange(5)
        sc = [rand_scalar(kind if (kind == 4 or rng.random() < 0.8) else rng.randrange(4)) for _ in range(n)]
        if n and rng.random() < 0.2:
            sc = [sc[0]] * n  # all equal: one bucket per window
        elif n and rng.random() < 0.2:
            pool = [rand_scalar(0) for _ in range(rng.randrange(1, 6))]  # a handful of huge buckets: the overflow slices
            sc = [rng.choice(pool) for _ in range(n)]
        want = C.msm_g1(srs_blob[96 * off:96 * (off + n)], sc) if n else bytes(96)
        got = e.msm(params.gs, sc, offset=off) if n else e.msm(params.gs, [], n=0, offset=off)
        cases["msm"] += 1
        if got != want:
            fails += 1
            print("MSM MISMATCH", dict(window_bits=wb, rows=rows, naf=naf, nmax=nmax, n=n, off=off, kind=kind, seed=seed), flush=True)
    if nmax >= 33:
        n = rng.randrange(1, min(nmax, 3000) + 1)
        batch = rng.randrange(1, 12) if rng.random() < 0.7 else rng.randrange(12, 40)   # deep batches: two MSMs in flight per lane
        sc = [rand_scalar(rng.randrange(4)) for _ in range(n * batch)]
        got = e.msm_batch(params.gs, sc, n, batch)
        cases["msm_batch"] += 1
        for b in range(batch):
            if got[b] != C.msm_g1(srs_blob[:96 * n], sc[b * n:(b + 1) * n]):
                fails += 1
                print("MSM_BATCH MISMATCH", dict(window_bits=wb, n=n, batch=batch, b=b, seed=seed), flush=True)
        # single-point witness on a random polynomial (remainder test both ways)
        n = rng.randrange(2, min(nmax, 5000) + 1)
        coeffs = [rand_scalar(rng.randrange(2)) for _ in range(n)]
        if coeffs[-1] == 0:
            coeffs[-1] = 1
        poly = kzg_amd.Polynomial(coeffs)
        x = rand_scalar(rng.randrange(2))
        y = C.poly_eval(coeffs, x)
        prover = kzg_amd.KZGProver(params)
        cases["witness"] += 1
        if (TAU - x) % R:
            want = C.g1_mul(C.g1_generator(), (C.poly_eval(coeffs, TAU) - y) * pow(TAU - x, -1, R) % R)
            if prover.create_witness(poly, (x, y)) != want:
                fails += 1
                print("WITNESS MISMATCH", dict(n=n, seed=seed), flush=True)
        try:
            prover.create_witness(poly, (x, (y + 1) % R))
            fails += 1
            print("WITNESS accepted a wrong y", dict(n=n, seed=seed), flush=True)
        except kzg_amd.PointNotOnPolynomial:
            pass
        # batched witness: r == the interpolant, w == [(p(tau) - I(tau)) / Z(tau)] G  (known-tau identity)
        # (one round in three: a long polynomial, where the division's transforms are two-pass and Z's takes the short-input path
        # or not depending on k against the first-row limit 2^floor(log N / 2))
        big = rng.random() < 0.34 and nmax >= 20000
        n = rng.randrange(9000, min(nmax, 70000) + 1) if big else rng.randrange(2, min(nmax, 2000) + 1)
        k = rng.choice([rng.randrange(1, 24), rng.randrange(60, 70), rng.randrange(120, 135), rng.randrange(250, 262)]) if big else rng.randrange(1, min(n + 2, 24))
        coeffs = [rand_scalar(rng.randrange(2)) for _ in range(n)]
        coeffs[-1] = coeffs[-1] or 1
        poly = kzg_amd.Polynomial(coeffs)
        xs = rng.sample(range(1, 1 << 20), k) if rng.random() < 0.5 else [rng.randrange(R) for _ in range(k)]
        if rng.random() < 0.3:
            xs[0] = 7  # on the first division coset
        if len(set(xs)) == k and all((TAU - x) % R for x in xs):
            ys = [C.poly_eval(coeffs, x) for x in xs]
            cases["witness_batched"] += 1
            wit = prover.create_witness_batched(poly, xs, ys)
            I = M.Polynomial.lagrange_interpolation(xs, ys)
            zt = 1
            for x in xs:
                zt = zt * (TAU - x) % R
            want = C.g1_mul(C.g1_generator(), (C.poly_eval(coeffs, TAU) - I.eval(TAU)) * pow(zt, -1, R) % R)
            if wit.elem() != want or wit.polynomial().coeffs != I.coeffs[:len(wit.polynomial().coeffs)]:
                fails += 1
                print("WITNESS_BATCHED MISMATCH", dict(n=n, k=k, seed=seed), flush=True)
            ys[-1] = (ys[-1] + 1) % R
            try:
                prover.create_witness_batched(poly, xs, ys)
                fails += 1
                print("WITNESS_BATCHED accepted a wrong y", dict(n=n, k=k, seed=seed), flush=True)
            except kzg_amd.PointNotOnPolynomial:
                pass
    params.gs.free()
    e.set_option("window_rows", 0)
    e.set_option("naf_window", 0)
    # concurrent blocking callers: 6 host threads, each its own polynomial, commit + witness on ONE context and one SRS
    if rng.random() < 0.25:
        import threading
        nn = rng.choice([300, 5000, 40000])
        e.set_option("window_bits", 0)
        pc = kzg_amd.setup(e, TAU, nn, g2_len=0)
        prover = kzg_amd.KZGProver(pc)
        jobs = []
        for t in range(6):
            co = [rand_scalar(rng.randrange(2)) for _ in range(nn)]
            x = rand_scalar(0)
            y = C.poly_eval(co, x)
            ptau = C.poly_eval(co, TAU)
            jobs.append((kzg_amd.Polynomial(co), x, y, C.g1_mul(C.g1_generator(), ptau),
                         C.g1_mul(C.g1_generator(), (ptau - y) * pow(TAU - x, -1, R) % R) if (TAU - x) % R else None))
        bad = []

        def work(t):
            poly, x, y, wc, ww = jobs[t]
            for _ in range(3):
                if prover.commit(poly) != wc:
                    bad.append(("commit", t))
                if ww is not None and prover.create_witness(poly, (x, y)) != ww:
                    bad.append(("witness", t))
        th = [threading.Thread(target=work, args=(t,)) for t in range(6)]
        for x_ in th:
            x_.start()
        for x_ in th:
            x_.join()
        cases["concurrent"] = cases.get("concurrent", 0) + 1
        if bad:
            fails += 1
            print("CONCURRENT CALLERS MISMATCH", dict(n=nn, bad=bad[:4], seed=seed), flush=True)
        pc.gs.free()
    # device group (one GPU, RCCL all-gather forced on): commit and witness over the group
    n = rng.choice([1, 7, 300, 5000])
    gsrs = group.setup(TAU, n)
    coeffs = [rand_scalar(rng.randrange(4)) for _ in range(rng.randrange(1, n + 1))]
    cases["group"] += 1
    if group.commit(gsrs, coeffs) != C.g1_mul(C.g1_generator(), C.poly_eval(coeffs, TAU)):
        fails += 1
        print("GROUP COMMIT MISMATCH", dict(n=n, m=len(coeffs), seed=seed), flush=True)
    gsrs.free()
    # KZGProverEvalForm::create_witness over the group
    if rng.random() < 0.4:
        dd = 1 << rng.randrange(1, 12)
        lag1 = kzg_amd.setup_lagrange(e, TAU, dd)
        glag = group.upload(lag1.download(), dd)
        cf = [rand_scalar(rng.randrange(2)) for _ in range(dd)]
        evs = C.fft(cf)
        mm = rng.randrange(dd)
        wmm = pow(kzg_amd.compute_omega(dd)[2], mm, R)
        cases["group_witness_eval"] += 1
        if (TAU - wmm) % R and group.create_witness_eval(glag, evs, mm) != C.g1_mul(C.g1_generator(), (C.poly_eval(cf, TAU) - evs[mm]) * pow(TAU - wmm, -1, R) % R):
            fails += 1
            print("GROUP EVAL WITNESS MISMATCH", dict(d=dd, m=mm, seed=seed), flush=True)
        glag.free(); lag1.free()
    # fft_mul with a short operand (short-input transform) and with two long ones: product checked by oracle evaluation
    if rng.random() < 0.5:
        na = rng.choice([1, 2, 17, 64, 65, 129, 300, 5000])
        nb = rng.choice([100, 5000, 9000, 33000, 120000])
        pa, pb = [rand_scalar(rng.randrange(2)) for _ in range(na)], [rand_scalar(0) for _ in range(nb)]
        prod = e.poly_mul(pa, pb) if rng.random() < 0.5 else e.poly_mul(pb, pa)
        cases["poly_mul"] += 1
        xx = rand_scalar(0)
        if len(prod) != na + nb - 1 or C.poly_eval(prod, xx) != C.poly_eval(pa, xx) * C.poly_eval(pb, xx) % R:
            fails += 1
            print("POLY_MUL MISMATCH", dict(na=na, nb=nb, seed=seed), flush=True)
    # compute_lagrange_basis from the monomial SRS alone (group iNTT) == the known-tau closed form
    if rng.random() < 0.3:
        d = 1 << rng.randrange(0, 12)
        pm = kzg_amd.setup(e, TAU, d, g2_len=0)
        a, b = kzg_amd.compute_lagrange_basis(pm), kzg_amd.setup_lagrange(e, TAU, d)
        cases["lagrange_intt"] += 1
        if a.download() != b.download():
            fails += 1
            print("LAGRANGE iNTT MISMATCH", dict(d=d, seed=seed), flush=True)
        pm.gs.free(); a.free(); b.free()
    # eval form: commit(fft(p)) == [p(tau)]G, witness at index m == coeff-form witness at omega^m
    log_d = rng.randrange(0, 11)
    d = 1 << log_d
    pe = kzg_amd.setup(e, TAU, d, g2_len=0)
    lag = kzg_amd.setup_lagrange(e, TAU, d)
    coeffs = [rand_scalar(rng.randrange(2)) for _ in range(d)]
    ev = kzg_amd.EvaluationDomain.from_coeffs(coeffs)
    ev.fft(e)
    pf = kzg_amd.KZGProverEvalForm(pe, lag)
    cases["eval_form"] += 1
    ptau = C.poly_eval(coeffs, TAU)
    if pf.commit(ev) != C.g1_mul(C.g1_generator(), ptau):
        fails += 1
        print("EVAL COMMIT MISMATCH", dict(d=d, seed=seed), flush=True)
    m = rng.randrange(d)
    wm = pow(pf.omega(), m, R)
    if (TAU - wm) % R:
        want = C.g1_mul(C.g1_generator(), (ptau - ev.coeffs[m]) * pow(TAU - wm, -1, R) % R)
        if pf.create_witness(ev, m) != want:
            fails += 1
            print("EVAL WITNESS MISMATCH", dict(d=d, m=m, seed=seed), flush=True)
    pe.gs.free()
    lag.free()
    log_n = rng.randrange(0, 15) if rng.randrange(4) else rng.randrange(15, 19)   # one round in four above the two-pass boundary
    xs = [rand_scalar(rng.randrange(4)) % R for _ in range(1 << log_n)]
    got = e.ntt(xs, log_n)
    cases["ntt"] += 1
    if got != C.fft(xs) or e.ntt(got, log_n, inverse=True) != xs:
        fails += 1
        print("NTT MISMATCH", dict(log_n=log_n, seed=seed), flush=True)
e.set_option("window_bits", 0)
e.set_option("sort_single_pass", 0)
e.set_option("defer_tail", 1)
group.close()
summary = "seed %d, %.0f s: fuzz done %s failures: %d" % (seed, budget, cases, fails)
print(summary, flush=True)
try:  # RCCL prints its banner at exit, which pushes this line out of a `tail`: keep a copy where gpurun brings it back
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "fuzz.log"), "a") as f:
        f.write(summary + "\n")
except OSError:
    pass
sys.exit(1 if fails else 0)
