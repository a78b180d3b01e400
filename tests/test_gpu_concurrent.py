"""The reference's call shape: KZGProver is Clone + &self (src/coeff_form.rs:37-64), so many host threads call commit() /
create_witness() at once.  On one kzg_ctx each blocking call leases a lane (runtime.hip, CtxGate in common.h); these tests drive
one context from 16 threads with mixed calls and check every result against the oracle, and share one resident SRS between
contexts."""
import ctypes
import threading

import pytest

import kzg_amd
from kzg_amd import _lib as L
from oracle import c_oracle as C
from oracle import kzg_model as M

pytestmark = pytest.mark.gpu
R = M.R
TAU = 0x5EED5EED5EED5EED
b32 = lambda v: (v % R).to_bytes(32, "little")  # noqa: E731


def _run_threads(n_threads, fn):
    errs = []

    def wrap(t):
        try:
            fn(t)
        except BaseException as e:  # noqa: BLE001
            errs.append((t, repr(e)))

    th = [threading.Thread(target=wrap, args=(t,)) for t in range(n_threads)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs


@pytest.mark.parametrize("n,threads,rounds", [(1 << 12, 16, 6), (1 << 17, 16, 3)])
def test_concurrent_mixed_calls_small(engine, n, threads, rounds):
    """commit (coeff + eval form), create_witness (both forms), a wrong-y witness and an exclusive call (NTT) interleaved from
    16 threads on ONE context: every result equals the oracle's."""
    engine.set_option("streams", 16)
    params = kzg_amd.setup(engine, TAU, n, g2_len=0)
    lag = kzg_amd.setup_lagrange(engine, TAU, n)
    log_n = n.bit_length() - 1
    _, _, omega = kzg_amd.compute_omega(n)
    G = C.g1_generator()
    polys = []
    for t in range(threads):
        buf = engine.alloc_scalars(n).fill_random(9000 + t)
        raw = buf.download()
        ev = engine.alloc_scalars(n)
        ev.upload(C.fft_bytes(raw, log_n))
        ptau = C.poly_eval_bytes(raw, n, TAU)
        x = kzg_amd.splitmix_scalar(31, t)
        y = C.poly_eval_bytes(raw, n, x)
        m = (977 * t + 5) % n
        xm = pow(omega, m, R)
        ym = C.poly_eval_bytes(raw, n, xm)
        polys.append(dict(buf=buf, ev=ev, commit=C.g1_mul(G, ptau), x=x, y=y, m=m,
                          wit=C.g1_mul(G, (ptau - y) * pow(TAU - x, -1, R) % R),
                          wit_m=C.g1_mul(G, (ptau - ym) * pow(TAU - xm, -1, R) % R)))

    def work(t):
        p = polys[t]
        lib, ctx = engine.lib, engine.ctx
        out = ctypes.create_string_buffer(96)
        for r in range(rounds):
            rc = lib.kzg_commit_coeff(ctx, params.gs.handle, p["buf"].ptr, n, p["buf"].sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
            assert rc == 0 and out.raw == p["commit"], ("commit", t, r, rc)
            rc = lib.kzg_witness_coeff(ctx, params.gs.handle, p["buf"].ptr, n, b32(p["x"]), b32(p["y"]), p["buf"].sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
            assert rc == 0 and out.raw == p["wit"], ("witness", t, r, rc)
            rc = lib.kzg_commit_eval(ctx, lag.handle, p["ev"].ptr, n, p["ev"].sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
            assert rc == 0 and out.raw == p["commit"], ("commit_eval", t, r, rc)
            rc = lib.kzg_witness_eval(ctx, lag.handle, p["ev"].ptr, n, p["m"], p["ev"].sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
            assert rc == 0 and out.raw == p["wit_m"], ("witness_eval", t, r, rc)
            rc = lib.kzg_witness_coeff(ctx, params.gs.handle, p["buf"].ptr, n, b32(p["x"]), b32(p["y"] + 1), p["buf"].sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
            assert rc == L.KZG_ERR_POINT_NOT_ON_POLY, ("wrong y", t, r, rc)
            assert b"point not on polynomial" in lib.kzg_last_error(ctx)   # the calling thread's own failure
            if t % 5 == 0:   # an exclusive call in the middle of the others' leases (host-resident data: nothing shared)
                xs = [t, r, 3, 4]
                assert engine.ntt(xs, 2) == C.fft(xs)
            # host-resident coefficients (staged on the leased lane)
            if r == 0 and n <= 1 << 12:
                rc = lib.kzg_commit_coeff(ctx, params.gs.handle, p["buf"].download(), n, L.FR_CANONICAL, 0, out, L.G1_AFFINE_MONT)
                assert rc == 0 and out.raw == p["commit"]

    try:
        _run_threads(threads, work)
    finally:
        engine.set_option("streams", 8)
        for p in polys:
            p["buf"].free()
            p["ev"].free()
        params.gs.free()
        lag.free()


def test_concurrent_2_20_sixteen_threads(engine):
    """16 host threads, degree 2^20 (BASELINE configs[1]'s size), mixed commit / create_witness on one context and one resident
    SRS; expected values from the oracle (downloaded coefficients, Horner, one scalar multiplication each)."""
    n, threads = 1 << 20, 16
    engine.set_option("streams", 16)
    params = kzg_amd.setup(engine, TAU, n, g2_len=0)
    G = C.g1_generator()
    polys = []
    for t in range(threads):
        buf = engine.alloc_scalars(n).fill_random(7000 + t, u64_valued=(t % 4 == 3))
        raw = buf.download()
        ptau = C.poly_eval_bytes(raw, n, TAU)
        x = kzg_amd.splitmix_scalar(47, t)
        y = C.poly_eval_bytes(raw, n, x)
        polys.append(dict(buf=buf, commit=C.g1_mul(G, ptau), x=x, y=y, wit=C.g1_mul(G, (ptau - y) * pow(TAU - x, -1, R) % R)))
        del raw

    def work(t):
        p = polys[t]
        lib, ctx = engine.lib, engine.ctx
        out = ctypes.create_string_buffer(96)
        for r in range(4):
            if (t + r) % 2 == 0:
                rc = lib.kzg_commit_coeff(ctx, params.gs.handle, p["buf"].ptr, n, p["buf"].sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
                assert rc == 0 and out.raw == p["commit"], ("commit", t, r, rc)
            else:
                rc = lib.kzg_witness_coeff(ctx, params.gs.handle, p["buf"].ptr, n, b32(p["x"]), b32(p["y"]), p["buf"].sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
                assert rc == 0 and out.raw == p["wit"], ("witness", t, r, rc)

    try:
        _run_threads(threads, work)
        # and the same SRS from a second context on the same device (an SRS is not bound to the context that built it),
        # both contexts busy at once
        e2 = kzg_amd.Engine(0)

        def work2(t):
            eng = engine if t % 2 == 0 else e2
            out = ctypes.create_string_buffer(96)
            p = polys[t]
            for _ in range(2):
                rc = eng.lib.kzg_commit_coeff(eng.ctx, params.gs.handle, p["buf"].ptr, n, p["buf"].sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
                assert rc == 0 and out.raw == p["commit"], ("two contexts", t, rc)

        _run_threads(8, work2)
        e2.close()
    finally:
        engine.set_option("streams", 8)
        for p in polys:
            p["buf"].free()
        params.gs.free()


def test_concurrent_2_20_every_leased_call(engine):
    """16 host threads at degree 2^20 mixing every call that leases a lane: commit, create_witness, create_witness_batched
    (k = 256: BASELINE configs[3], primary reading), fft, verify_poly -- KZGProver is Clone + &self, so the reference allows all of
    them at once (src/coeff_form.rs:59-111, src/ft.rs:111-140).  Every expected value from the oracle: downloaded coefficients,
    its Horner loop, one scalar multiplication; the NTT against its serial_fft restatement."""
    n, threads, k = 1 << 20, 16, 256
    engine.set_option("streams", 16)
    params = kzg_amd.setup(engine, TAU, n, g2_len=0)
    G = C.g1_generator()
    polys = []
    for t in range(4):  # four polynomials, four threads each
        buf = engine.alloc_scalars(n).fill_random(5100 + t)
        raw = buf.download()
        ptau = C.poly_eval_bytes(raw, n, TAU)
        x = kzg_amd.splitmix_scalar(53, t)
        y = C.poly_eval_bytes(raw, n, x)
        xs = [kzg_amd.splitmix_scalar(600 + t, i) for i in range(k)]
        # the opening values by the engine (256 oracle evaluations at 2^20 would take half a minute per polynomial); a sample of them
        # against the oracle here, and the witness identity below holds only if ALL of them are right (the division is exact iff
        # I agrees with p at every opening point)
        ys = [engine.poly_eval(buf, v) for v in xs]
        assert all(C.poly_eval_bytes(raw, n, xs[i]) == ys[i] for i in (0, 101, 255))
        polys.append(dict(buf=buf, raw=raw, ptau=ptau, commit=C.g1_mul(G, ptau), x=x, y=y,
                          wit=C.g1_mul(G, (ptau - y) * pow(TAU - x, -1, R) % R), xs=xs, ys=ys,
                          xb=kzg_amd.pack_scalars(xs), yb=kzg_amd.pack_scalars(ys), fft=C.fft_bytes(raw, 20)))
    # the batched witness once alone: w == [(p(tau) - I(tau)) / Z(tau)] G with I from the call itself checked at the opening points
    for p in polys:
        out = ctypes.create_string_buffer(96)
        rbuf, rlen = ctypes.create_string_buffer(32 * k), ctypes.c_size_t()
        rc = engine.lib.kzg_witness_coeff_batched(engine.ctx, params.gs.handle, p["buf"].ptr, n, p["xb"], p["yb"], k, p["buf"].sfmt,
                                                  L.IN_DEVICE, out, L.G1_AFFINE_MONT, rbuf, ctypes.byref(rlen))
        assert rc == 0 and rlen.value == k, engine.last_error()
        I = kzg_amd.unpack_scalars(rbuf.raw)
        assert all(C.poly_eval(I, p["xs"][i]) == p["ys"][i] for i in range(0, k, 15))
        Z = 1
        for v in p["xs"]:
            Z = Z * (TAU - v) % R
        p["wb"] = C.g1_mul(G, (p["ptau"] - C.poly_eval(I, TAU)) * pow(Z, -1, R) % R)
        assert out.raw == p["wb"]
        p["I"] = rbuf.raw
    work_bufs = [engine.alloc_scalars(n) for _ in range(threads)]

    def work(t):
        p = polys[t % 4]
        lib, ctx = engine.lib, engine.ctx
        out = ctypes.create_string_buffer(96)
        rbuf, rlen = ctypes.create_string_buffer(32 * k), ctypes.c_size_t()
        ok = ctypes.c_int(0)
        for r in range(5):
            what = (t + r) % 5
            if what == 0:
                rc = lib.kzg_commit_coeff(ctx, params.gs.handle, p["buf"].ptr, n, p["buf"].sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
                assert rc == 0 and out.raw == p["commit"], ("commit", t, r, rc)
            elif what == 1:
                rc = lib.kzg_witness_coeff(ctx, params.gs.handle, p["buf"].ptr, n, b32(p["x"]), b32(p["y"]), p["buf"].sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
                assert rc == 0 and out.raw == p["wit"], ("witness", t, r, rc)
            elif what == 2:
                rc = lib.kzg_witness_coeff_batched(ctx, params.gs.handle, p["buf"].ptr, n, p["xb"], p["yb"], k, p["buf"].sfmt, L.IN_DEVICE, out,
                                                   L.G1_AFFINE_MONT, rbuf, ctypes.byref(rlen))
                assert rc == 0 and out.raw == p["wb"] and rbuf.raw == p["I"], ("witness_batched", t, r, rc)
            elif what == 3:
                w = work_bufs[t]
                w.upload(p["raw"])
                assert lib.kzg_ntt_fr(ctx, w.ptr, 20, 0, L.IN_DEVICE) == 0
                assert w.download() == p["fft"], ("fft", t, r)
                assert lib.kzg_ntt_fr(ctx, w.ptr, 20, 1, L.IN_DEVICE) == 0
                assert w.download() == p["raw"], ("ifft", t, r)
            else:
                rc = lib.kzg_verify_poly_coeff(ctx, params.gs.handle, p["commit"], L.G1_AFFINE_MONT, p["buf"].ptr, n, p["buf"].sfmt, L.IN_DEVICE,
                                               ctypes.byref(ok))
                assert rc == 0 and ok.value == 1, ("verify_poly", t, r, rc)
                # a wrong opening value in the batch: the calling thread's own error
                yb_bad = p["yb"][:32] + b32(p["ys"][1] + 1) + p["yb"][64:]
                rc = lib.kzg_witness_coeff_batched(ctx, params.gs.handle, p["buf"].ptr, n, p["xb"], yb_bad, k, p["buf"].sfmt, L.IN_DEVICE, out,
                                                   L.G1_AFFINE_MONT, rbuf, ctypes.byref(rlen))
                assert rc == L.KZG_ERR_POINT_NOT_ON_POLY, ("wrong y in the batch", t, r, rc)

    try:
        _run_threads(threads, work)
    finally:
        engine.set_option("streams", 8)
        for p in polys:
            p["buf"].free()
        for w in work_bufs:
            w.free()
        params.gs.free()


def test_concurrent_batched_openings_share_and_churn_point_sets():
    """create_witness_batched from 16 threads on one context whose point-set cache has FOUR slots: threads 0-7 all open their own
    polynomials at ONE shared point set (concurrent hits on one entry, pinned by several readers), threads 8-15 each cycle through three
    point sets of their own (misses, fills and evictions while the others read).  Every witness is the oracle's
    [(p(tau) - I(tau)) / Z(tau)]G and every interpolant passes through its points."""
    import random
    n, k, rounds = 1 << 12, 8, 6
    e = kzg_amd.Engine(0)
    e.set_option("witness_cache_slots", 4)
    params = kzg_amd.setup(e, TAU, n, g2_len=0)
    prover = kzg_amd.KZGProver(params)
    G = C.g1_generator()
    rng99 = random.Random(99)
    shared = [rng99.randrange(R) for _ in range(k)]

    def work(t):
        rng = random.Random(1000 + t)
        own = [[rng.randrange(R) for _ in range(k)] for _ in range(3)]
        for rnd in range(rounds):
            xs = shared if t < 8 else own[rnd % 3]
            coeffs = [rng.randrange(R) for _ in range(n)]
            ys = [C.poly_eval(coeffs, x) for x in xs]
            wit = prover.create_witness_batched(kzg_amd.Polynomial(coeffs), xs, ys)
            icoef = wit.r.slice_coeffs()
            assert all(C.poly_eval(icoef, x) == y for x, y in zip(xs, ys))
            z = 1
            for x in xs:
                z = z * (TAU - x) % R
            assert wit.w == C.g1_mul(G, (C.poly_eval(coeffs, TAU) - C.poly_eval(icoef, TAU)) * pow(z, -1, R) % R), (t, rnd)

    _run_threads(16, work)
    h, m = ctypes.c_uint64(), ctypes.c_double()
    assert e.lib.kzg_prof_get(e.ctx, b"point_set_cache", ctypes.byref(h), ctypes.byref(m)) == 0
    assert h.value >= 8 and m.value >= 8, (h.value, m.value)      # both paths were exercised
    params.gs.free()
    e.close()
