"""-m gpu parity tests of the verifier half through the C ABI: G2 parameters, G2 multi-exponentiation, pairing
checks and KZGVerifier / KZGVerifierEvalForm (src/coeff_form.rs:114-183, src/eval_form.rs:149-218), against the
python oracle (oracle/pairing_model.py).  The scenarios restate the reference's own verifier tests
(src/coeff_form.rs:279-400) with a known tau."""
import ctypes
import random

import pytest

import kzg_amd
from kzg_amd import _lib as L
from oracle import c_oracle as C
from oracle import kzg_model as M, pairing_model as P

pytestmark = pytest.mark.gpu


def g1b(p):  # affine Montgomery 96 B
    return M.g1_to_affine_mont(p)


def random_polynomial(rng, min_coeffs, max_coeffs):  # src/coeff_form.rs:207-219
    num = rng.randrange(min_coeffs, max_coeffs)
    coeffs = [0] * max_coeffs
    for i in range(num):
        coeffs[i] = rng.getrandbits(64)
    return kzg_amd.Polynomial(coeffs)


def test_setup_g2_matches_oracle_in_every_format(engine):
    tau = 0x1234567
    hs = kzg_amd.setup_g2(engine, tau, 5)
    want = P.setup_g2(tau, 5)
    assert len(hs) == 5
    assert hs.download() == b"".join(P.g2_to_affine_mont(p) for p in want)
    assert hs.download(pfmt=L.G2_JACOBIAN_MONT) == b"".join(P.g2_to_jacobian_mont(p) for p in want)
    assert hs.download(pfmt=L.G2_UNCOMPRESSED) == b"".join(P.g2_to_uncompressed(p) for p in want)
    assert hs.download(1, 3, pfmt=L.G2_COMPRESSED) == b"".join(P.g2_to_compressed(p) for p in want[1:4])
    # upload round trips (incl. the identity) in every format
    pts = want + [None, P.g2_neg(want[2])]
    enc = {L.G2_AFFINE_MONT: P.g2_to_affine_mont, L.G2_JACOBIAN_MONT: P.g2_to_jacobian_mont,
           L.G2_UNCOMPRESSED: P.g2_to_uncompressed, L.G2_COMPRESSED: P.g2_to_compressed}
    for fmt, f in enc.items():
        up = kzg_amd.SrsG2.upload(engine, b"".join(f(p) for p in pts), len(pts), fmt)
        assert up.download(pfmt=L.G2_UNCOMPRESSED) == b"".join(P.g2_to_uncompressed(p) for p in pts), fmt
        up.free()
    # a point off the twist / a non-residue x are rejected
    bad = bytearray(P.g2_to_uncompressed(want[1]))
    bad[-1] ^= 1
    with pytest.raises(kzg_amd.EngineError):
        kzg_amd.SrsG2.upload(engine, bytes(bad), 1, L.G2_UNCOMPRESSED)
    x = (5, 0)
    while P.f2_sqrt(P.f2_add(P.f2_mul(P.f2_sqr(x), x), P.G2_B)) is not None:
        x = (x[0] + 1, 0)
    badc = bytearray(x[1].to_bytes(48, "big") + x[0].to_bytes(48, "big"))
    badc[0] |= 0x80
    with pytest.raises(kzg_amd.EngineError):
        kzg_amd.SrsG2.upload(engine, bytes(badc), 1, L.G2_COMPRESSED)
    hs.free()


def test_msm_g2_matches_oracle(engine):
    rng = random.Random(5)
    tau = rng.getrandbits(64)
    hs = kzg_amd.setup_g2(engine, tau, 70)
    want = P.setup_g2(tau, 70)
    for n, off in [(1, 0), (2, 3), (70, 0), (33, 37), (0, 0)]:
        sc = [rng.randrange(M.R) for _ in range(n)]
        if n > 2:
            sc[0], sc[1] = 0, M.R - 1
        got = hs.msm(sc, offset=off, ofmt=L.G2_UNCOMPRESSED)
        assert got == P.g2_to_uncompressed(P.g2_multi_exp(want[off:off + n], sc)), (n, off)
    with pytest.raises(kzg_amd.ReferencePanic):
        hs.msm([1] * 5, offset=68)
    hs.free()


def test_pairing_check_matches_oracle(engine):
    rng = random.Random(6)
    a, b = rng.randrange(M.R), rng.randrange(M.R)
    Pa, Qb = M.g1_mul(M.G1, a), P.g2_mul(P.G2, b)
    neg_ab = M.g1_neg(M.g1_mul(M.G1, a * b % M.R))
    wrong = M.g1_neg(M.g1_mul(M.G1, (a * b + 1) % M.R))
    checks = [[(Pa, Qb), (neg_ab, P.G2)],        # e(aG, bH) e(-abG, H) = 1
              [(Pa, Qb), (wrong, P.G2)],         # off by one
              [(None, Qb), (Pa, None)],          # identity members contribute 1
              [(M.G1, P.G2), (M.g1_neg(M.G1), P.G2)],
              [(M.G1, P.G2), (M.G1, P.G2)]]
    g1 = b"".join(g1b(p) for c in checks for p, _ in c)
    g2 = b"".join(P.g2_to_affine_mont(q) for c in checks for _, q in c)
    ok = ctypes.create_string_buffer(len(checks))
    rc = engine.lib.kzg_pairing_check(engine.ctx, g1, L.G1_AFFINE_MONT, g2, L.G2_AFFINE_MONT, 2, len(checks), ok)
    assert rc == 0, engine.last_error()
    want = [P.pairing_product_is_one(c) for c in checks]
    assert [bool(x) for x in ok.raw] == want == [True, False, True, True, False]
    # three- and four-pair products, zcash encodings
    c3 = [(Pa, Qb), (M.g1_mul(M.G1, 7), P.g2_mul(P.G2, 9)), (M.g1_neg(M.g1_mul(M.G1, (a * b + 63) % M.R)), P.G2)]
    ok = ctypes.create_string_buffer(1)
    rc = engine.lib.kzg_pairing_check(engine.ctx, b"".join(M.g1_to_compressed(p) for p, _ in c3), L.G1_ZCASH_COMPRESSED,
                                      b"".join(P.g2_to_compressed(q) for _, q in c3), L.G2_COMPRESSED, 3, 1, ok)
    assert rc == 0 and ok.raw == b"\x01"


def test_verify_eval_reference_scenarios(engine):  # test_eval_basic (src/coeff_form.rs:317-342)
    rng = random.Random(69)
    tau = rng.getrandbits(64)
    params = kzg_amd.setup(engine, tau, 13)
    prover, verifier = kzg_amd.KZGProver(params), kzg_amd.KZGVerifier(params)
    p = random_polynomial(rng, 5, 13)
    c = prover.commit(p)
    x = rng.getrandbits(64)
    y = p.eval(engine, x)
    w = prover.create_witness(p, (x, y))
    assert verifier.verify_eval((x, y), c, w)
    assert not verifier.verify_eval((x, (y + 12345) % M.R), c, w)
    assert not verifier.verify_eval(((x + 1) % M.R, y), c, w)
    # the oracle's verifier agrees on both
    op = P.setup(tau, 13)
    ov = P.KZGVerifier(op)
    cP, wP = M.g1_from_uncompressed(prover.commit(p, ofmt=L.G1_ZCASH_UNCOMPRESSED)), \
        M.g1_from_uncompressed(prover.create_witness(p, (x, y), ofmt=L.G1_ZCASH_UNCOMPRESSED))
    assert ov.verify_eval((x, y), cP, wP) and not ov.verify_eval((x, (y + 12345) % M.R), cP, wP)
    # degree-1 edge case: p = 3 + X at (1, 4)
    p1 = kzg_amd.Polynomial([3, 1] + [0] * 11)
    c1 = prover.commit(p1)
    w1 = prover.create_witness(p1, (1, 4))
    assert verifier.verify_eval((1, 4), c1, w1)
    assert not verifier.verify_eval((1, 5), c1, w1)
    # many openings in one launch, mixed verdicts, zcash-compressed inputs; a constant polynomial's witness is the identity
    pc = kzg_amd.Polynomial([42] + [0] * 12)
    cc, wc = prover.commit(pc), prover.create_witness(pc, (9, 42))
    assert wc == bytes(96)
    pts = [(x, y), (x, (y + 1) % M.R), (1, 4), (9, 42), (9, 43)]
    got = verifier.verify_eval_many(pts, [c, c, c1, cc, cc], [w, w, w1, wc, wc])
    assert got == [True, False, True, True, False]
    comp = lambda blob: M.g1_to_compressed(M.g1_from_uncompressed(blob))  # noqa: E731
    cz = prover.commit(p, ofmt=L.G1_ZCASH_UNCOMPRESSED)
    wz = prover.create_witness(p, (x, y), ofmt=L.G1_ZCASH_UNCOMPRESSED)
    assert verifier.verify_eval((x, y), comp(cz), comp(wz), pfmt=L.G1_ZCASH_COMPRESSED)
    params.gs.free()
    params.hs.free()


def test_verify_eval_batched_reference_scenarios(engine):  # test_eval_batched{,_all_points} (:344-399)
    rng = random.Random(70)
    tau = rng.getrandbits(64)
    params = kzg_amd.setup(engine, tau, 15)
    prover, verifier = kzg_amd.KZGProver(params), kzg_amd.KZGVerifier(params)
    p = random_polynomial(rng, 8, 15)
    c = prover.commit(p)
    xs = [rng.getrandbits(64) for _ in range(8)]
    ys = [p.eval(engine, x) for x in xs]
    w = prover.create_witness_batched(p, xs, ys)
    assert verifier.verify_eval_batched(xs, c, w)
    xs2 = [rng.getrandbits(64) for _ in range(8)]
    assert not verifier.verify_eval_batched(xs2, c, w)
    # oracle verdicts on the same data
    ov = P.KZGVerifier(P.setup(tau, 15))
    cP = M.g1_from_uncompressed(prover.commit(p, ofmt=L.G1_ZCASH_UNCOMPRESSED))
    wP = M.g1_from_uncompressed(prover.create_witness_batched(p, xs, ys, ofmt=L.G1_ZCASH_UNCOMPRESSED).w)
    rP = M.Polynomial.new_from_coeffs(list(w.r.coeffs), w.r.degree)
    assert ov.verify_eval_batched(xs, cP, wP, rP) and not ov.verify_eval_batched(xs2, cP, wP, rP)
    # all points: as many openings as coefficients
    p2 = random_polynomial(rng, 13, 14)
    c2 = prover.commit(p2)
    xs = [rng.getrandbits(64) for _ in range(p2.num_coeffs())]
    ys = [p2.eval(engine, x) for x in xs]
    w2 = prover.create_witness_batched(p2, xs, ys)
    assert verifier.verify_eval_batched(xs, c2, w2)
    # one point: the reference's interpolant quirk (r = X + (y - x)) still verifies, as it does upstream
    w1 = prover.create_witness_batched(p, xs[:1], [p.eval(engine, xs[0])])
    rP1 = M.Polynomial.new_from_coeffs(list(w1.r.coeffs), w1.r.degree)
    got = verifier.verify_eval_batched(xs[:1], c, w1)
    assert got == ov.verify_eval_batched(xs[:1], cP, M.g1_from_uncompressed(
        prover.create_witness_batched(p, xs[:1], [p.eval(engine, xs[0])], ofmt=L.G1_ZCASH_UNCOMPRESSED).w), rP1)
    # more points than hs holds -> the reference's slice panic
    small = kzg_amd.setup(engine, tau, 15, g2_len=4)
    with pytest.raises(kzg_amd.ReferencePanic):
        kzg_amd.KZGVerifier(small).verify_eval_batched(xs2, c, w)
    for prm in (params, small):
        prm.gs.free()
        prm.hs.free()


def test_eval_form_verifier(engine):  # src/eval_form.rs:173-217
    rng = random.Random(71)
    tau = rng.getrandbits(64)
    d = 16
    params = kzg_amd.setup(engine, tau, d)
    lag_g = kzg_amd.setup_lagrange(engine, tau, d)
    lag_h = kzg_amd.setup_lagrange_g2(engine, tau, d)
    want_h = P.lagrange_basis_g2_known_tau(tau, d)
    assert lag_h.download(pfmt=L.G2_UNCOMPRESSED) == b"".join(P.g2_to_uncompressed(p) for p in want_h)
    # compute_lagrange_basis from hs alone (src/eval_form.rs:254-280) yields the same elements
    full = kzg_amd.setup(engine, tau, d, g2_len=d)
    from_hs = kzg_amd.compute_lagrange_basis_g2(full)
    assert from_hs.download() == lag_h.download()
    for h in (full.gs, full.hs, from_hs):
        h.free()
    prover = kzg_amd.KZGProverEvalForm(params, lag_g)
    verifier = kzg_amd.KZGVerifierEvalForm(params, lag_g, lag_h)
    coeffs = [rng.getrandbits(64) for _ in range(d)]
    evals = kzg_amd.EvaluationDomain.from_coeffs(coeffs)
    evals.fft(engine)
    c = prover.commit(evals)
    for i in (0, 5, d - 1):
        w = prover.create_witness(evals, i)
        y = evals.coeffs[i]
        assert verifier.verify_eval((i, y), c, w)
        assert not verifier.verify_eval((i, (y + 1) % M.R), c, w)
        assert not verifier.verify_eval(((i + 1) % d, y), c, w)
    # verify_eval_all exactly as written upstream: the oracle restates the same formula
    ov = P.KZGVerifierEvalForm(P.setup(tau, d), M.lagrange_basis_g1_known_tau(tau, d), want_h)
    cP = M.g1_from_uncompressed(prover.commit(evals, ofmt=L.G1_ZCASH_UNCOMPRESSED))
    for wit in (prover.create_witness_all(), prover.create_witness(evals, 3)):
        wP = None if wit == bytes(96) else \
            M.g1_from_uncompressed(prover.create_witness(evals, 3, ofmt=L.G1_ZCASH_UNCOMPRESSED))
        assert verifier.verify_eval_all(evals.coeffs, c, wit) == ov.verify_eval_all(evals.coeffs, cP, wP)
    for h in (params.gs, params.hs, lag_g, lag_h):
        h.free()


def test_verify_eval_degenerate_secret(engine):
    """tau = 0: hs[1] (and gs[1..]) are the identity -- the stored-lines path must treat the pair as 1, like the oracle."""
    params = kzg_amd.setup(engine, 0, 6)
    assert params.hs.download(1, 1) == bytes(192)
    prover, verifier = kzg_amd.KZGProver(params), kzg_amd.KZGVerifier(params)
    p = kzg_amd.Polynomial([7, 3, 9, 1, 0, 0])
    c = prover.commit(p)
    assert c == g1b(M.g1_mul(M.G1, 7))
    x, y = 5, p.eval(engine, 5)
    w = prover.create_witness(p, (x, y))
    op = P.setup(0, 6, fast=False)
    ov = P.KZGVerifier(op)
    cP = M.g1_mul(M.G1, 7)
    wP = M.KZGProver(op).create_witness(M.Polynomial([7, 3, 9, 1, 0, 0]), (x, y))
    assert w == g1b(wP)
    for pt in [(x, y), (x, (y + 1) % M.R), (0, 7), (0, 8)]:
        assert verifier.verify_eval(pt, c, w) == ov.verify_eval(pt, cP, wP), pt
    params.gs.free()
    params.hs.free()


def test_batched_opening_5000_points_prover_and_verifier(engine):
    """create_witness_batched + verify_eval_batched (src/coeff_form.rs:83-111, :144-182) with 5000 opening points -- above the 4096 the
    interpolation kernels took until round 4 (limit now 16384).  The witness against the known-tau identity with p(tau), I(tau) by the
    oracle; the interpolant at sampled opening points; the pairing check accepts it and rejects a shifted point set."""
    rng = random.Random(71)
    tau = rng.getrandbits(64)
    n, k = 6000, 5000
    params = kzg_amd.setup(engine, tau, n, g2_len=k + 1)
    prover, verifier = kzg_amd.KZGProver(params), kzg_amd.KZGVerifier(params)
    coeffs = [rng.randrange(M.R) for _ in range(n)]
    p = kzg_amd.Polynomial(coeffs)
    c = prover.commit(p)
    xs = [rng.randrange(M.R) for _ in range(k)]
    ys = [C.poly_eval(coeffs, x) for x in xs]
    w = prover.create_witness_batched(p, xs, ys)
    I = list(w.r.coeffs)
    assert len(I) == k and all(C.poly_eval(I, xs[i]) == ys[i] for i in range(0, k, 499))
    Z = 1
    for x in xs:
        Z = Z * (tau - x) % M.R
    assert w.w == C.g1_mul(C.g1_generator(), (C.poly_eval(coeffs, tau) - C.poly_eval(I, tau)) * pow(Z, -1, M.R) % M.R)
    assert verifier.verify_eval_batched(xs, c, w)
    assert not verifier.verify_eval_batched([(x + 1) % M.R for x in xs], c, w)
    ys[1234] = (ys[1234] + 1) % M.R
    with pytest.raises(kzg_amd.PointNotOnPolynomial):
        prover.create_witness_batched(p, xs, ys)
    params.gs.free()
    params.hs.free()
