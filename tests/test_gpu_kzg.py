"""End-to-end KZGProver / KZGProverEvalForm parity via the known-tau identities and the C
oracle, mirroring the reference's tests (src/coeff_form.rs:271-398, src/eval_form.rs:318-483)."""
import ctypes
import random

import pytest

import kzg_amd
from kzg_amd import _lib as L
from oracle import c_oracle as C
from oracle import kzg_model as M
from tests.gpu_common import rand_scalars

pytestmark = pytest.mark.gpu
G = C.g1_generator


def random_polynomial(rng, min_coeffs, max_coeffs):
    """src/coeff_form.rs:207-219: u64-valued coefficients, never the zero polynomial."""
    num = rng.randrange(min_coeffs, max_coeffs)
    coeffs = [0] * max_coeffs
    for i in range(num):
        coeffs[i] = rng.getrandbits(64)
    return kzg_amd.Polynomial(coeffs)


def test_basic_and_modify_single_coeff(engine):  # test_basic / test_modify_single_coeff
    rng = random.Random(69)
    tau = rng.getrandbits(64)
    params = kzg_amd.setup(engine, tau, 12)
    prover, verifier = kzg_amd.KZGProver(params), kzg_amd.KZGVerifier(params)
    p = random_polynomial(rng, 3, 12)
    c = prover.commit(p)
    assert c == C.g1_mul(G(), C.poly_eval(p.slice_coeffs(), tau))
    assert verifier.verify_poly(c, p)
    assert not verifier.verify_poly(c, random_polynomial(rng, 2, 12))
    mod = kzg_amd.Polynomial(list(p.coeffs))
    mod.coeffs[2] = (mod.coeffs[2] + 1) % M.R
    assert not verifier.verify_poly(c, mod)
    params.gs.free()


def test_eval_basic(engine):  # test_eval_basic incl. the degree-1 edge case (src/coeff_form.rs:317-342)
    rng = random.Random(70)
    tau = rng.getrandbits(64)
    params = kzg_amd.setup(engine, tau, 13)
    prover = kzg_amd.KZGProver(params)
    p = random_polynomial(rng, 5, 13)
    x = rng.getrandbits(64)
    y = p.eval(engine, x)
    assert y == C.poly_eval(p.slice_coeffs(), x)
    w = prover.create_witness(p, (x, y))
    ptau = C.poly_eval(p.slice_coeffs(), tau)
    assert w == C.g1_mul(G(), (ptau - y) * M.fr_inv(tau - x) % M.R)
    with pytest.raises(kzg_amd.PointNotOnPolynomial):
        prover.create_witness(p, (x, (y + 1) % M.R))
    # degree 1: p = 3 + X, opening (1, 4): quotient is the constant 1 -> gs[0] * 1
    p1 = kzg_amd.Polynomial([3, 1] + [0] * 11)
    assert p1.num_coeffs() == 2
    assert prover.create_witness(p1, (1, 4)) == G()
    with pytest.raises(kzg_amd.PointNotOnPolynomial):
        prover.create_witness(p1, (1, 5))
    # constant polynomial: (p - y) == 0 -> identity; otherwise error (src/polynomial.rs:194-199)
    p0 = kzg_amd.Polynomial([7])
    assert prover.create_witness(p0, (5, 7)) == bytes(96)
    with pytest.raises(kzg_amd.PointNotOnPolynomial):
        prover.create_witness(p0, (5, 8))
    params.gs.free()


@pytest.mark.parametrize("log_n", [10, 13])
def test_config1_commit_and_witness(engine, log_n):
    """BASELINE config 1 (2^10) and a two-pass-size case: full-width coefficients and opening point."""
    n = 1 << log_n
    rng = random.Random(1000 + log_n)
    tau = rng.getrandbits(64)
    params = kzg_amd.setup(engine, tau, n)
    prover = kzg_amd.KZGProver(params)
    coeffs = rand_scalars(rng, n)
    p = kzg_amd.Polynomial(coeffs)
    x = rng.randrange(M.R)
    y = C.poly_eval(coeffs, x)
    ptau = C.poly_eval(coeffs, tau)
    assert prover.commit(p) == C.g1_mul(G(), ptau)
    w = prover.create_witness(p, (x, y))
    assert w == C.g1_mul(G(), (ptau - y) * M.fr_inv(tau - x) % M.R)
    if log_n == 10:  # and against the oracle's long_division + Pippenger
        qb, nz = C.witness_quotient_bytes(C.scalars_to_bytes(coeffs), n, x, y)
        assert not nz and w == C.msm_g1_raw(params.gs.download(0, n - 1), qb, n - 1)
    params.gs.free()


@pytest.mark.parametrize("n,k", [(1 << 10, 21), (1 << 13, 5), (3, 4), (1, 3)])
def test_create_witness_many(engine, n, k):
    """kzg_witness_coeff_many == k calls of create_witness on the same polynomial (secondary reading):
    each witness against the known-tau identity and the single-opening entry point; wrong y -> that opening flagged as
    PointNotOnPolynomial while the others stay valid; device-resident coefficients give the same bytes."""
    rng = random.Random(7000 + n + k)
    tau = rng.getrandbits(64)
    params = kzg_amd.setup(engine, tau, max(n, 2))
    prover = kzg_amd.KZGProver(params)
    coeffs = rand_scalars(rng, n)
    p = kzg_amd.Polynomial(coeffs)
    ptau = C.poly_eval(coeffs, tau)
    xs = [rng.randrange(M.R) for _ in range(k)]
    pts = [(x, C.poly_eval(coeffs, x)) for x in xs]
    ws, ok = prover.create_witness_many(p, pts)
    assert ok == [True] * k
    for (x, y), w in zip(pts, ws):
        assert w == C.g1_mul(G(), (ptau - y) * M.fr_inv(tau - x) % M.R)
    assert ws[0] == prover.create_witness(p, pts[0])
    bad = list(pts)
    bad[1] = (bad[1][0], (bad[1][1] + 1) % M.R)
    ws2, ok2 = prover.create_witness_many(p, bad)
    assert ok2 == [j != 1 for j in range(k)] and [w for j, w in enumerate(ws2) if j != 1] == [w for j, w in enumerate(ws) if j != 1]
    e = engine
    rc = e.lib.kzg_witness_coeff_many(e.ctx, params.gs.handle, kzg_amd.api.pack_scalars(coeffs), n,
                                      kzg_amd.api.pack_scalars([b[0] for b in bad]), kzg_amd.api.pack_scalars([b[1] for b in bad]), k,
                                      L.FR_CANONICAL, 0, ctypes.create_string_buffer(96 * k), L.G1_AFFINE_MONT, None)
    assert rc == L.KZG_ERR_POINT_NOT_ON_POLY                      # no status array: the reference's Err, for the whole call
    buf = e.alloc_scalars(n)
    buf.upload(kzg_amd.api.pack_scalars(coeffs))
    ws3, ok3 = prover.create_witness_many(None, pts, coeffs_device=buf)
    assert ws3 == ws and ok3 == ok
    buf.free()
    params.gs.free()


@pytest.mark.parametrize("d", [8, 16, 1024])
def test_eval_form(engine, d):
    """test_basic / test_eval_basic / test_div_by_omega_i of src/eval_form.rs with the Lagrange SRS."""
    rng = random.Random(500 + d)
    tau = rng.getrandbits(64)
    params = kzg_amd.setup(engine, tau, d)
    lag = kzg_amd.setup_lagrange(engine, tau, d)
    if d <= 16:  # the reference's O(d^3) compute_lagrange_basis, literally, in the python model
        mp = M.KZGParams(M.setup_g1(tau, d))
        assert lag.download() == b"".join(M.g1_to_affine_mont(P) for P in M.compute_lagrange_basis_g1(mp))
    prover = kzg_amd.KZGProverEvalForm(params, lag)
    assert prover.degree() == d and prover.omega() == M.compute_omega(d)[2]
    coeffs = [rng.getrandbits(64) for _ in range(d)]
    evals = kzg_amd.EvaluationDomain.from_coeffs(coeffs)
    evals.fft(engine)
    assert evals.coeffs == C.fft(coeffs)
    c = prover.commit(evals)
    # config-3 identity: eval-form commit of NTT(p) == coeff-form commit of p == [p(tau)]G
    assert c == kzg_amd.KZGProver(params).commit(kzg_amd.Polynomial(coeffs))
    assert c == C.g1_mul(G(), C.poly_eval(coeffs, tau))
    assert kzg_amd.KZGVerifierEvalForm(params, lag).verify_poly(c, evals)
    other = kzg_amd.EvaluationDomain.from_coeffs([rng.getrandbits(64) for _ in range(d)])
    assert not kzg_amd.KZGVerifierEvalForm(params, lag).verify_poly(c, other)
    i = 3
    w = prover.create_witness(evals, i)
    xi = pow(prover.omega(), i, M.R)
    assert w == kzg_amd.KZGProver(params).create_witness(kzg_amd.Polynomial(coeffs), (xi, evals.coeffs[i]))
    assert w == C.g1_mul(G(), (C.poly_eval(coeffs, tau) - evals.coeffs[i]) * M.fr_inv(tau - xi) % M.R)
    with pytest.raises(kzg_amd.ReferencePanic):
        prover.create_witness(evals, d)
    short = kzg_amd.EvaluationDomain.from_coeffs(coeffs[: d // 2])
    with pytest.raises(kzg_amd.ReferencePanic):
        prover.commit(short)  # assert!(self.d == evals.d)
    assert prover.create_witness_all() == bytes(96)
    # throughput form: many openings of the same evaluation vector == one create_witness call each
    idx = [0, 3, d - 1, 3, 1] + [rng.randrange(d) for _ in range(14)]
    many = prover.create_witness_many(evals, idx)
    assert many[1] == w and many[3] == w
    for i2, w2 in zip(idx, many):
        x2 = pow(prover.omega(), i2, M.R)
        assert w2 == C.g1_mul(G(), (C.poly_eval(coeffs, tau) - evals.coeffs[i2]) * M.fr_inv(tau - x2) % M.R)
    with pytest.raises(kzg_amd.ReferencePanic):
        prover.create_witness_many(evals, [1, d])
    params.gs.free(); lag.free()


def _batched_oracle(coeffs, xs, ys, tau):
    """Reference semantics via the python model (sub-product tree + long_division) and known tau."""
    p = M.Polynomial(coeffs)
    tree = M.SubProductTree.new_from_points(xs)
    I = M.Polynomial.lagrange_interpolation_with_tree(xs, ys, tree)
    Z = 1
    for v in xs:
        Z = Z * (tau - v) % M.R
    w = C.g1_mul(G(), (p.eval(tau) - I.eval(tau)) * M.fr_inv(Z) % M.R)
    return I, w


@pytest.mark.parametrize("n,k", [(15, 8), (64, 2), (300, 7), (1024, 256), (5000, 33)])
def test_create_witness_batched(engine, n, k):
    """test_eval_batched (src/coeff_form.rs:344-375): r == interpolant, w == [(p - I)/Z]_1; other points fail."""
    rng = random.Random(600 + n + k)
    tau = rng.getrandbits(64)
    params = kzg_amd.setup(engine, tau, n)
    prover = kzg_amd.KZGProver(params)
    coeffs = [rng.getrandbits(64) for _ in range(n)] if n < 100 else rand_scalars(rng, n)
    p = kzg_amd.Polynomial(coeffs)
    xs = rand_scalars(rng, k)
    ys = [C.poly_eval(coeffs, x) for x in xs]
    wit = prover.create_witness_batched(p, xs, ys)
    I, w = _batched_oracle(coeffs, xs, ys, tau)
    assert wit.polynomial().coeffs == I.coeffs[:k] and wit.polynomial().degree == k - 1
    assert wit.elem() == w
    if n <= 300:  # the reference's own path end to end: long_division by the tree product + MSM
        num = M.Polynomial(coeffs).sub_ref(I)
        psi, rem = num.long_division(M.SubProductTree.new_from_points(xs).product)
        assert rem is None
        assert wit.elem() == C.msm_g1(params.gs.download(0, psi.num_coeffs()), psi.slice_coeffs())
    bad = list(ys)
    bad[k // 2] = (bad[k // 2] + 1) % M.R
    with pytest.raises(kzg_amd.PointNotOnPolynomial):
        prover.create_witness_batched(p, xs, bad)
    params.gs.free()


def test_create_witness_batched_edge_cases(engine):
    rng = random.Random(77)
    tau = rng.getrandbits(64)
    n = 14
    params = kzg_amd.setup(engine, tau, 15)
    prover = kzg_amd.KZGProver(params)
    coeffs = [rng.getrandbits(64) for _ in range(n)]
    p = kzg_amd.Polynomial(coeffs)
    # test_eval_batched_all_points (src/coeff_form.rs:377-397): k == num_coeffs -> zero quotient -> identity
    xs = [rng.getrandbits(64) for _ in range(n)]
    ys = [C.poly_eval(coeffs, x) for x in xs]
    wit = prover.create_witness_batched(p, xs, ys)
    assert wit.elem() == bytes(96)
    assert wit.polynomial().coeffs == coeffs           # the interpolant is p itself
    ys2 = list(ys); ys2[3] = (ys2[3] + 5) % M.R
    with pytest.raises(kzg_amd.PointNotOnPolynomial):
        prover.create_witness_batched(p, xs, ys2)
    # k == 1 quirk (src/polynomial.rs:244-247): r = X + (y - x), w = [(p - r)/(X - x)]
    x = rng.getrandbits(64)
    y = C.poly_eval(coeffs, x)
    wit = prover.create_witness_batched(p, [x], [y])
    assert wit.polynomial().coeffs == [(y - x) % M.R, 1] and wit.polynomial().degree == 1
    Ir, w = _batched_oracle(coeffs, [x], [y], tau)
    assert Ir.coeffs == [(y - x) % M.R, 1] and wit.elem() == w
    with pytest.raises(kzg_amd.PointNotOnPolynomial):
        prover.create_witness_batched(p, [x], [(y + 1) % M.R])
    # duplicate opening points: the reference panics (invert of zero)
    with pytest.raises(kzg_amd.ReferencePanic):
        prover.create_witness_batched(p, [5, 5, 6], [C.poly_eval(coeffs, 5)] * 2 + [C.poly_eval(coeffs, 6)])
    # opening points ON the cosets 7*H, 7^2*H the (p - I)/Z division would evaluate on (7 = the coset shift itself):
    # the engine must move to a coset without roots of Z; results equal the oracle's schoolbook division
    w16 = pow(M.FR_ROOT_OF_UNITY, 1 << (M.FR_S - 4), M.R)
    # (round 4: the first choice is the plain domain H itself -- no scaling passes -- so openings AT roots of unity, the usual case
    # of an evaluation-form protocol, are the ones that move on: x = 1, w^3, both together with 7 and 49 -> 7^3 * H)
    for xs in ([5, 6, 7], [7, 49, 7 * w16 % M.R], [49 * w16 % M.R, 343, 11], [1, 5, 6], [pow(w16, 3, M.R), 2, 3], [1, 7, w16, 49 * w16 % M.R],
               [w16, pow(w16, 2, M.R), pow(w16, 15, M.R)]):
        ys = [C.poly_eval(coeffs, x) for x in xs]
        wit = prover.create_witness_batched(p, xs, ys)
        Ir, w = _batched_oracle(coeffs, xs, ys, tau)
        assert wit.elem() == w and wit.polynomial().coeffs == Ir.coeffs, xs
    params.gs.free()


def test_create_witness_batched_openings_at_roots_of_unity_2_17(engine):
    """Openings at points of the evaluation domain itself (x_i = w^(m_i), what an evaluation-form protocol opens at): Z vanishes on
    H, so the division moves to the coset 7 * H; a second polynomial opened at random points stays on H.  Both against the known-tau
    identity with oracle-evaluated p(tau), I(tau); two-pass transforms, Z through the short-input path."""
    n, k = 1 << 17, 40
    tau = 0x1234567
    params = kzg_amd.setup(engine, tau, n, g2_len=0)
    prover = kzg_amd.KZGProver(params)
    buf = engine.alloc_scalars(n).fill_random(1717)
    raw = buf.download()
    coeffs = kzg_amd.unpack_scalars(raw)
    _, _, omega = kzg_amd.compute_omega(n)
    rng = random.Random(1718)
    G = C.g1_generator()
    ptau = C.poly_eval_bytes(raw, n, tau)
    for xs in ([pow(omega, rng.randrange(n), M.R) for _ in range(k)], rand_scalars(rng, k)):
        xs = list(dict.fromkeys(xs))
        ys = [C.poly_eval_bytes(raw, n, x) for x in xs]
        wit = prover.create_witness_batched(kzg_amd.Polynomial(coeffs), xs, ys)
        I = list(wit.polynomial().coeffs)
        assert all(C.poly_eval(I, x) == y for x, y in zip(xs, ys))
        Z = 1
        for x in xs:
            Z = Z * (tau - x) % M.R
        assert wit.elem() == C.g1_mul(G, (ptau - C.poly_eval(I, tau)) * pow(Z, -1, M.R) % M.R)
    buf.free()
    params.gs.free()


@pytest.mark.parametrize("log_n", [0, 3, 9, 13])
def test_coset_fft_roundtrip_and_values(engine, log_n):
    """coset_fft / icoset_fft (src/ft.rs:168-178) vs the python model; fft_composition round trips."""
    rng = random.Random(700 + log_n)
    xs = rand_scalars(rng, 1 << log_n)
    got = engine.coset_ntt(xs, log_n)
    if log_n <= 9:
        e = M.EvaluationDomain.from_coeffs(xs)
        e.coset_fft()
        assert got == e.coeffs
        e.icoset_fft()
        assert e.coeffs == xs
    assert engine.coset_ntt(got, log_n, inverse=True) == xs
    assert engine.coset_ntt(engine.coset_ntt(xs, log_n, inverse=True), log_n) == xs


@pytest.mark.parametrize("d", [1, 2, 4, 8, 16, 64, 1 << 10, 1 << 16])
def test_compute_lagrange_basis_from_monomial(engine, d):
    """compute_lagrange_basis (src/eval_form.rs:254-280) from the monomial SRS alone: the G1 inverse group-NTT (gfft.hip) gives
    the same group elements as the closed form with a known secret, and as the reference's literal O(d^3) construction."""
    tau = 0xABCDEF0123
    params = kzg_amd.setup(engine, tau, d)
    lag = kzg_amd.compute_lagrange_basis(params)
    want = kzg_amd.setup_lagrange(engine, tau, d)
    assert lag.download() == want.download()
    if d <= 16:
        mp = M.KZGParams(M.setup_g1(tau, d))
        assert lag.download() == b"".join(M.g1_to_affine_mont(P) for P in M.compute_lagrange_basis_g1(mp))
    odd = kzg_amd.setup(engine, tau, 6)
    with pytest.raises(kzg_amd.ReferencePanic):
        kzg_amd.compute_lagrange_basis(odd)
    params.gs.free(); lag.free(); want.free(); odd.gs.free()


def test_cpp_host_mirror(engine, tmp_path):
    """include/kzg_mi355x.hpp: compile the C++ mirror test against the shared library and run it (the single-GPU surface; the device
    group part runs in tests/test_gpu_mgpu.py)."""
    import subprocess
    from tests.gpu_common import build_cpp_mirror
    out = subprocess.run([build_cpp_mirror(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "cpp mirror ok" in out.stdout, (out.returncode, out.stdout, out.stderr)


def test_create_witness_batched_point_set_cache():
    """create_witness_batched keeps what depends on the opening POINTS alone (Z, weights, coset shift, 1 / Z on the coset) per point set
    (witness.hip, PointSetCache).  In a context of its own with TWO slots: many polynomials at one point set (hits) give the oracle's
    [(p(tau) - I(tau)) / Z(tau)]G each; a wrong value is still refused on a hit; a point set with duplicates fails every time and is
    never kept; a third point set evicts the least recently used one; results with the cache switched off are identical."""
    import ctypes
    tau = 0x5EED1234
    n, k = 1 << 12, 16
    rng = random.Random(4242)
    G = C.g1_generator()

    def want(coeffs, xs, ys, icoef):
        z = 1
        for x in xs:
            z = z * (tau - x) % M.R
        return C.g1_mul(G, (C.poly_eval(coeffs, tau) - C.poly_eval(icoef, tau)) * pow(z, -1, M.R) % M.R)

    def stats(e):
        h, m = ctypes.c_uint64(), ctypes.c_double()
        assert e.lib.kzg_prof_get(e.ctx, b"point_set_cache", ctypes.byref(h), ctypes.byref(m)) == 0
        return h.value, int(m.value)

    results = {}
    for slots in (2, 0):
        e = kzg_amd.Engine(0)
        e.set_option("witness_cache_slots", slots)
        params = kzg_amd.setup(e, tau, n, g2_len=0)
        prover = kzg_amd.KZGProver(params)
        rng = random.Random(4242)
        sets = [[rng.randrange(M.R) for _ in range(k)] for _ in range(3)]
        got = []
        for rnd in range(3):                      # A A A B A C B ... : hits, a miss, an eviction
            for xs in (sets[0], sets[0], sets[1], sets[0], sets[2], sets[1]):
                coeffs = [rng.randrange(M.R) for _ in range(n)]
                ys = [C.poly_eval(coeffs, x) for x in xs]
                wit = prover.create_witness_batched(kzg_amd.Polynomial(coeffs), xs, ys)
                icoef = wit.r.slice_coeffs()
                assert all(C.poly_eval(icoef, x) == y for x, y in zip(xs, ys))
                assert wit.w == want(coeffs, xs, ys, icoef)
                got.append(wit.w)
        # a wrong value on a cached point set
        coeffs = [rng.randrange(M.R) for _ in range(n)]
        ys = [C.poly_eval(coeffs, x) for x in sets[0]]
        ys[3] = (ys[3] + 1) % M.R
        with pytest.raises(kzg_amd.PointNotOnPolynomial):
            prover.create_witness_batched(kzg_amd.Polynomial(coeffs), sets[0], ys)
        # duplicates: refused twice (the failing point set is not kept), and the good ones still work afterwards
        dup = list(sets[1])
        dup[5] = dup[2]
        for _ in range(2):
            with pytest.raises(kzg_amd.ReferencePanic):
                prover.create_witness_batched(kzg_amd.Polynomial(coeffs), dup, [C.poly_eval(coeffs, x) for x in dup])
        ys = [C.poly_eval(coeffs, x) for x in sets[1]]
        wit = prover.create_witness_batched(kzg_amd.Polynomial(coeffs), sets[1], ys)
        assert wit.w == want(coeffs, sets[1], ys, wit.r.slice_coeffs())
        hits, misses = stats(e)
        if slots:
            assert hits >= 6 and misses >= 3, (hits, misses)      # A twice in a row per round at least; B / C alternate through two slots
        else:
            assert hits == 0
        results[slots] = got
        params.gs.free()
        e.close()
    assert results[2] == results[0]
    # a context that opened small polynomials first: the pool grows when a larger point set arrives, and the larger set is then hit
    e = kzg_amd.Engine(0)
    params = kzg_amd.setup(e, tau, 1 << 14, g2_len=0)
    prover = kzg_amd.KZGProver(params)
    for n2, k2 in ((1 << 10, 4), (1 << 14, 8), (1 << 14, 8), (1 << 10, 4)):
        xs = [rng.randrange(M.R) for _ in range(k2)] if n2 == 1 << 10 else list(range(100, 100 + k2))
        coeffs = [rng.randrange(M.R) for _ in range(n2)]
        ys = [C.poly_eval(coeffs, x) for x in xs]
        wit = prover.create_witness_batched(kzg_amd.Polynomial(coeffs), xs, ys)
        assert wit.w == want(coeffs, xs, ys, wit.r.slice_coeffs())
    hits, _ = stats(e)
    assert hits >= 1            # the second 2^14 opening found the entry the grown pool holds
    params.gs.free()
    e.close()
