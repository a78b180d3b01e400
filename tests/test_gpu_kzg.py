"""End-to-end KZGProver / KZGProverEvalForm parity via the known-tau identities and the C
oracle, mirroring the reference's tests (src/coeff_form.rs:271-398, src/eval_form.rs:318-483)."""
import random

import pytest

import kzg_amd
from kzg_amd import _lib as L
from oracle import c_oracle as C
from oracle import kzg_model as M
from tests.gpu_common import engine, rand_scalars  # noqa: F401

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
    params.gs.free(); lag.free()
