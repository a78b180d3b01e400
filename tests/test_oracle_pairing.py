"""Pins the verifier-side oracle (oracle/pairing_model.py).  The reference holds no literal G2 / Gt values (its
pairing and G2 group are third-party), so the pin is mathematical -- group laws, bilinearity, non-degeneracy, the
final-exponentiation chain against plain square-and-multiply -- plus the reference's own verifier scenarios
(src/coeff_form.rs:279-400) restated with a known tau."""
import random

from oracle import kzg_model as M, pairing_model as P


def test_selfcheck_and_final_exponentiation_chain():
    assert P.selfcheck(full=True)


def test_bilinearity_and_inverses():
    rng = random.Random(11)
    e = P.pairing(M.G1, P.G2)
    for _ in range(2):
        a, b = rng.randrange(M.R), rng.randrange(M.R)
        assert P.pairing(M.g1_mul(M.G1, a), P.g2_mul(P.G2, b)) == P.f12_pow(e, a * b % M.R)
    assert P.pairing(M.g1_neg(M.G1), P.G2) == P.f12_inv(e) == P.pairing(M.G1, P.g2_neg(P.G2))
    assert P.pairing(None, P.G2) == P.F12_ONE == P.pairing(M.G1, None)
    # additivity in each argument
    a, b = 5, 9
    lhs = P.pairing(M.g1_add(M.g1_mul(M.G1, a), M.g1_mul(M.G1, b)), P.G2)
    assert lhs == P.f12_mul(P.pairing(M.g1_mul(M.G1, a), P.G2), P.pairing(M.g1_mul(M.G1, b), P.G2))


def test_g2_encodings_and_group_law():
    rng = random.Random(12)
    pts = [P.g2_mul(P.G2, rng.randrange(M.R)) for _ in range(3)] + [None]
    for p in pts:
        assert P.g2_is_on_curve(p)
        assert P.g2_from_compressed(P.g2_to_compressed(p)) == p
        assert P.g2_from_uncompressed(P.g2_to_uncompressed(p)) == p
    a, b, c = pts[:3]
    assert P.g2_add(P.g2_add(a, b), c) == P.g2_add(a, P.g2_add(b, c))
    assert P.g2_add(a, P.g2_neg(a)) is None
    assert P.g2_multi_exp([a, b], [3, M.R - 1]) == P.g2_add(P.g2_mul(a, 3), P.g2_neg(b))
    assert P.setup_g2(7, 4) == [P.g2_mul(P.G2, 7 ** i) for i in range(4)]


def random_polynomial(rng, min_coeffs, max_coeffs):  # src/coeff_form.rs:207-219
    num = rng.randrange(min_coeffs, max_coeffs)
    coeffs = [rng.getrandbits(64) if i < num else 0 for i in range(max_coeffs)]
    p = M.Polynomial.new_from_coeffs(coeffs, num - 1)
    p.shrink_degree()
    return p


def test_reference_verifier_scenarios():
    rng = random.Random(69)
    tau = rng.getrandbits(64)
    params = P.setup(tau, 15)
    prover, verifier = M.KZGProver(params), P.KZGVerifier(params)
    # test_eval_basic (:317-342)
    p = random_polynomial(rng, 5, 13)
    c = prover.commit(p)
    x = rng.getrandbits(64)
    y = p.eval(x)
    w = prover.create_witness(p, (x, y))
    assert verifier.verify_poly(c, p)
    assert verifier.verify_eval((x, y), c, w)
    assert not verifier.verify_eval((x, (y + 1) % M.R), c, w)
    p1 = M.Polynomial([3, 1] + [0] * 11)
    c1 = prover.commit(p1)
    w1 = prover.create_witness(p1, (1, 4))
    assert verifier.verify_eval((1, 4), c1, w1) and not verifier.verify_eval((1, 5), c1, w1)
    # test_eval_batched (:344-376)
    p = random_polynomial(rng, 8, 15)
    c = prover.commit(p)
    xs = [rng.getrandbits(64) for _ in range(8)]
    r, wb = prover.create_witness_batched(p, xs, [p.eval(v) for v in xs])
    assert verifier.verify_eval_batched(xs, c, wb, r)
    assert not verifier.verify_eval_batched([rng.getrandbits(64) for _ in range(8)], c, wb, r)


def test_eval_form_verifier_scenarios():
    rng = random.Random(70)
    tau, d = rng.getrandbits(64), 8
    params = P.setup(tau, d)
    lag_g, lag_h = M.lagrange_basis_g1_known_tau(tau, d), P.lagrange_basis_g2_known_tau(tau, d)
    # the closed form equals the reference's definition sum_j l_ij [tau^j]H (src/eval_form.rs:254-280) -- via the pairing:
    # e(G, L_i(tau) H) = e(L_i(tau) G, H)
    assert P.pairing(M.G1, lag_h[3]) == P.pairing(lag_g[3], P.G2)
    prover = M.KZGProverEvalForm(params, lag_g)
    verifier = P.KZGVerifierEvalForm(params, lag_g, lag_h)
    evals = M.EvaluationDomain.from_coeffs([rng.getrandbits(64) for _ in range(d)])
    evals.fft()
    c = prover.commit(evals)
    w = prover.create_witness(evals, 2)
    assert verifier.verify_eval((2, evals.coeffs[2]), c, w)
    assert not verifier.verify_eval((2, (evals.coeffs[2] + 1) % M.R), c, w)
