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


def test_golden_verify_fixture_matches_oracle_and_host_tower():
    """tests/golden/verify.json regenerates from the model, and the C++ tower (host build of the HIP source) agrees on
    every pairing verdict in it."""
    import ctypes
    import os
    import subprocess
    import tempfile
    from tests import golden_util as GU
    g = GU.load("verify.json")
    tau = GU.sc(g["tau"])
    hs = P.setup_g2(tau, 6)
    assert [P.g2_to_compressed(h).hex() for h in hs] == g["hs_compressed"]
    assert P.g2_to_compressed(P.g2_multi_exp(hs, [GU.sc(h) for h in g["msm_g2"]["scalars"]])).hex() == g["msm_g2"]["result"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        so = os.path.join(td, "libhosttower.so")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(root, "tests", "host_tower.cpp")])
        lib = ctypes.CDLL(so)
        for c in g["pairing_checks"]:
            g1 = [M.g1_from_compressed(bytes.fromhex(h)) for h in c["g1"]]
            g2 = [P.g2_from_compressed(bytes.fromhex(h)) for h in c["g2"]]
            assert P.pairing_product_is_one(list(zip(g1, g2))) == c["is_one"]
            b1 = b"".join(bytes(96) if p is None else p[0].to_bytes(48, "little") + p[1].to_bytes(48, "little") for p in g1)
            b2 = b"".join(bytes(192) if q is None else b"".join(v.to_bytes(48, "little") for v in (q[0][0], q[0][1], q[1][0], q[1][1]))
                          for q in g2)
            assert bool(lib.ht_pairing_product_is_one(b1, b2, 2)) == c["is_one"]
