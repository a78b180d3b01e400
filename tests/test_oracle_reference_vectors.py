"""Pins the oracle (python model) against every literal known-answer test the reference holds for the
hot path: src/polynomial.rs:494-690 and the property tests of src/ft.rs:411-479."""
import random

from oracle import kzg_model as M

R = M.R


def P(coeffs):
    return M.Polynomial([c % R for c in coeffs])


def test_selfcheck_constants():
    assert M.selfcheck()


def test_long_division_literals():  # src/polynomial.rs:497-577
    q, r = P([3, 0, -5, 0, 3]).long_division(P([2, 1, 0, 0, 0]))
    assert r is not None and r == P([31, 0, 0, 0, 0])
    assert q == P([-14, 7, -6, 3, 0])
    q, r = P([4, -3, 2, 1]).long_division(P([-7, 1, 0, 0]))
    assert r is not None and r == P([424, 0, 0, 0])
    assert q == P([60, 9, 1, 0])
    q, r = P([10, 13, 6, 1]).long_division(P([2, 1, 0, 0]))
    assert r is None
    assert q == P([5, 4, 1, 0])


def test_eval_basic_literals():  # src/polynomial.rs:579-597
    p = P([34, 0, 7, 4, 0, 1])
    assert p.eval(0) == 34 and p.eval(1) == 46 and p.eval(5) == 3834


def test_new_subproduct_tree():  # src/polynomial.rs:599-637
    def verify(tree):
        if tree.left is not None and tree.right is not None:
            assert tree.product == tree.left.product.best_mul(tree.right.product)
            verify(tree.left)
            verify(tree.right)
    verify(M.SubProductTree.new_from_points([2, 5, 7, 90, 111, 31, 29]))
    verify(M.SubProductTree.new_from_points([2, 5, 7, 90, 111]))


def test_fast_multi_eval():  # src/polynomial.rs:639-664
    p = P([2, 5, 7, 90, 111])
    xs = list(range(1, 9))
    assert p.multi_eval(xs) == [p.eval(x) for x in xs]


def test_interpolation():  # src/polynomial.rs:666-690
    I = M.Polynomial.lagrange_interpolation([2], [8])
    assert I.eval(2) == 8 and I.coeffs == [6, 1] and I.degree == 1   # the X + (y - x) quirk
    xs, ys = [2, 5, 7, 90, 111, 31, 29], [8, 1, 43, 2, 87, 122, 13]
    I = M.Polynomial.lagrange_interpolation(xs, ys)
    assert [I.eval(x) for x in xs] == ys


def test_polynomial_arith_fft_mul_equals_naive():  # src/ft.rs:411-434
    rng = random.Random(42)
    for ca in (1, 5, 10, 50):
        for cb in (1, 5, 10, 50):
            a = M.Polynomial([rng.randrange(R) for _ in range(ca)], ca - 1)
            b = M.Polynomial([rng.randrange(R) for _ in range(cb)], cb - 1)
            assert a.mul_naive(b) == a.fft_mul(b)


def test_fft_composition():  # src/ft.rs:447-479
    rng = random.Random(1)
    for k in range(0, 8):
        v = [rng.randrange(R) for _ in range(1 << k)]
        d = M.EvaluationDomain.from_coeffs(v)
        d.ifft(); d.fft()
        assert d.coeffs == v
        d.fft(); d.ifft()
        assert d.coeffs == v
        d.icoset_fft(); d.coset_fft()
        assert d.coeffs == v
        d.coset_fft(); d.icoset_fft()
        assert d.coeffs == v


def test_fft_is_evaluation_at_powers_of_omega():
    rng = random.Random(2)
    v = [rng.randrange(R) for _ in range(16)]
    d = M.EvaluationDomain.from_coeffs(v)
    p = M.Polynomial(v)
    d.fft()
    assert d.coeffs == [p.eval(pow(d.omega, i, R)) for i in range(16)]


def test_div_by_omega_i_matches_long_division():  # src/eval_form.rs:318-339
    rng = random.Random(69)
    d, exp, omega = M.compute_omega(10)
    w3 = pow(omega, 3, R)
    top = M.Polynomial([rng.getrandbits(64) for _ in range(d)])
    y = top.eval(w3)
    top.coeffs[0] = (top.coeffs[0] - y) % R
    naive, rem = top.long_division(P([-w3, 1]))
    assert rem is None
    e = M.EvaluationDomain(top.coeffs, d, exp, omega)
    e.fft()
    for f in (M.div_by_omega_i, M.div_by_omega_i_fast):
        s = f(e, 3)
        s.ifft()
        assert s.to_polynomial() == naive


def test_known_tau_identities_small():  # replaces the pairing checks of the reference's tests
    rng = random.Random(3)
    tau = rng.getrandbits(64)
    params = M.setup(tau, 16)
    assert params.gs == M.setup_g1(tau, 16)           # chain gs[i] = gs[i-1]*s == [s^i]G
    prover = M.KZGProver(params)
    p = M.Polynomial([rng.getrandbits(64) for _ in range(13)])
    assert prover.commit(p) == M.g1_mul(M.G1, p.eval(tau))
    x = rng.getrandbits(64); y = p.eval(x)
    assert prover.create_witness(p, (x, y)) == M.g1_mul(M.G1, (p.eval(tau) - y) * M.fr_inv(tau - x) % R)
    try:
        prover.create_witness(p, (x, y + 1))
        assert False
    except M.PointNotOnPolynomial:
        pass
    xs = [rng.getrandbits(64) for _ in range(8)]
    ys = [p.eval(v) for v in xs]
    I, w = prover.create_witness_batched(p, xs, ys)
    Z = 1
    for v in xs:
        Z = Z * (tau - v) % R
    assert w == M.g1_mul(M.G1, (p.eval(tau) - I.eval(tau)) * M.fr_inv(Z) % R)
    pe = M.KZGParams(M.setup_g1(tau, 8))
    lag = M.compute_lagrange_basis_g1(pe)
    assert lag == M.lagrange_basis_g1_known_tau(tau, 8)
