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


# Compressed encodings of [1]G, [2]G, [3]G on BLS12-381 G1 as they circulate in public test material (the Ethereum consensus
# "interop" validator public keys of the secret keys 1, 2 and 3; the first one is the generator of the zkcrypto / IETF
# pairing-friendly-curves documents).  [upstream-memory]: written down from memory of that public material, not read from a file in
# this image -- the reference crate holds no literal G1 value and blstrs is absent.  They are the only values in this
# repository that were not produced by the oracle's own author: 48 bytes each that an implementation with a wrong doubling,
# addition, Montgomery constant, sign convention or serialisation rule would not reproduce.
PUBLISHED_G1 = {
    1: "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb",
    2: "a572cbea904d67468808c8eb50a9450c9721db309128012543902d0ac358a62ae28f75bb8f1c7c42c39a8c5529bf0f4e",
    3: "89ece308f9d1f0131765212deca99697b112d61f9be9a5f1f3780a51335b3ff981747a0b2ca2179b96d2c0c9024e5224",
}


def test_g1_layer_against_published_points():
    """The oracle's G1 law and zcash serialisation (python model and C oracle) reproduce the published [k]G; so do the commitment
    of the constant polynomial k and the sums G + G, 2G + G -- doubling, addition and scalar multiplication each pinned."""
    from oracle import c_oracle as C
    G = M.G1
    for k, hexv in PUBLISHED_G1.items():
        want = bytes.fromhex(hexv)
        assert M.g1_to_compressed(M.g1_mul(G, k)) == want
        assert M.g1_to_compressed(C.blob_to_point(C.g1_mul(C.g1_generator(), k))) == want
        assert M.g1_to_compressed(C.blob_to_point(C.msm_g1(C.setup_g1(12345, 1), [k]))) == want   # commit of the constant k
    assert M.g1_to_compressed(M.g1_add(G, G)) == bytes.fromhex(PUBLISHED_G1[2])
    assert M.g1_to_compressed(M.g1_add(M.g1_mul(G, 2), G)) == bytes.fromhex(PUBLISHED_G1[3])
    # decompression of the published bytes gives points on the curve and in the subgroup that re-encode to themselves
    for k, hexv in PUBLISHED_G1.items():
        Pk = M.g1_from_compressed(bytes.fromhex(hexv))
        assert M.g1_is_on_curve(Pk) and M.g1_to_compressed(Pk) == bytes.fromhex(hexv)
        assert M.g1_mul(Pk, M.R) is None   # the identity: in the r-torsion subgroup


# The same for G2: the compressed generator as published (zkcrypto / IETF pairing-friendly-curves, Ethereum), in full, and the
# leading 8 bytes of [2]G2 as they appear in public BLS12-381 precompile test material.  [upstream-memory], as above.
PUBLISHED_G2_GENERATOR = ("93e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
                          "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8")
PUBLISHED_2G2_PREFIX = "aa4edef9c1ed7f72"


def test_g2_layer_against_published_points():
    from oracle import pairing_model as PM
    assert PM.g2_to_compressed(PM.G2).hex() == PUBLISHED_G2_GENERATOR
    assert PM.g2_to_compressed(PM.g2_mul(PM.G2, 2)).hex().startswith(PUBLISHED_2G2_PREFIX)
    assert PM.g2_to_compressed(PM.g2_add(PM.G2, PM.G2)).hex().startswith(PUBLISHED_2G2_PREFIX)
    P = PM.g2_from_compressed(bytes.fromhex(PUBLISHED_G2_GENERATOR))
    assert PM.g2_is_on_curve(P) and P == PM.G2


# Constants of the public BLS12-381 implementations in their IN-MEMORY (Montgomery) form, [upstream-memory] like the points above:
# zkcrypto bls12_381 `Scalar` / `Fp` (R = 2^256 resp. 2^384, little-endian u64 limbs -- the radix and limb order blst's blst_fr /
# blst_fp use) and `G1Affine::generator()`; the 2^k-th roots of unity as c-kzg-4844 tabulates them (SCALE2_ROOT_OF_UNITY[k]).
# They pin what the zero-copy formats KZG_FR_MONT_LE_32 and KZG_G1_AFFINE_MONT_96 claim to be.
def _limbs(*w):
    return b"".join(int(x).to_bytes(8, "little") for x in w)


PUBLISHED_FR_ONE_MONT = _limbs(0x00000001fffffffe, 0x5884b7fa00034802, 0x998c4fefecbc4ff5, 0x1824b159acc5056f)      # Scalar R
PUBLISHED_FR_R2 = _limbs(0xc999e990f3f29c6d, 0x2b6cedcb87925c23, 0x05d314967254398f, 0x0748d9d99f59ff11)            # Scalar R2
PUBLISHED_FR_ROOT_OF_UNITY_MONT = _limbs(0xb9b58d8c5f0e466a, 0x5b1b4c801819d7ec, 0x0af53ae352a31e64, 0x5bf3adda19e9b27b)
PUBLISHED_FQ_ONE_MONT = _limbs(0x760900000002fffd, 0xebf4000bc40c0002, 0x5f48985753c758ba, 0x77ce585370525745,
                               0x5c071a97a256ec6d, 0x15f65ec3fa80e493)                                                # Fp R
PUBLISHED_G1_GENERATOR_MONT = (_limbs(0x5cb38790fd530c16, 0x7817fc679976fff5, 0x154f95c7143ba1c1, 0xf0ae6acdf3d0e747,
                                      0xedce6ecc21dbf440, 0x120177419e0bfb75) +
                               _limbs(0xbaac93d50ce72271, 0x8c22631a7918fd8e, 0xdd595f13570725ce, 0x51ac582950405194,
                                      0x0e1c8c3fad0059c0, 0x0bbc3efc5008a26a))
PUBLISHED_ROOTS_OF_UNITY = {    # 2^k-th roots, canonical
    2: 0x0000000000000000_8d51ccce760304d0_ec03000276030000_0001000000000000,
    3: 0x345766f603fa66e7_8c0625cd70d77ce2_b38b21c28713b700_7228fd3397743f7a,
    32: 0x16a2a19edfe81f20_d09b681922c813b4_b63683508c2280b9_3829971f439f0d2b,
}
PUBLISHED_MONT_INV = {"fr": 0xfffffffeffffffff, "fq": 0x89f3fffcfffcfffd}    # -p^-1 mod 2^64


def test_montgomery_layouts_against_published_constants():
    from oracle import c_oracle as C
    assert M.fr_to_mont_le(1) == PUBLISHED_FR_ONE_MONT
    assert M.fr_to_mont_le(1 << 256) == PUBLISHED_FR_R2                     # R2 = 2^512 mod r is the Montgomery form of 2^256
    assert M.fr_to_mont_le(M.FR_ROOT_OF_UNITY) == PUBLISHED_FR_ROOT_OF_UNITY_MONT
    assert ((1 << 384) % M.Q).to_bytes(48, "little") == PUBLISHED_FQ_ONE_MONT
    assert M.g1_to_affine_mont(M.G1) == PUBLISHED_G1_GENERATOR_MONT
    assert C.g1_generator() == PUBLISHED_G1_GENERATOR_MONT                  # the C oracle's 96-byte affine-Montgomery blob
    assert M.FR_ROOT_OF_UNITY == PUBLISHED_ROOTS_OF_UNITY[32]
    for k, w in PUBLISHED_ROOTS_OF_UNITY.items():
        assert pow(M.FR_ROOT_OF_UNITY, 1 << (32 - k), M.R) == w
        if k < 32:
            assert M.compute_omega(1 << k)[2] == w         # src/ft.rs:55-76
    assert (-pow(M.R, -1, 1 << 64)) % (1 << 64) == PUBLISHED_MONT_INV["fr"]
    assert (-pow(M.Q, -1, 1 << 64)) % (1 << 64) == PUBLISHED_MONT_INV["fq"]
