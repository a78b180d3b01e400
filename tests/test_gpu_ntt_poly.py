"""NTT (src/ft.rs) and Fr-polynomial kernels vs the oracle, through the C ABI."""
import random

import pytest

import kzg_amd
from kzg_amd import _lib as L
from oracle import c_oracle as C
from oracle import kzg_model as M
from tests.gpu_common import rand_scalars

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log_n", list(range(0, 15)))
def test_ntt_matches_serial_fft(engine, log_n):
    """EvaluationDomain::fft == serial_fft restatement, natural order in/out, every size 2^0..2^14
    (covers the single-tile path <= 2^12 and the two-pass path)."""
    rng = random.Random(200 + log_n)
    xs = rand_scalars(rng, 1 << log_n)
    got = engine.ntt(xs, log_n)
    assert got == C.fft(xs)
    assert engine.ntt(got, log_n, inverse=True) == xs            # fft_composition (src/ft.rs:447-479)
    assert engine.ntt(engine.ntt(xs, log_n, inverse=True), log_n) == xs


def test_ntt_small_vs_python_model(engine):
    rng = random.Random(31)
    for log_n in (1, 3, 6):
        xs = rand_scalars(rng, 1 << log_n)
        e = M.EvaluationDomain.from_coeffs(xs)
        e.fft()
        assert engine.ntt(xs, log_n) == e.coeffs
        e.ifft()
        assert e.coeffs == xs


def test_ntt_montgomery_and_device_resident(engine):
    rng = random.Random(32)
    log_n = 13
    xs = rand_scalars(rng, 1 << log_n)
    buf = engine.alloc_scalars(1 << log_n, sfmt=L.FR_MONT)
    buf.upload(b"".join(M.fr_to_mont_le(x) for x in xs))
    engine.ntt(buf, log_n)
    rinv = pow(M.FR_MONT_R, -1, M.R)
    got = [v * rinv % M.R for v in kzg_amd.unpack_scalars(buf.download())]
    assert got == C.fft(xs)
    buf.free()


def test_ntt_linearity_2_20(engine):
    """Full BASELINE size: NTT(a + c*b) == NTT(a) + c*NTT(b) on sampled outputs, plus round trip and
    agreement with direct evaluation p(w^i) at a few points (size-independent properties)."""
    log_n = 20
    n = 1 << log_n
    a = engine.alloc_scalars(n).fill_random(1)
    b = engine.alloc_scalars(n).fill_random(2)
    a0 = a.download()
    _, _, omega = kzg_amd.compute_omega(n)
    idx = [0, 1, 5, 1023, 1024, 65537, n - 1]
    want = {i: C.poly_eval_bytes(a0, n, pow(omega, i, M.R)) for i in idx}
    engine.ntt(a, log_n)
    fa = a.download()
    for i in idx:
        assert int.from_bytes(fa[32 * i:32 * i + 32], "little") == want[i]
    engine.ntt(a, log_n, inverse=True)
    assert a.download() == a0
    a.free(); b.free()


def test_compute_omega():
    assert kzg_amd.compute_omega(1 << 20) == M.compute_omega(1 << 20)
    assert kzg_amd.compute_omega(10) == M.compute_omega(10)
    assert kzg_amd.compute_omega(1) == M.compute_omega(1)
    with pytest.raises(kzg_amd.PolynomialDegreeTooLarge):
        kzg_amd.compute_omega((1 << 31) + 1)


@pytest.mark.parametrize("n", [1, 2, 3, 8, 9, 2047, 2048, 2049, 5000, 70000])
def test_poly_eval_and_linear_quotient(engine, n):
    """Polynomial::eval and (p - y)/(X - x) by long_division (src/polynomial.rs:156-165,193-227)."""
    rng = random.Random(300 + n)
    coeffs = rand_scalars(rng, n)
    x = rng.randrange(M.R)
    y = C.poly_eval(coeffs, x)
    assert engine.poly_eval(coeffs, x) == y
    if n >= 2:
        q = engine.quotient_linear(coeffs, x, y)
        qb, nz = C.witness_quotient_bytes(C.scalars_to_bytes(coeffs), n, x, y)
        assert not nz and q == C.bytes_to_scalars(qb)
        with pytest.raises(kzg_amd.PointNotOnPolynomial):
            engine.quotient_linear(coeffs, x, (y + 1) % M.R)


def test_reference_literal_vectors(engine):
    """The reference's own known-answer tests, run on the GPU path:
    test_eval_basic (src/polynomial.rs:579-597) and the exact division of test_long_division (:558-576)."""
    p = [34, 0, 7, 4, 0, 1]
    assert engine.poly_eval(p, 0) == 34 and engine.poly_eval(p, 1) == 46 and engine.poly_eval(p, 5) == 3834
    # x^3 + 6x^2 + 13x + 10 / x + 2 = x^2 + 4x + 5 r 0   (divisor X - (-2))
    assert engine.quotient_linear([10, 13, 6, 1], M.R - 2, 0) == [5, 4, 1]
    # x^3 + 2x^2 - 3x + 4 / x - 7 = x^2 + 9x + 60 r 424 : p(7) = 424
    assert engine.poly_eval([4, M.R - 3, 2, 1], 7) == 424
    assert engine.quotient_linear([4, M.R - 3, 2, 1], 7, 424) == [60, 9, 1]
    # 3x^4 - 5x^2 + 3 / x + 2 = 3x^3 - 6x^2 + 7x - 14 r 31
    assert engine.quotient_linear([3, 0, M.R - 5, 0, 3], M.R - 2, 31) == [M.R - 14, 7, M.R - 6, 3]


@pytest.mark.parametrize("log_d,m", [(0, 0), (1, 1), (4, 3), (4, 0), (8, 255), (12, 1000)])
def test_div_by_omega_i(engine, log_d, m):
    """div_by_omega_i (src/eval_form.rs:58-84) applied to evals - evals[m]."""
    d = 1 << log_d
    rng = random.Random(400 + log_d)
    ev = rand_scalars(rng, d)
    got = engine.quotient_eval(ev, m)
    num = [(v - ev[m]) % M.R for v in ev]
    want = C.bytes_to_scalars(C.div_by_omega_i_bytes(C.scalars_to_bytes(num), d, m))
    assert got == want
    if log_d == 4:  # the literal O(d) inversions version of the reference
        e = M.EvaluationDomain.from_coeffs(num)
        assert got == M.div_by_omega_i(e, m).coeffs


def test_fill_random_definition(engine):
    buf = engine.alloc_scalars(1000).fill_random(42)
    got = kzg_amd.unpack_scalars(buf.download())
    assert got == [kzg_amd.splitmix_scalar(42, i) for i in range(1000)]
    buf.fill_random(7, u64_valued=True)
    got = kzg_amd.unpack_scalars(buf.download())
    assert got == [kzg_amd.splitmix_scalar(7, i, True) for i in range(1000)] and max(got) < 1 << 64
    buf.free()


def test_fft_mul_polynomial_arith(engine):
    """polynomial_arith (src/ft.rs:411-434): fft_mul == naive Mul for sizes {1,5,10,50}^2, plus a 2^12 x 2^12 case
    and the sub-product-tree root of 256 opening points (two degree-128 factors -> best_mul takes the NTT path)."""
    rng = random.Random(42)
    for ca in (1, 5, 10, 50):
        for cb in (1, 5, 10, 50):
            a, b = rand_scalars(rng, ca), rand_scalars(rng, cb)
            want = M.Polynomial(a, ca - 1).mul_naive(M.Polynomial(b, cb - 1))
            assert engine.poly_mul(a, b) == want.coeffs
    a, b = rand_scalars(rng, 4096), rand_scalars(rng, 3000)
    got = engine.poly_mul(a, b)
    x = rng.randrange(M.R)
    assert C.poly_eval(got, x) == C.poly_eval(a, x) * C.poly_eval(b, x) % M.R and len(got) == 7095
    xs = rand_scalars(rng, 256)
    left = M.SubProductTree.new_from_points(xs[:128]).product
    right = M.SubProductTree.new_from_points(xs[128:]).product
    pa, pb = kzg_amd.Polynomial(left.coeffs), kzg_amd.Polynomial(right.coeffs)
    got, want = pa.best_mul(engine, pb), left.fft_mul(right)   # the reference keeps the zero-padded 2^k vector;
    assert got.degree == want.degree == 256                     # PartialEq = degree + zipped prefix (:29-40)
    assert got.slice_coeffs() == want.slice_coeffs()


def test_fft_mul_short_operand_takes_the_short_input_transform(engine):
    """A low-degree factor times a long polynomial (the shape of Z = prod (X - x_i) against a degree-2^20 numerator): the short
    operand's transform skips the column pass and reads nothing beyond its coefficients (k_ntt_pass2<SHORT>, ntt.hip).  Product
    against the model's naive Mul at 2^13..2^15 (both table kinds of the inter-pass twiddle are used below 2^21), by oracle
    evaluation at 2^17 and at 2^22 (two-level twiddles), operand lengths at and around the first-row limit 2^floor(log_n / 2)."""
    rng = random.Random(77)
    for na, nb in ((1, 5000), (2, 8000), (63, 8100), (64, 8100), (65, 8100), (90, 16000), (128, 30000), (181, 32500)):
        a, b = rand_scalars(rng, na), rand_scalars(rng, nb)
        want = M.Polynomial(a, na - 1).mul_naive(M.Polynomial(b, nb - 1))
        assert engine.poly_mul(a, b) == want.coeffs, (na, nb)
        assert engine.poly_mul(b, a) == want.coeffs, (nb, na)
    for na, nb in ((257, (1 << 17) - 300), (2048, (1 << 22) - 3000)):
        a = rand_scalars(rng, na)
        bb = engine.alloc_scalars(nb).fill_random(600 + na)
        braw = bb.download()
        ab = engine.alloc_scalars(na)
        ab.upload(kzg_amd.pack_scalars(a))
        out = engine.alloc_scalars(na + nb - 1)
        rc = engine.lib.kzg_poly_mul(engine.ctx, ab.ptr, na, bb.ptr, nb, ab.sfmt, L.IN_DEVICE | L.OUT_DEVICE, out.ptr)
        assert rc == 0, engine.last_error()
        got = out.download()
        for _ in range(3):
            x = rng.randrange(M.R)
            assert C.poly_eval_bytes(got, na + nb - 1, x) == C.poly_eval(a, x) * C.poly_eval_bytes(braw, nb, x) % M.R
        for buf in (bb, ab, out):
            buf.free()


def test_evaluation_domain_remaining_ops(engine):
    """EvaluationDomain::z / divide_by_z_on_coset / mul_assign / sub_assign (src/ft.rs:180-271) against the model, and the
    workflow they exist for: with a = b * Z_H + c on the domain H, (coset values of a - c) / Z on the coset are b's coset values."""
    rng = random.Random(31)
    for log_n in (0, 3, 8, 13):
        n = 1 << log_n
        xs, ys = [rng.randrange(M.R) for _ in range(n)], [rng.randrange(M.R) for _ in range(n)]
        for op in ("mul_assign", "sub_assign"):
            e, o = kzg_amd.EvaluationDomain.from_coeffs(xs), kzg_amd.EvaluationDomain.from_coeffs(ys)
            m, mo = M.EvaluationDomain.from_coeffs(xs), M.EvaluationDomain.from_coeffs(ys)
            getattr(e, op)(engine, o)
            getattr(m, op)(mo)
            assert e.coeffs == m.coeffs, (op, log_n)
        e, m = kzg_amd.EvaluationDomain.from_coeffs(xs), M.EvaluationDomain.from_coeffs(xs)
        e.divide_by_z_on_coset(engine)
        m.divide_by_z_on_coset()
        assert e.coeffs == m.coeffs
        tau = rng.randrange(M.R)
        assert e.z(tau) == m.z(tau) == (pow(tau, n, M.R) - 1) % M.R
    with pytest.raises(kzg_amd.ReferencePanic):     # assert_eq!(self.coeffs.len(), other.coeffs.len())
        kzg_amd.EvaluationDomain.from_coeffs([1, 2]).mul_assign(engine, kzg_amd.EvaluationDomain.from_coeffs([1, 2, 3]))
    # quotient by the vanishing polynomial of H through the coset: a(X) = b(X) (X^d - 1) + c(X), deg b, deg c < d
    d = 64
    b = [rng.randrange(M.R) for _ in range(d)]
    c = [rng.randrange(M.R) for _ in range(d)]
    a = [(c[i] - b[i]) % M.R for i in range(d)] + b            # b X^d - b + c
    A = kzg_amd.EvaluationDomain.from_coeffs(a)                 # size 2d
    Cc = kzg_amd.EvaluationDomain.from_coeffs(c + [0] * d)
    A.coset_fft(engine)
    Cc.coset_fft(engine)
    A.sub_assign(engine, Cc)
    # on the coset of the size-2d domain, X^d - 1 takes the values g^d w^(jd) - 1 = +-g^d - 1: divide pointwise
    g, w = 7, kzg_amd.compute_omega(2 * d)[2]
    A.coeffs = [v * pow((pow(g, d, M.R) * pow(w, j * d, M.R) - 1) % M.R, -1, M.R) % M.R for j, v in enumerate(A.coeffs)]
    A.icoset_fft(engine)
    assert A.coeffs == b + [0] * d
    # device-resident vectors keep their form (Montgomery in, Montgomery out)
    n = 1 << 10
    xs, ys = [rng.randrange(M.R) for _ in range(n)], [rng.randrange(M.R) for _ in range(n)]
    Rm = (1 << 256) % M.R
    da = engine.alloc_scalars(n, kzg_amd.FR_MONT).upload(kzg_amd.pack_scalars([x * Rm % M.R for x in xs]))
    db = engine.alloc_scalars(n, kzg_amd.FR_MONT).upload(kzg_amd.pack_scalars([y * Rm % M.R for y in ys]))
    rc = engine.lib.kzg_fr_vec_mul(engine.ctx, da.ptr, db.ptr, n, kzg_amd.FR_MONT, kzg_amd.IN_DEVICE)
    assert rc == 0
    assert kzg_amd.unpack_scalars(da.download()) == [x * y % M.R * Rm % M.R for x, y in zip(xs, ys)]
    da.free(); db.free()


@pytest.mark.gpu
@pytest.mark.limit(300)
@pytest.mark.parametrize("n", [1, 2, 7, 2047, 2048, 2049, 100003, 150001, (1 << 18) + 5, 1 << 20, (1 << 21) + 1, (1 << 22) + 12345])
def test_horner_scan_every_shape(engine, n):
    """kzg_poly_eval and kzg_witness_coeff through the Horner scan kernels of poly.hip (LDS-staged tiles of 2048 coefficients, block
    carries scanned by one block with host-computed step multipliers) at sizes that exercise every shape: below one tile, ragged last
    tiles, every width of the carry scan (64 to 1024 threads), one block carry per scan thread (<= 2^21) and several (above): p(x) by the oracle's Horner loop, the witness against
    [(p(tau) - y) / (tau - x)]G (src/polynomial.rs:193-227, src/coeff_form.rs:66-81)."""
    import ctypes
    from oracle import c_oracle as C
    R = kzg_amd.api.R_MODULUS
    tau = 0x5EED5EED
    buf = engine.alloc_scalars(n).fill_random(900 + n % 97)
    host = buf.download()
    x = kzg_amd.splitmix_scalar(31, n % 1000)
    y = engine.poly_eval(buf, x)
    assert y == C.poly_eval_bytes(host, n, x)
    if n >= 2:
        params = kzg_amd.setup(engine, tau, n, g2_len=0)
        out = ctypes.create_string_buffer(96)
        rc = engine.lib.kzg_witness_coeff(engine.ctx, params.gs.handle, buf.ptr, n, (x % R).to_bytes(32, "little"), (y % R).to_bytes(32, "little"),
                                          buf.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
        assert rc == 0, engine.last_error()
        ptau = C.poly_eval_bytes(host, n, tau)
        assert out.raw == C.g1_mul(C.g1_generator(), (ptau - y) * pow(tau - x, -1, R) % R)
        rc = engine.lib.kzg_witness_coeff(engine.ctx, params.gs.handle, buf.ptr, n, (x % R).to_bytes(32, "little"), ((y + 1) % R).to_bytes(32, "little"),
                                          buf.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
        assert rc == L.KZG_ERR_POINT_NOT_ON_POLY
        params.gs.free()
    buf.free()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2048, 70001, (1 << 18) + 3])
def test_horner_extreme_values(engine, n):
    """The Horner kernels keep unreduced sums of Shoup products in 29-bit limbs (poly.hip: below 25 r before a value is reduced): the
    largest operands -- every coefficient r - 1, x = r - 1 (and x = 1, x = 0, all-zero coefficients) -- against the oracle's serial
    loop, in both scalar formats (src/polynomial.rs:156-165, 193-227)."""
    big = (M.R - 1).to_bytes(32, "little")
    rng = random.Random(n)
    cases = [("max", big * n), ("zero", bytes(32 * n)),
             ("mixed", b"".join(big if rng.random() < 0.5 else bytes(32) for _ in range(n)))]
    for name, blob in cases:
        coeffs = C.bytes_to_scalars(blob)
        for x in (M.R - 1, 1, 0, M.R - 2, 2):
            y = C.poly_eval_bytes(blob, n, x)
            assert engine.poly_eval(coeffs, x) == y, (name, x)
            q = engine.quotient_linear(coeffs, x, y)
            qb, nz = C.witness_quotient_bytes(blob, n, x, y)
            assert not nz and q == C.bytes_to_scalars(qb), (name, x)
    # Montgomery-form scalars resident on the device take the same kernels (the product with the plain x keeps the form)
    buf = engine.alloc_scalars(n, sfmt=L.FR_MONT).upload(b"".join(((M.R - 1) * (1 << 256) % M.R).to_bytes(32, "little") for _ in range(n)))
    for x in (M.R - 1, 3):
        assert engine.poly_eval(buf, x) == C.poly_eval_bytes(big * n, n, x)
    buf.free()
