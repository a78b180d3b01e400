"""NTT parity above 2^14 (src/ft.rs:111-178): bit-exact against the C oracle's serial_fft restatement at EVERY size 2^15..2^22
and at 2^24 -- forward, inverse, and the coset transforms -- so that the four-step kernels' inter-pass twiddle table (n <= 2^21)
and the two-level twiddle branch taken for log_n > 21 (ntt.hip) are both compared element for element, plus direct evaluations
p(w^i) by the oracle's Horner loop as a check that does not go through any FFT."""
import ctypes

import pytest

import kzg_amd
from kzg_amd import _lib as L
from oracle import c_oracle as C
from oracle import kzg_model as M

pytestmark = pytest.mark.gpu
R = M.R


def _coset_scale(blob, n, g):
    """distribute_powers (src/ft.rs:142-160 as used by coset_fft): element i times g^i, on canonical LE bytes"""
    out, pw = bytearray(len(blob)), 1
    for i in range(n):
        out[32 * i:32 * i + 32] = (int.from_bytes(blob[32 * i:32 * i + 32], "little") * pw % R).to_bytes(32, "little")
        pw = pw * g % R
    return bytes(out)


@pytest.mark.limit(300)
@pytest.mark.parametrize("log_n", [15, 16, 17, 18, 19, 20, 21, 22, 23])   # 23: the smallest size that takes three passes (ntt_run3)
def test_ntt_bit_exact_vs_oracle(engine, log_n):
    n = 1 << log_n
    buf = engine.alloc_scalars(n).fill_random(500 + log_n)
    a0 = buf.download()
    want = C.fft_bytes(a0, log_n)
    engine.ntt(buf, log_n)
    assert buf.download() == want                      # EvaluationDomain::fft
    engine.ntt(buf, log_n, inverse=True)
    assert buf.download() == a0                        # fft_composition (src/ft.rs:447-479)
    engine.ntt(buf, log_n, inverse=True)
    assert buf.download() == C.fft_bytes(a0, log_n, inverse=True)   # EvaluationDomain::ifft on its own
    # two direct evaluations by the oracle's Horner loop: element i of the transform is p(w^i)
    _, _, omega = kzg_amd.compute_omega(n)
    for i in (1, n - 3):
        assert int.from_bytes(want[32 * i:32 * i + 32], "little") == C.poly_eval_bytes(a0, n, pow(omega, i, R))
    buf.free()


@pytest.mark.limit(300)
@pytest.mark.parametrize("opts", [{"ntt_kernel": 0, "ntt_three_from": 0}, {"ntt_kernel": 1, "ntt_three_from": 0}, {"ntt_kernel": 2, "ntt_three_from": 0},
                                  {"ntt_kernel": 1, "ntt_three_from": 20}, {"ntt_kernel": 2, "ntt_three_from": 22}],
                         ids=["r5-kernels", "fused", "fused-two-butterflies", "three-pass-from-2^20", "three-pass-two-butterflies"])
def test_ntt_kernel_variants_bit_exact_vs_oracle(engine, opts):
    """Every selectable form of the pass kernels (option ntt_kernel: round 5's three-phase passes, the fused k_ntt_tile, two butterflies
    per thread) and of the decomposition (two passes / three passes of <= 2^8 points) against the oracle's serial_fft, forward and
    inverse: even and odd sub-transform lengths, the full inter-pass table (<= 2^21) and the two-level product, 2^13 (the smallest
    two-pass size) up to 2^22; the defaults are restored afterwards."""
    defaults = {"ntt_kernel": 1, "ntt_three_from": 23}
    try:
        for k, v in opts.items():
            engine.set_option(k, v)
        for log_n in (13, 14, 17, 20, 21, 22):
            n = 1 << log_n
            buf = engine.alloc_scalars(n).fill_random(700 + log_n)
            a0 = buf.download()
            engine.ntt(buf, log_n)
            assert buf.download() == C.fft_bytes(a0, log_n), (opts, log_n)
            engine.ntt(buf, log_n, inverse=True)
            assert buf.download() == a0, (opts, log_n)
            buf.free()
    finally:
        for k, v in defaults.items():
            engine.set_option(k, v)


@pytest.mark.limit(300)
@pytest.mark.parametrize("log_n", [13, 16, 20, 23])
def test_ntt_extreme_values_bit_exact_vs_oracle(engine, log_n):
    """Inputs at the ends of the lazy arithmetic's ranges -- all r - 1, r - 1 alternating with 0 and with 1, a lone r - 1 among zeros,
    in canonical and in Montgomery form (R mod r and r - 1 as raw words): every limb chain of the butterflies sees its largest operands.
    Forward and inverse against the oracle's serial_fft (two-pass sizes, and 2^23: three passes)."""
    n = 1 << log_n
    big = (R - 1).to_bytes(32, "little")
    zero, one = bytes(32), (1).to_bytes(32, "little")
    patterns = [big * n, (big + zero) * (n // 2), (one + big) * (n // 2), zero * (n - 1) + big, big + zero * (n - 1)]
    buf = engine.alloc_scalars(n)
    for k, blob in enumerate(patterns[:3] if log_n >= 23 else patterns):
        buf.upload(blob)
        engine.ntt(buf, log_n)
        assert buf.download() == C.fft_bytes(blob, log_n), (log_n, k)
        buf.upload(blob)
        engine.ntt(buf, log_n, inverse=True)
        assert buf.download() == C.fft_bytes(blob, log_n, inverse=True), (log_n, k)
    buf.free()


@pytest.mark.parametrize("log_n", [15, 18, 20, 22])
def test_coset_ntt_bit_exact_vs_oracle(engine, log_n):
    """coset_fft = distribute_powers(7) then fft; icoset_fft = ifft then distribute_powers(7^-1) (src/ft.rs:142-178)."""
    n = 1 << log_n
    buf = engine.alloc_scalars(n).fill_random(600 + log_n)
    a0 = buf.download()
    rc = engine.lib.kzg_coset_ntt_fr(engine.ctx, buf.ptr, log_n, 0, buf.sfmt, L.IN_DEVICE)
    assert rc == 0, engine.last_error()
    assert buf.download() == C.fft_bytes(_coset_scale(a0, n, 7), log_n)
    rc = engine.lib.kzg_coset_ntt_fr(engine.ctx, buf.ptr, log_n, 1, buf.sfmt, L.IN_DEVICE)
    assert rc == 0 and buf.download() == a0
    buf.upload(a0)
    rc = engine.lib.kzg_coset_ntt_fr(engine.ctx, buf.ptr, log_n, 1, buf.sfmt, L.IN_DEVICE)
    assert rc == 0 and buf.download() == _coset_scale(C.fft_bytes(a0, log_n, inverse=True), n, pow(7, -1, R))
    buf.free()


def test_ntt_2_24_bit_exact_and_sampled(engine):
    """The largest size the ABI advertises: bit-exact against the oracle's FFT over all 2^24 outputs (about half a minute of CPU),
    a dozen outputs also against direct Horner evaluation (no FFT involved), round trip, Montgomery-form input."""
    log_n = 24
    n = 1 << log_n
    buf = engine.alloc_scalars(n).fill_random(2424)
    a0 = buf.download()
    engine.ntt(buf, log_n)
    got = buf.download()
    want = C.fft_bytes(a0, log_n)
    assert got == want
    _, _, omega = kzg_amd.compute_omega(n)
    idx = [0, 1, 2, 4095, 4096, 4097, (1 << 12) * 4095 + 1, n // 2, n // 2 + 1, n - 1] + [kzg_amd.splitmix_scalar(9, j) % n for j in range(2)]
    for i in idx:
        assert int.from_bytes(got[32 * i:32 * i + 32], "little") == C.poly_eval_bytes(a0, n, pow(omega, i, R)), i
    engine.ntt(buf, log_n, inverse=True)
    assert buf.download() == a0
    del got
    # Montgomery-form data through the same kernels (linear map with Montgomery-form twiddles)
    rc = engine.lib.kzg_fill_random_fr(engine.ctx, buf.ptr, n, 2424, 0, L.FR_MONT)
    assert rc == 0
    mbuf = kzg_amd.DeviceBuffer.__new__(kzg_amd.DeviceBuffer)
    mbuf.engine, mbuf.n, mbuf.sfmt, mbuf.ptr = engine, n, L.FR_MONT, buf.ptr
    engine.ntt(mbuf, log_n)
    head = mbuf.download(4)
    rinv = pow(M.FR_MONT_R, -1, R)
    for i in range(4):
        assert int.from_bytes(head[32 * i:32 * i + 32], "little") * rinv % R == int.from_bytes(want[32 * i:32 * i + 32], "little")
    buf.free()


def test_verify_poly_eval_2_22(engine):
    """KZGVerifierEvalForm::verify_poly (src/eval_form.rs:162-171) above 2^21: ifft of the evaluations (two-level twiddles) then
    the monomial MSM equals the coefficient-form commitment; a perturbed evaluation is rejected."""
    log_n = 22
    n = 1 << log_n
    tau = 0x5EED5EED5EED5EED
    params = kzg_amd.setup(engine, tau, n, g2_len=0)
    buf = engine.alloc_scalars(n).fill_random(2222)
    a0 = buf.download()
    out = ctypes.create_string_buffer(96)
    rc = engine.lib.kzg_commit_coeff(engine.ctx, params.gs.handle, buf.ptr, n, buf.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
    assert rc == 0 and out.raw == C.g1_mul(C.g1_generator(), C.poly_eval_bytes(a0, n, tau))
    engine.ntt(buf, log_n)
    ok = ctypes.c_int()
    rc = engine.lib.kzg_verify_poly_eval(engine.ctx, params.gs.handle, out.raw, L.G1_AFFINE_MONT, buf.ptr, n, buf.sfmt, L.IN_DEVICE, ctypes.byref(ok))
    assert rc == 0 and ok.value == 1
    ev = bytearray(buf.download(1, offset=12345))
    ev[0] ^= 1
    v = kzg_amd.DeviceBuffer.__new__(kzg_amd.DeviceBuffer)
    v.engine, v.n, v.sfmt, v.ptr = engine, 1, buf.sfmt, ctypes.c_void_p(buf.ptr.value + 32 * 12345)
    v.upload(bytes(ev))
    rc = engine.lib.kzg_verify_poly_eval(engine.ctx, params.gs.handle, out.raw, L.G1_AFFINE_MONT, buf.ptr, n, buf.sfmt, L.IN_DEVICE, ctypes.byref(ok))
    assert rc == 0 and ok.value == 0
    buf.free()
    params.gs.free()



@pytest.mark.parametrize("log_n,full", [(25, True), (26, False), (28, False)])
def test_ntt_above_2_24(engine, log_n, full):
    """EvaluationDomain::fft beyond the two-pass range (the reference goes to 2^31, src/ft.rs:55-76): one more four-step level
    (A = 2, 4, 16 columns of 2^24) around the two-pass transform.  2^25 bit-exact against the oracle's FFT; every size: outputs
    against direct Horner evaluation (indices with k mod A != 0 and k / A large: both index components), round trip, and the
    inverse on its own at a sampled index through the forward identity ifft(x)[i] = fft(x)[(n - i) mod n] / n."""
    n = 1 << log_n
    buf = engine.alloc_scalars(n).fill_random(700 + log_n)
    a0 = buf.download()
    engine.ntt(buf, log_n)
    _, _, omega = kzg_amd.compute_omega(n)
    idx = [1, (n >> 1) + 3, n - 2] if log_n < 28 else [n - 5]
    got = {i: int.from_bytes(buf.download(1, offset=i), "little") for i in idx}
    if full:
        assert buf.download() == C.fft_bytes(a0, log_n)
    for i in idx:
        assert got[i] == C.poly_eval_bytes(a0, n, pow(omega, i, R)), i
    engine.ntt(buf, log_n, inverse=True)
    assert buf.download() == a0                       # fft_composition
    engine.ntt(buf, log_n, inverse=True)              # ifft(x)[i] = fft(x)[(n - i) mod n] / n
    i = idx[0]
    inv_n = pow(n, -1, R)
    assert int.from_bytes(buf.download(1, offset=(n - i) % n), "little") == got[i] * inv_n % R
    buf.free()


def test_coset_ntt_and_fft_mul_above_2_24(engine):
    """coset_fft / icoset_fft (src/ft.rs:168-178) and fft_mul (src/polynomial.rs:167-183) beyond 2^24 (the reference's bound is
    2^31; here 2^28 / 2^27 since round 4): at 2^25, coset outputs against direct Horner evaluation at 7 w^i by the oracle, the
    round trip, and a product of two 2^24-coefficient polynomials checked by oracle evaluation."""
    log_n = 25
    n = 1 << log_n
    buf = engine.alloc_scalars(n).fill_random(925)
    a0 = buf.download()
    assert engine.lib.kzg_coset_ntt_fr(engine.ctx, buf.ptr, log_n, 0, buf.sfmt, L.IN_DEVICE) == 0, engine.last_error()
    _, _, omega = kzg_amd.compute_omega(n)
    for i in (3, (n >> 1) + 7):
        assert int.from_bytes(buf.download(1, offset=i), "little") == C.poly_eval_bytes(a0, n, 7 * pow(omega, i, R) % R), i
    assert engine.lib.kzg_coset_ntt_fr(engine.ctx, buf.ptr, log_n, 1, buf.sfmt, L.IN_DEVICE) == 0, engine.last_error()
    assert buf.download() == a0
    # (p * q)(x) == p(x) q(x): p = the first 2^24 coefficients, q = the rest minus a few (the product has 2^25 - 9 coefficients)
    na, nb = 1 << 24, (1 << 24) - 8
    out = engine.alloc_scalars(na + nb - 1)
    pa = kzg_amd.DeviceBuffer.__new__(kzg_amd.DeviceBuffer)
    pa.engine, pa.n, pa.sfmt, pa.ptr = engine, na, buf.sfmt, buf.ptr
    pb = kzg_amd.DeviceBuffer.__new__(kzg_amd.DeviceBuffer)
    pb.engine, pb.n, pb.sfmt, pb.ptr = engine, nb, buf.sfmt, ctypes.c_void_p(buf.ptr.value + 32 * na)
    rc = engine.lib.kzg_poly_mul(engine.ctx, pa.ptr, na, pb.ptr, nb, buf.sfmt, L.IN_DEVICE | L.OUT_DEVICE, out.ptr)
    assert rc == 0, engine.last_error()
    prod = out.download()
    x = kzg_amd.splitmix_scalar(926, 0)
    assert C.poly_eval_bytes(prod, na + nb - 1, x) == C.poly_eval_bytes(a0[:32 * na], na, x) * C.poly_eval_bytes(a0[32 * na:32 * (na + nb)], nb, x) % R
    out.free()
    buf.free()


def test_create_witness_batched_2_25(engine):
    """create_witness_batched (src/coeff_form.rs:83-111) above the old 2^24 bound: degree 2^25 - 1, 64 opening points; the division's
    transforms take the extra four-step level (ntt_run_large).  w == [(p(tau) - I(tau)) / Z(tau)]G with p(tau) by the oracle's Horner
    loop on the downloaded coefficients; the opening values by the engine, two of them against the oracle (the identity holds only if
    all are right); a wrong value is refused."""
    TAU = 0x5EED5EED5EED5EED
    n, k = 1 << 25, 64
    params = kzg_amd.setup(engine, TAU, n, g2_len=0)
    buf = engine.alloc_scalars(n).fill_random(2525)
    raw = buf.download()
    xs = [kzg_amd.splitmix_scalar(2526, i) for i in range(k)]
    ys = [engine.poly_eval(buf, v) for v in xs]
    assert all(C.poly_eval_bytes(raw, n, xs[i]) == ys[i] for i in (0, 63))
    out = ctypes.create_string_buffer(96)
    rbuf, rlen = ctypes.create_string_buffer(32 * k), ctypes.c_size_t()
    rc = engine.lib.kzg_witness_coeff_batched(engine.ctx, params.gs.handle, buf.ptr, n, kzg_amd.pack_scalars(xs), kzg_amd.pack_scalars(ys), k,
                                              buf.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT, rbuf, ctypes.byref(rlen))
    assert rc == 0 and rlen.value == k, engine.last_error()
    I = kzg_amd.unpack_scalars(rbuf.raw)
    assert all(C.poly_eval(I, xs[i]) == ys[i] for i in range(0, k, 9))
    Z = 1
    for v in xs:
        Z = Z * (TAU - v) % R
    ptau = C.poly_eval_bytes(raw, n, TAU)
    assert out.raw == C.g1_mul(C.g1_generator(), (ptau - C.poly_eval(I, TAU)) * pow(Z, -1, R) % R)
    ys[5] = (ys[5] + 1) % R
    rc = engine.lib.kzg_witness_coeff_batched(engine.ctx, params.gs.handle, buf.ptr, n, kzg_amd.pack_scalars(xs), kzg_amd.pack_scalars(ys), k,
                                              buf.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT, rbuf, ctypes.byref(rlen))
    assert rc == L.KZG_ERR_POINT_NOT_ON_POLY
    buf.free()
    params.gs.free()
