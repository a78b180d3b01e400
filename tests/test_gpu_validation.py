"""Input validation at the C ABI: points that blstrs' G1Affine / G2Affine deserialisation would reject upstream
-- non-canonical limbs, off the curve, outside the r-torsion subgroup -- are rejected with KZG_ERR_BAD_POINT in EVERY format
(the zero-copy Montgomery ones included), scalars >= r are taken mod r, and the documented size limits return KZG_ERR_SHAPE."""
import ctypes
import random

import pytest

import kzg_amd
from kzg_amd import _lib as L
from oracle import c_oracle as C
from oracle import kzg_model as M
from oracle import pairing_model as PM
from tests.gpu_common import rand_scalars

pytestmark = pytest.mark.gpu
TAU = 0x77AA55CC33


def _non_subgroup_g1():
    """A point of E(Fq) outside G1: almost every curve point is (the cofactor is ~2^126)."""
    x = 5
    while True:
        rhs = (x * x * x + 4) % M.Q
        y = pow(rhs, (M.Q + 1) // 4, M.Q)
        if y * y % M.Q == rhs:
            P = (x, y)
            acc, base, k = None, P, M.R      # [r]P with plain additions (g1_mul reduces its scalar mod r)
            while k:
                if k & 1:
                    acc = M.g1_add(acc, base)
                base = M.g1_add(base, base)
                k >>= 1
            if acc is not None:
                return P
        x += 1


def _non_subgroup_g2():
    x = (3, 1)
    while True:
        rhs = PM.f2_add(PM.f2_mul(PM.f2_sqr(x), x), (4, 4))
        y = PM.f2_sqrt(rhs)
        if y is not None:
            P = (x, y)
            acc, base, k = None, P, M.R
            while k:
                if k & 1:
                    acc = PM.g2_add(acc, base)
                base = PM.g2_add(base, base)
                k >>= 1
            if acc is not None:
                return P
        x = (x[0] + 1, x[1])


def _upload_rc(engine, blob, n, pfmt):
    h = ctypes.c_void_p()
    rc = engine.lib.kzg_srs_upload_g1(engine.ctx, blob, n, pfmt, ctypes.byref(h))
    if rc == 0:
        engine.lib.kzg_srs_free(engine.ctx, h)
    return rc


def test_g1_upload_rejects_invalid_points_in_every_format(engine):
    good = C.setup_g1(TAU, 4)
    assert _upload_rc(engine, good, 4, L.G1_AFFINE_MONT) == 0
    Rq = M.FQ_MONT_R
    gx, gy = C.blob_to_point(good[:96])
    mont = lambda v: (v * Rq % M.Q).to_bytes(48, "little")  # noqa: E731
    cases = {
        "off_curve": mont(gx) + mont((gy + 1) % M.Q),
        "non_canonical_limbs": (gx * Rq % M.Q + M.Q).to_bytes(48, "little") + mont(gy),
    }
    T = _non_subgroup_g1()
    assert M.g1_is_on_curve(T)
    cases["non_subgroup"] = mont(T[0]) + mont(T[1])
    for name, bad in cases.items():
        blob = good[:96] + bad + good[192:]
        assert _upload_rc(engine, blob, 4, L.G1_AFFINE_MONT) == L.KZG_ERR_BAD_POINT, name
        # the same point as a Jacobian (Z = 1) blob
        jac = b"".join(good[96 * i: 96 * i + 96] + mont(1) for i in (0,)) + bad + mont(1)
        assert _upload_rc(engine, jac, 2, L.G1_JACOBIAN_MONT) == L.KZG_ERR_BAD_POINT, name
    # canonical wire formats: on the curve but outside the subgroup
    unc = M.g1_to_uncompressed(T)
    cmp_ = M.g1_to_compressed(T)
    assert _upload_rc(engine, unc, 1, L.G1_ZCASH_UNCOMPRESSED) == L.KZG_ERR_BAD_POINT
    assert _upload_rc(engine, cmp_, 1, L.G1_ZCASH_COMPRESSED) == L.KZG_ERR_BAD_POINT
    # a caller that vouches for its points (option trusted_points) skips the subgroup test, never the on-curve test
    engine.set_option("trusted_points", 1)
    try:
        assert _upload_rc(engine, good[:96] + cases["non_subgroup"], 2, L.G1_AFFINE_MONT) == 0
        assert _upload_rc(engine, good[:96] + cases["off_curve"], 2, L.G1_AFFINE_MONT) == L.KZG_ERR_BAD_POINT
    finally:
        engine.set_option("trusted_points", 0)
    # a Jacobian point off the curve with Z != 1
    z = 7
    jac = mont(gx * z * z % M.Q) + mont((gy * z * z * z + 1) % M.Q) + mont(z)
    assert _upload_rc(engine, jac, 1, L.G1_JACOBIAN_MONT) == L.KZG_ERR_BAD_POINT
    jac = mont(gx * z * z % M.Q) + mont(gy * z * z * z % M.Q) + mont(z)
    assert _upload_rc(engine, jac, 1, L.G1_JACOBIAN_MONT) == 0


def test_verifier_rejects_non_subgroup_witness(engine):
    """verify_eval's rewritten pairing equation is only sound for r-torsion points: W + T (T of cofactor order) must not be
    accepted silently -- it is rejected as a bad point, as G1Affine deserialisation would upstream."""
    rng = random.Random(3)
    n = 64
    params = kzg_amd.setup(engine, TAU, n, g2_len=2)
    coeffs = rand_scalars(rng, n)
    poly = kzg_amd.Polynomial(coeffs)
    prover, verifier = kzg_amd.KZGProver(params), kzg_amd.KZGVerifier(params)
    cm = prover.commit(poly)
    x = rng.randrange(M.R)
    y = C.poly_eval(coeffs, x)
    w = prover.create_witness(poly, (x, y))
    assert verifier.verify_eval((x, y), cm, w)
    T = _non_subgroup_g1()
    bad_w = M.g1_to_affine_mont(M.g1_add(C.blob_to_point(w), T))
    with pytest.raises(kzg_amd.EngineError):
        verifier.verify_eval((x, y), cm, bad_w)
    off = bytearray(w)
    off[50] ^= 1                                           # y limb changed: off the curve
    with pytest.raises(kzg_amd.EngineError):
        verifier.verify_eval((x, y), cm, bytes(off))
    params.gs.free()
    params.hs.free()


def test_g2_upload_rejects_invalid_points(engine):
    hs = PM.setup_g2(TAU, 2)
    good = b"".join(PM.g2_to_affine_mont(P) for P in hs)
    s = kzg_amd.SrsG2.upload(engine, good, 2)
    s.free()
    T = _non_subgroup_g2()
    assert PM.g2_is_on_curve(T)
    h = ctypes.c_void_p()
    bad = good[:192] + PM.g2_to_affine_mont(T)
    assert engine.lib.kzg_srs_upload_g2(engine.ctx, bad, 2, L.G2_AFFINE_MONT, ctypes.byref(h)) == L.KZG_ERR_BAD_POINT
    assert engine.lib.kzg_srs_upload_g2(engine.ctx, PM.g2_to_uncompressed(T), 1, L.G2_UNCOMPRESSED, ctypes.byref(h)) == L.KZG_ERR_BAD_POINT
    off = bytearray(good)
    off[200] ^= 1
    assert engine.lib.kzg_srs_upload_g2(engine.ctx, bytes(off), 2, L.G2_AFFINE_MONT, ctypes.byref(h)) == L.KZG_ERR_BAD_POINT


def test_non_canonical_scalars_are_taken_mod_r(engine):
    """a scalar >= r in the canonical format used to give a wrong commitment in the c = 17 / 15-window mode only.
    Every window width now computes sum (s_i mod r) P_i."""
    rng = random.Random(9)
    n = 1 << 12
    for wb in (0, 17):
        engine.set_option("window_bits", wb)
        try:
            params = kzg_amd.setup(engine, TAU, n, g2_len=0)
            sc = rand_scalars(rng, n)
            big = list(sc)
            for i in range(0, n, 7):
                big[i] = sc[i] + M.R if sc[i] + M.R < (1 << 256) else sc[i]
            big[1] = (1 << 256) - 1                                  # > 2r: needs both subtractions
            sc[1] = ((1 << 256) - 1) % M.R
            raw = b"".join(v.to_bytes(32, "little") for v in big)
            out = ctypes.create_string_buffer(96)
            rc = engine.lib.kzg_msm_g1(engine.ctx, params.gs.handle, 0, raw, n, L.FR_CANONICAL, 0, out, L.G1_AFFINE_MONT)
            assert rc == 0
            assert out.raw == engine.msm(params.gs, sc) == C.g1_mul(C.g1_generator(), C.poly_eval(sc, TAU)), wb
            params.gs.free()
        finally:
            engine.set_option("window_bits", 0)


def test_documented_limits_return_shape_errors(engine):
    lib, ctx = engine.lib, engine.ctx
    buf = ctypes.create_string_buffer(64)
    # NTT: log_n < 32 by the field (PolynomialDegreeTooLarge), <= 28 by the engine (two LDS passes under one outer level of 16);
    # so is the coset transform (round 4; 2^24 before)
    assert lib.kzg_ntt_fr(ctx, buf, 32, 0, 0) == L.KZG_ERR_DEGREE_TOO_LARGE
    assert lib.kzg_ntt_fr(ctx, buf, 29, 0, L.IN_DEVICE) == L.KZG_ERR_SHAPE
    assert "2^28" in engine.last_error()
    assert lib.kzg_coset_ntt_fr(ctx, buf, 29, 0, L.FR_CANONICAL, L.IN_DEVICE) == L.KZG_ERR_SHAPE
    assert "2^28" in engine.last_error()
    assert lib.kzg_coset_ntt_fr(ctx, buf, 32, 0, L.FR_CANONICAL, L.IN_DEVICE) == L.KZG_ERR_DEGREE_TOO_LARGE
    # create_witness_batched: at most 16384 opening points
    params = kzg_amd.setup(engine, TAU, 8, g2_len=0)
    k = 16385
    xs = kzg_amd.pack_scalars(list(range(1, k + 1)))
    rlen = ctypes.c_size_t()
    rc = lib.kzg_witness_coeff_batched(ctx, params.gs.handle, kzg_amd.pack_scalars([1] * 8), 8, xs, xs, k, L.FR_CANONICAL, 0, buf,
                                       L.G1_AFFINE_MONT, ctypes.create_string_buffer(32 * k), ctypes.byref(rlen))
    assert rc == L.KZG_ERR_SHAPE and "16384" in engine.last_error()
    # MSM range beyond the SRS, with an offset that would wrap in size_t
    out = ctypes.create_string_buffer(96)
    rc = lib.kzg_msm_g1(ctx, params.gs.handle, ctypes.c_size_t(2 ** 64 - 2), kzg_amd.pack_scalars([1] * 4), 4, L.FR_CANONICAL, 0, out,
                        L.G1_AFFINE_MONT)
    assert rc == L.KZG_ERR_SHAPE
    # the footprint query agrees with the formula in the header
    nbytes = ctypes.c_size_t()
    assert lib.kzg_srs_footprint(1 << 20, 0, 0, ctypes.byref(nbytes)) == 0
    assert nbytes.value == (1 << 20) * (96 + 15 * 128)
    assert lib.kzg_srs_footprint(1 << 20, 0, 4, ctypes.byref(nbytes)) == 0 and nbytes.value == (1 << 20) * (96 + 4 * 128)
    params.gs.free()


@pytest.mark.parametrize("rows", [1, 4, 7])
def test_window_rows_low_memory_srs(engine, rows):
    """Option window_rows: an SRS that keeps only `rows` of its window rows (multi-pass MSM + doubling chain) gives the same
    commitments, witnesses and batch results as the full table."""
    rng = random.Random(rows)
    n = 1 << 13
    for wb in (0, 17):
        engine.set_option("window_bits", wb)
        engine.set_option("window_rows", rows)
        try:
            params = kzg_amd.setup(engine, TAU, n, g2_len=0)
            c, W = params.gs.window_info()
            assert engine.lib.kzg_srs_table_rows(params.gs.handle) == min(rows, W)
            for sc in (rand_scalars(rng, n), [M.R - 1] * n, rand_scalars(rng, n, "u64"), rand_scalars(rng, 1000)):
                assert engine.msm(params.gs, sc) == C.g1_mul(C.g1_generator(), C.poly_eval(sc, TAU)), (rows, wb)
            polys = [rand_scalars(rng, 3000) for _ in range(5)]
            got = engine.msm_batch(params.gs, [x for p in polys for x in p], 3000, 5)
            assert got == [C.g1_mul(C.g1_generator(), C.poly_eval(p, TAU)) for p in polys]
            poly = kzg_amd.Polynomial(rand_scalars(rng, n))
            x = rng.randrange(M.R)
            y = C.poly_eval(poly.coeffs, x)
            w = kzg_amd.KZGProver(params).create_witness(poly, (x, y))
            assert w == C.g1_mul(C.g1_generator(), (C.poly_eval(poly.coeffs, TAU) - y) * M.fr_inv(TAU - x) % M.R)
            params.gs.free()
        finally:
            engine.set_option("window_rows", 0)
            engine.set_option("window_bits", 0)


def test_srs_of_another_gpu_is_refused():
    """An SRS is resident on ONE GPU; handing it to a context on another one (a host with several kzg_ctx, one per GPU, mixing up its
    handles) must be an error at the door, not a kernel on GPU a chasing pointers into GPU b.  Every MSM of every entry point passes
    msm_run's check.  One-GPU box: the SRS's device field is changed through the hooks build."""
    from tests.gpu_common import HooksEngine
    h = HooksEngine(0)
    lib = h.lib
    vp, sz, i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    lib.kzg_srs_setup_g1.argtypes = [vp, vp, i32, sz, ctypes.POINTER(vp)]
    lib.kzg_msm_g1.argtypes = [vp, vp, sz, vp, sz, i32, i32, vp, i32]
    lib.kzg_msm_g1_batch.argtypes = [vp, vp, sz, vp, sz, sz, i32, i32, vp, i32]
    lib.kzg_witness_coeff.argtypes = [vp, vp, vp, sz, vp, vp, i32, i32, vp, i32]
    lib.kzg_test_srs_set_device.argtypes = [vp, i32]
    lib.kzg_srs_free.argtypes = [vp, vp]
    srs = vp()
    n = 300
    assert lib.kzg_srs_setup_g1(h.ctx, (TAU % M.R).to_bytes(32, "little"), L.FR_CANONICAL, n, ctypes.byref(srs)) == 0
    rng = random.Random(31)
    coeffs = rand_scalars(rng, n)
    blob = kzg_amd.pack_scalars(coeffs)
    out = ctypes.create_string_buffer(96 * 4)
    want = C.g1_mul(C.g1_generator(), C.poly_eval(coeffs, TAU))
    assert lib.kzg_msm_g1(h.ctx, srs, 0, blob, n, L.FR_CANONICAL, 0, out, L.G1_AFFINE_MONT) == 0 and out.raw[:96] == want
    assert lib.kzg_test_srs_set_device(srs, 5) == 0
    assert lib.kzg_msm_g1(h.ctx, srs, 0, blob, n, L.FR_CANONICAL, 0, out, L.G1_AFFINE_MONT) == L.KZG_ERR_SHAPE
    assert "resident on GPU 5" in h.last_error()
    assert lib.kzg_msm_g1_batch(h.ctx, srs, 0, blob * 2, n, 2, L.FR_CANONICAL, 0, out, L.G1_AFFINE_MONT) == L.KZG_ERR_SHAPE
    x = (7).to_bytes(32, "little")
    y = C.poly_eval(coeffs, 7).to_bytes(32, "little")
    assert lib.kzg_witness_coeff(h.ctx, srs, blob, n, x, y, L.FR_CANONICAL, 0, out, L.G1_AFFINE_MONT) == L.KZG_ERR_SHAPE
    assert lib.kzg_test_srs_set_device(srs, 0) == 0                      # and the context is fine afterwards
    assert lib.kzg_msm_g1(h.ctx, srs, 0, blob, n, L.FR_CANONICAL, 0, out, L.G1_AFFINE_MONT) == 0 and out.raw[:96] == want
    lib.kzg_srs_free(h.ctx, srs)
    h.close()


def test_pipeline_narrowing_is_reported_not_silent():
    """a batched pipeline that finds fewer hardware queues than it wants narrows itself -- and now says so:
    kzg_ctx_info reports the plan (`narrowed_from=14+4`), and the context prints one line on stderr.  Forced here with option
    hw_queues = 4 (what a process whose host never asked for queues gets), in a child process so that the line can be read."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import kzg_amd\n"
            "e = kzg_amd.Engine(0); print('BEFORE', e.info())\n"
            "n = 1 << 12; p = kzg_amd.setup(e, 77, n, g2_len=0); buf = e.alloc_scalars(n * 24).fill_random(3)\n"
            "a = e.msm_batch(p.gs, buf, n, 24); print('FULL', e.info())\n"
            "e.set_option('hw_queues', 4); b = e.msm_batch(p.gs, buf, n, 24); print('NARROW', e.info()); assert a == b\n"
            "e.set_option('hw_queues', 0); c = e.msm_batch(p.gs, buf, n, 24); print('AGAIN', e.info()); assert a == c\n" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = {ln.split(" ", 1)[0]: ln for ln in r.stdout.splitlines() if ln.split(" ", 1)[0] in ("BEFORE", "FULL", "NARROW", "AGAIN")}
    assert "lanes=0 accum_streams=0" in lines["BEFORE"] and "narrowed_from=none" in lines["BEFORE"]
    assert "lanes=13 accum_streams=4" in lines["FULL"] and "narrowed_from=none" in lines["FULL"]
    assert "lanes=3 accum_streams=1 hw_queues_found=4 narrowed_from=13+4" in lines["NARROW"], lines["NARROW"]
    assert "narrowed_from=none" in lines["AGAIN"]
    warn = [ln for ln in r.stderr.splitlines() if ln.startswith("kzg: device 0: this context's 13 + 4 streams found only 4 hardware queues")]
    assert len(warn) == 1, r.stderr[-1500:]          # once per context, not once per call
