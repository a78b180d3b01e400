"""Positional tables (kzg_srs::naf, msm.hip k_naf_recode / k_bin_scatter_naf, naf.h): 255 table rows 2^j P, scalars recoded in
width-18 non-adjacent form, 2^16 buckets of weight 2 b + 1.  Forced at small sizes (option naf_window = 18) so that the oracle's
Pippenger covers ragged term counts, offsets, batches and the scalar shapes that stress the recoding (carries through long runs
of ones, values around 2^254 and r, u64-valued, zero), and compared with the 17-bit window tables at 2^17."""
import ctypes
import random

import pytest

import kzg_amd
from kzg_amd import _lib as L
from oracle import c_oracle as C
from oracle import kzg_model as M
from tests.gpu_common import rand_scalars

pytestmark = pytest.mark.gpu
R = M.R
TAU = 0x5EED5EED5EED5EED


@pytest.fixture()
def naf_engine(engine):
    engine.set_option("naf_window", 18)
    yield engine
    engine.set_option("naf_window", 0)


def test_naf_small_sizes_vs_oracle(naf_engine):
    e = naf_engine
    rng = random.Random(18)
    n = 5000
    params = kzg_amd.setup(e, TAU, n, g2_len=0)
    assert params.gs.window_info() == (18, 15) and e.lib.kzg_srs_table_rows(params.gs.handle) == 255
    blob = C.setup_g1(TAU, n)
    assert params.gs.download() == blob
    G = C.g1_generator()
    shapes = {
        "random": lambda k: rand_scalars(rng, k),
        "u64": lambda k: rand_scalars(rng, k, "u64"),
        "all_equal": lambda k: [rng.randrange(R)] * k,
        "r_minus_1": lambda k: [R - 1] * k,
        "ones_runs": lambda k: [((1 << rng.randrange(1, 255)) - 1) % R for _ in range(k)],        # a carry through the whole run
        "around_2^254": lambda k: [((1 << 254) + d) % R for d in (-2, -1, 0, 1, 2, 1 << 200, -(1 << 200))] * (k // 7 + 1),
        "alternating": lambda k: [int("aaaaaaaa" * 8, 16) % R, int("55555555" * 8, 16) % R] * (k // 2 + 1),
        "single_top_digit": lambda k: [(1 << 236) * ((1 << 17) - 1) % R] * k,
        "zeros_and_small": lambda k: [i % 3 for i in range(k)],
    }
    for k in (1, 2, 63, 64, 1023, 1024, 1025, 2049, 5000):
        for name, make in shapes.items():
            sc = make(k)[:k]
            assert e.msm(params.gs, sc) == C.msm_g1(blob[:96 * k], sc), (name, k)
    assert e.msm(params.gs, [], n=0) == bytes(96)
    assert e.msm(params.gs, [0] * 3000) == bytes(96)
    sub = rand_scalars(rng, 2100)                      # a sub-range of the SRS (offset) and the batched pipeline
    want = C.msm_g1(blob[96 * 1234:96 * (1234 + 2100)], sub)
    assert e.msm(params.gs, sub, offset=1234) == want
    polys = [rand_scalars(rng, 3000) for _ in range(5)]
    got = e.msm_batch(params.gs, [x for p in polys for x in p], 3000, 5)
    assert got == [C.g1_mul(G, C.poly_eval(p, TAU)) for p in polys]
    # the prover on top of it
    poly = kzg_amd.Polynomial(rand_scalars(rng, n))
    x = rng.randrange(R)
    y = C.poly_eval(poly.coeffs, x)
    w = kzg_amd.KZGProver(params).create_witness(poly, (x, y))
    assert w == C.g1_mul(G, (C.poly_eval(poly.coeffs, TAU) - y) * M.fr_inv(TAU - x) % R)
    params.gs.free()


def test_naf_montgomery_scalars_and_formats(naf_engine):
    e = naf_engine
    rng = random.Random(19)
    n = 3000
    params = kzg_amd.setup(e, TAU, n, g2_len=0)
    sc = rand_scalars(rng, n)
    want = C.g1_mul(C.g1_generator(), C.poly_eval(sc, TAU))
    buf = e.alloc_scalars(n, sfmt=L.FR_MONT)
    buf.upload(b"".join(M.fr_to_mont_le(v) for v in sc))
    out = ctypes.create_string_buffer(96)
    rc = e.lib.kzg_msm_g1(e.ctx, params.gs.handle, 0, buf.ptr, n, L.FR_MONT, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
    assert rc == 0 and out.raw == want
    # non-canonical scalars (>= r) are taken mod r before the recoding
    big = [(v + R) if v + R < (1 << 256) else v for v in sc[:500]]
    raw = b"".join(v.to_bytes(32, "little") for v in big)
    assert e.msm(params.gs, raw, n=500) == C.g1_mul(C.g1_generator(), C.poly_eval(sc[:500], TAU))
    buf.free()
    params.gs.free()


def test_naf_equals_window_tables_2_17(engine):
    """positional tables (on request) against the default 17-bit window tables at 2^17: same commitments for full-width,
    u64-valued and all-equal coefficients; expected values from the oracle."""
    n = 1 << 17
    p_win = kzg_amd.setup(engine, TAU, n, g2_len=0)
    engine.set_option("naf_window", 18)
    try:
        p_naf = kzg_amd.setup(engine, TAU, n, g2_len=0)
    finally:
        engine.set_option("naf_window", 0)
    assert p_naf.gs.window_info() == (18, 15) and p_win.gs.window_info() == (17, 15)
    G = C.g1_generator()
    out = ctypes.create_string_buffer(96)
    for seed, u64 in ((1, False), (2, True)):
        buf = engine.alloc_scalars(n).fill_random(seed, u64_valued=u64)
        want = C.g1_mul(G, C.poly_eval_bytes(buf.download(), n, TAU))
        for p in (p_naf, p_win):
            rc = engine.lib.kzg_commit_coeff(engine.ctx, p.gs.handle, buf.ptr, n, buf.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
            assert rc == 0 and out.raw == want, (seed, p.gs.window_info())
        buf.free()
    eq = engine.alloc_scalars(n)
    v = 0x1234567890ABCDEF1122334455667788990011223344556677889900AABBCCDD % R
    eq.upload(M.fr_to_le(v) * n)
    want = C.g1_mul(G, v * (pow(TAU, n, R) - 1) * pow(TAU - 1, -1, R) % R)        # v (tau^n - 1) / (tau - 1)
    for p in (p_naf, p_win):
        rc = engine.lib.kzg_commit_coeff(engine.ctx, p.gs.handle, eq.ptr, n, eq.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
        assert rc == 0 and out.raw == want
    eq.free()
    p_naf.gs.free()
    p_win.gs.free()


def test_naf_is_opt_in(engine):
    """kzg_srs_footprint follows the default policy: window tables (positional tables only with option naf_window = 18)."""
    b = ctypes.c_size_t()
    for log_n, rows in ((16, 20), (17, 15), (20, 15), (22, 15), (23, 13), (24, 13)):
        assert engine.lib.kzg_srs_footprint(1 << log_n, 0, 0, ctypes.byref(b)) == 0
        assert b.value == (1 << log_n) * (96 + rows * 128), (log_n, b.value)
    p = kzg_amd.setup(engine, TAU, 1 << 17, g2_len=0)
    assert p.gs.window_info() == (17, 15) and engine.lib.kzg_srs_table_rows(p.gs.handle) == 15
    p.gs.free()
    with pytest.raises(kzg_amd.ReferencePanic):
        engine.set_option("naf_window", 7)
    with pytest.raises(kzg_amd.ReferencePanic):
        engine.set_option("heavy_bins", 3)
    engine.set_option("heavy_bins", 0)
