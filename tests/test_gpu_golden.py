"""The HIP path against the committed golden fixtures (tests/golden/*.json), through the C ABI, using the
canonical wire formats (32-byte LE scalars, zcash-compressed G1) on both input and output."""
import pytest

import kzg_amd
from kzg_amd import _lib as L
from tests import golden_util as GU

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", GU.load("msm.json")["cases"], ids=lambda c: c["name"])
def test_golden_msm(engine, case):
    raw = b"".join(bytes.fromhex(h) for h in case["points"])
    n = len(case["points"])
    srs = kzg_amd.Srs.upload(engine, raw, n, L.G1_ZCASH_COMPRESSED)
    scal = b"".join(bytes.fromhex(h) for h in case["scalars"])
    assert engine.msm(srs, scal, ofmt=L.G1_ZCASH_COMPRESSED).hex() == case["result"]
    srs.free()


@pytest.mark.parametrize("case", GU.load("ntt.json")["cases"], ids=lambda c: f"log{c['log_n']}")
def test_golden_ntt(engine, case):
    xs = [GU.sc(h) for h in case["input"]]
    want = [GU.sc(h) for h in case["fft"]]
    assert engine.ntt(xs, case["log_n"]) == want
    assert engine.ntt(want, case["log_n"], inverse=True) == xs
    assert engine.ntt(xs, case["log_n"], inverse=True) == [GU.sc(h) for h in case["ifft"]]     # EvaluationDomain::ifft of the input itself (src/ft.rs:115-140)
    assert kzg_amd.compute_omega(len(xs))[2] == GU.sc(case["omega"])


def test_golden_kzg(engine):
    g = GU.load("kzg.json")
    tau = GU.sc(g["tau"])
    params = kzg_amd.setup(engine, tau, 16)
    up = kzg_amd.Srs.upload(engine, b"".join(bytes.fromhex(h) for h in g["srs_compressed"]), 16, L.G1_ZCASH_COMPRESSED)
    assert params.gs.download() == up.download()
    prover = kzg_amd.KZGProver(params)
    c = g["coeff"]
    p = kzg_amd.Polynomial([GU.sc(h) for h in c["coeffs"]])
    assert prover.commit(p, ofmt=L.G1_ZCASH_COMPRESSED).hex() == c["commit"]
    assert prover.create_witness(p, (GU.sc(c["x"]), GU.sc(c["y"])), ofmt=L.G1_ZCASH_COMPRESSED).hex() == c["witness"]
    with pytest.raises(kzg_amd.PointNotOnPolynomial):
        prover.create_witness(p, (GU.sc(c["x"]), GU.sc(c["wrong_y"])))
    d1 = g["degree1"]
    p1 = kzg_amd.Polynomial([GU.sc(h) for h in d1["coeffs"]] + [0] * 11)
    assert prover.create_witness(p1, (GU.sc(d1["x"]), GU.sc(d1["y"])), ofmt=L.G1_ZCASH_COMPRESSED).hex() == d1["witness"]
    b = g["batched"]
    wit = prover.create_witness_batched(p, [GU.sc(h) for h in b["xs"]], [GU.sc(h) for h in b["ys"]],
                                        ofmt=L.G1_ZCASH_COMPRESSED)
    assert wit.elem().hex() == b["w"]
    assert wit.polynomial().coeffs == [GU.sc(h) for h in b["r"]]
    e = g["eval"]
    pe = kzg_amd.setup(engine, tau, e["d"])
    lag = kzg_amd.compute_lagrange_basis(pe)
    lag_up = kzg_amd.Srs.upload(engine, b"".join(bytes.fromhex(h) for h in e["lagrange_compressed"]), e["d"],
                                L.G1_ZCASH_COMPRESSED)
    assert lag.download() == lag_up.download()
    evp = kzg_amd.KZGProverEvalForm(pe, lag)
    ev = kzg_amd.EvaluationDomain.from_coeffs([GU.sc(h) for h in e["coeffs"]])
    ev.fft(engine)
    assert ev.coeffs == [GU.sc(h) for h in e["evals"]]
    assert evp.commit(ev, ofmt=L.G1_ZCASH_COMPRESSED).hex() == e["commit"]
    assert evp.create_witness(ev, e["index"], ofmt=L.G1_ZCASH_COMPRESSED).hex() == e["witness"]
    for s in (params.gs, up, pe.gs, lag, lag_up):
        s.free()


def test_golden_verify(engine):
    """Verifier half against tests/golden/verify.json (G2 parameters, G2 MSM, pairing products, verifier verdicts)."""
    import ctypes
    g = GU.load("verify.json")
    tau, n = GU.sc(g["tau"]), g["n"]
    params = kzg_amd.setup(engine, tau, n, g2_len=n)
    assert params.hs.download(0, 6, pfmt=L.G2_COMPRESSED).hex() == "".join(g["hs_compressed"])
    assert params.hs.download(1, 1, pfmt=L.G2_UNCOMPRESSED).hex() == g["hs1_uncompressed"]
    up = kzg_amd.SrsG2.upload(engine, bytes.fromhex("".join(g["hs_compressed"])), 6, L.G2_COMPRESSED)
    assert up.download() == params.hs.download(0, 6)
    assert up.msm([GU.sc(h) for h in g["msm_g2"]["scalars"]], ofmt=L.G2_COMPRESSED).hex() == g["msm_g2"]["result"]
    lag_h = kzg_amd.setup_lagrange_g2(engine, tau, 4)
    assert lag_h.download(pfmt=L.G2_COMPRESSED).hex() == "".join(g["lagrange_h_d4"])
    checks = g["pairing_checks"]
    ok = ctypes.create_string_buffer(len(checks))
    rc = engine.lib.kzg_pairing_check(engine.ctx, bytes.fromhex("".join(h for c in checks for h in c["g1"])),
                                      L.G1_ZCASH_COMPRESSED, bytes.fromhex("".join(h for c in checks for h in c["g2"])),
                                      L.G2_COMPRESSED, 2, len(checks), ok)
    assert rc == 0, engine.last_error()
    assert [bool(b) for b in ok.raw] == [c["is_one"] for c in checks]
    verifier = kzg_amd.KZGVerifier(params)
    v = g["verify_eval"]
    c, w = bytes.fromhex(v["commitment"]), bytes.fromhex(v["witness"])
    pts = [(GU.sc(a), GU.sc(b)) for a, b in v["points"]]
    assert verifier.verify_eval_many(pts, [c] * len(pts), [w] * len(pts), pfmt=L.G1_ZCASH_COMPRESSED) == v["ok"]
    prover = kzg_amd.KZGProver(params)
    poly = kzg_amd.Polynomial([GU.sc(h) for h in v["coeffs"]])
    assert prover.commit(poly, ofmt=L.G1_ZCASH_COMPRESSED) == c
    assert prover.create_witness(poly, pts[0], ofmt=L.G1_ZCASH_COMPRESSED) == w
    b = g["verify_eval_batched"]
    r = [GU.sc(h) for h in b["r"]]
    wit = kzg_amd.KZGBatchWitness(kzg_amd.Polynomial.new_from_coeffs(r, len(r) - 1), bytes.fromhex(b["w"]))
    assert verifier.verify_eval_batched([GU.sc(h) for h in b["xs"]], c, wit, pfmt=L.G1_ZCASH_COMPRESSED) == b["ok"]
    assert verifier.verify_eval_batched([GU.sc(h) for h in b["xs_bad"]], c, wit, pfmt=L.G1_ZCASH_COMPRESSED) == b["ok_bad"]
    for h in (params.gs, params.hs, up, lag_h):
        h.free()


@pytest.mark.parametrize("naf", [0, 18], ids=["windows_c17", "positional_naf18"])
@pytest.mark.parametrize("case", GU.load("prod.json")["commits"], ids=lambda c: f"2^{c['log_n']}_seed{c['seed']}")
def test_golden_production_path_commit(engine, case, naf):
    """tests/golden/prod.json: commitments at the sizes where the production MSM path runs (17-bit windows, two-level sort, 15
    table rows), computed by oracle/kzg_model.py alone (coefficient stream, Horner, one scalar multiplication).  Nothing on the
    right-hand side comes from this process: 48 committed bytes per case."""
    tau = GU.sc(GU.load("prod.json")["tau"])
    n = 1 << case["log_n"]
    engine.set_option("naf_window", naf)          # both production table layouts against the same 48 committed bytes
    try:
        params = kzg_amd.setup(engine, tau, n, g2_len=0)
    finally:
        engine.set_option("naf_window", 0)
    assert params.gs.window_info() == ((18, 15) if naf else (17, 15))
    buf = engine.alloc_scalars(n).fill_random(case["seed"], u64_valued=case["u64_valued"])
    import ctypes
    out = ctypes.create_string_buffer(48)
    rc = engine.lib.kzg_commit_coeff(engine.ctx, params.gs.handle, buf.ptr, n, buf.sfmt, L.IN_DEVICE, out, L.G1_ZCASH_COMPRESSED)
    assert rc == 0, engine.last_error()
    assert out.raw.hex() == case["commit"]
    # the batched pipeline (the path bench.py times) on the same polynomial, three copies
    out3 = ctypes.create_string_buffer(48 * 3)
    rep = engine.alloc_scalars(3 * n)
    blob = buf.download()
    rep.upload(blob * 3)
    rc = engine.lib.kzg_msm_g1_batch(engine.ctx, params.gs.handle, 0, rep.ptr, n, 3, rep.sfmt, L.IN_DEVICE, out3, L.G1_ZCASH_COMPRESSED)
    assert rc == 0, engine.last_error()
    assert [out3.raw[48 * j:48 * j + 48].hex() for j in range(3)] == [case["commit"]] * 3
    rep.free()
    buf.free()
    params.gs.free()


def test_golden_production_path_witness(engine):
    g = GU.load("prod.json")
    tau, w = GU.sc(g["tau"]), g["witness"]
    n = 1 << w["log_n"]
    params = kzg_amd.setup(engine, tau, n, g2_len=0)
    buf = engine.alloc_scalars(n).fill_random(w["seed"])
    import ctypes
    out = ctypes.create_string_buffer(48)
    rc = engine.lib.kzg_witness_coeff(engine.ctx, params.gs.handle, buf.ptr, n, bytes.fromhex(w["x"]), bytes.fromhex(w["y"]), buf.sfmt,
                                      L.IN_DEVICE, out, L.G1_ZCASH_COMPRESSED)
    assert rc == 0, engine.last_error()
    assert out.raw.hex() == w["witness"]
    buf.free()
    params.gs.free()


def test_published_points_through_the_engine(engine):
    """The only literals in the repository that were not produced by its own oracle (tests/test_oracle_reference_vectors.py:
    compressed [1]G, [2]G, [3]G and the G2 generator as published, [upstream-memory]): the engine's SRS generation, MSM, point sum,
    affine conversion and zcash serialisation reproduce them -- commit of the constant polynomial k against setup(tau, n).gs, the
    sum G + G and 2G + G, [2]H through the G2 multi-exponentiation."""
    import kzg_amd
    from kzg_amd import _lib as L
    from tests.test_oracle_reference_vectors import PUBLISHED_2G2_PREFIX, PUBLISHED_G1, PUBLISHED_G2_GENERATOR
    params = kzg_amd.setup(engine, 0x1234567, 4, g2_len=2)
    for k, hexv in PUBLISHED_G1.items():
        assert engine.msm(params.gs, [k], ofmt=L.G1_ZCASH_COMPRESSED).hex() == hexv             # gs[0] = G
        assert engine.msm(params.gs, [k, 0, 0, 0], ofmt=L.G1_ZCASH_COMPRESSED).hex() == hexv
    g = params.gs.download(0, 1)
    assert engine.g1_sum([g, g], ofmt=L.G1_ZCASH_COMPRESSED).hex() == PUBLISHED_G1[2]
    assert engine.g1_sum([g, g, g], ofmt=L.G1_ZCASH_COMPRESSED).hex() == PUBLISHED_G1[3]
    assert params.hs.download(0, 1, pfmt=L.G2_COMPRESSED).hex() == PUBLISHED_G2_GENERATOR           # hs[0] = H
    assert params.hs.msm([2], ofmt=L.G2_COMPRESSED).hex().startswith(PUBLISHED_2G2_PREFIX)
    # uploading the published bytes gives the same resident points as generating them
    up = kzg_amd.Srs.upload(engine, bytes.fromhex(PUBLISHED_G1[1] + PUBLISHED_G1[2] + PUBLISHED_G1[3]), 3, pfmt=L.G1_ZCASH_COMPRESSED)
    one = kzg_amd.setup(engine, 1, 1, g2_len=0)
    assert up.download(0, 1) == one.gs.download(0, 1)
    assert engine.msm(up, [1, 1, 1], ofmt=L.G1_ZCASH_COMPRESSED) == engine.msm(params.gs, [6], ofmt=L.G1_ZCASH_COMPRESSED)
    up.free()
    one.gs.free()
    params.gs.free()
    params.hs.free()


def test_published_montgomery_constants_through_the_engine(engine):
    """The zero-copy formats are what they claim to be: KZG_G1_AFFINE_MONT_96 / KZG_G1_JACOBIAN_MONT_144 hold the generator exactly as
    the public implementations' `G1Affine::generator()` / blst_p1 do (x, y[, z] as 6 little-endian u64 limbs, Montgomery radix 2^384),
    KZG_FR_MONT_LE_32 holds 1 as their `Scalar::one()` (radix 2^256), and the NTT's twiddles are the published roots of unity.
    Literals: tests/test_oracle_reference_vectors.py ([upstream-memory] of public material, not produced by this repository)."""
    import ctypes
    import kzg_amd
    from kzg_amd import _lib as L
    from oracle import kzg_model as M
    from tests import test_oracle_reference_vectors as V
    one = kzg_amd.setup(engine, 0xABCDEF, 2, g2_len=0)
    assert one.gs.download(0, 1) == V.PUBLISHED_G1_GENERATOR_MONT                                      # gs[0] = G, resident form
    # a scalar handed over in Montgomery form: the published R is 1, R2 is 2^256 mod r
    out = ctypes.create_string_buffer(48)
    for blob, k in ((V.PUBLISHED_FR_ONE_MONT, 1), (V.PUBLISHED_FR_R2, (1 << 256) % M.R)):
        rc = engine.lib.kzg_msm_g1(engine.ctx, one.gs.handle, 0, blob, 1, L.FR_MONT, 0, out, L.G1_ZCASH_COMPRESSED)
        assert rc == 0, engine.last_error()
        assert out.raw == engine.msm(one.gs, [k], ofmt=L.G1_ZCASH_COMPRESSED)
        if k == 1:
            assert out.raw.hex() == V.PUBLISHED_G1[1]
    # blst_p1 of the generator = (x, y, 1) in Montgomery form: uploads to the same resident point; the engine's own Jacobian output
    # of [1]G normalises to it
    up = kzg_amd.Srs.upload(engine, V.PUBLISHED_G1_GENERATOR_MONT + V.PUBLISHED_FQ_ONE_MONT, 1, pfmt=L.G1_JACOBIAN_MONT)
    assert up.download(0, 1) == V.PUBLISHED_G1_GENERATOR_MONT
    assert engine.msm(up, [3], ofmt=L.G1_ZCASH_COMPRESSED).hex() == V.PUBLISHED_G1[3]
    aff = engine.msm(one.gs, [1], ofmt=L.G1_AFFINE_MONT)
    assert aff == V.PUBLISHED_G1_GENERATOR_MONT
    # roots of unity: compute_omega and the transform itself
    for k in (2, 3):
        assert kzg_amd.compute_omega(1 << k)[2] == V.PUBLISHED_ROOTS_OF_UNITY[k]
    w4, w8 = V.PUBLISHED_ROOTS_OF_UNITY[2], V.PUBLISHED_ROOTS_OF_UNITY[3]
    assert engine.ntt([0, 1, 0, 0], 2) == [1, w4, M.R - 1, M.R - w4]                                    # fft(X) = (w^i)_i
    assert engine.ntt([0, 1] + [0] * 6, 3) == [pow(w8, i, M.R) for i in range(8)]
    assert pow(w8, 2, M.R) == w4 and pow(w4, 2, M.R) == M.R - 1
    # the 2^32-th root: 2^27 squarings away from the 32-point transform's twiddle
    w32pt = engine.ntt([0, 1] + [0] * 30, 5)[1]
    assert w32pt == pow(V.PUBLISHED_ROOTS_OF_UNITY[32], 1 << 27, M.R)
    up.free()
    one.gs.free()
