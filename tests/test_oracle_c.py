"""The C restatement (oracle/kzg_oracle.c) against the python model and the golden fixtures."""
import random

import pytest

from oracle import c_oracle as C
from oracle import kzg_model as M
from tests import golden_util as GU


def blob(P):
    return C.point_to_blob(P)


def test_g1_basics():
    rng = random.Random(7)
    g = C.g1_generator()
    assert C.blob_to_point(g) == M.G1 and C.g1_on_curve(g)
    k = rng.randrange(M.R)
    assert C.blob_to_point(C.g1_mul(g, k)) == M.g1_mul(M.G1, k)
    assert C.g1_add(C.g1_mul(g, M.R - 1), g) == bytes(96)        # r*G = O
    assert C.g1_add(g, g) == C.g1_mul(g, 2)
    P = C.g1_mul(g, 5)
    assert C.g1_from_uncompressed(C.g1_to_uncompressed(P)) == P
    assert C.g1_to_uncompressed(P) == M.g1_to_uncompressed(C.blob_to_point(P))


def test_setup_and_msm_vs_model():
    rng = random.Random(8)
    tau = rng.getrandbits(64)
    srs = C.setup_g1(tau, 40)
    gs = M.setup_g1(tau, 40)
    assert srs == b"".join(blob(P) for P in gs)
    sc = [rng.randrange(M.R) for _ in range(40)]
    sc[3], sc[5], sc[7] = 0, M.R - 1, 1
    assert C.msm_g1(srs, sc) == C.msm_g1(srs, sc, naive=True) == blob(M.g1_multi_exp(gs, sc))
    assert C.msm_g1(srs, sc) == C.g1_mul(C.g1_generator(), M.Polynomial(sc).eval(tau))
    for n in (1, 2, 3, 5):
        assert C.msm_g1(srs[: 96 * n], sc[:n]) == blob(M.g1_multi_exp(gs[:n], sc[:n]))


@pytest.mark.parametrize("case", GU.load("msm.json")["cases"], ids=lambda c: c["name"])
def test_golden_msm(case):
    pts = b"".join(blob(GU.pt(h)) for h in case["points"])
    sc = [GU.sc(h) for h in case["scalars"]]
    assert C.msm_g1(pts, sc) == blob(GU.pt(case["result"]))


@pytest.mark.parametrize("case", GU.load("ntt.json")["cases"], ids=lambda c: f"log{c['log_n']}")
def test_golden_ntt(case):
    xs = [GU.sc(h) for h in case["input"]]
    want = [GU.sc(h) for h in case["fft"]]
    assert C.fft(xs) == want
    assert C.fft(want, inverse=True) == xs
    assert C.fft(xs, inverse=True) == [GU.sc(h) for h in case["ifft"]]     # EvaluationDomain::ifft of the input itself (src/ft.rs:115-140)
    assert C.compute_omega(len(xs))[2] == GU.sc(case["omega"])


def test_golden_kzg():
    g = GU.load("kzg.json")
    srs = b"".join(blob(GU.pt(h)) for h in g["srs_compressed"])
    tau = GU.sc(g["tau"])
    assert srs == C.setup_g1(tau, 16)
    c = g["coeff"]
    coeffs = [GU.sc(h) for h in c["coeffs"]]
    n = len(coeffs)
    assert C.msm_g1(srs[: 96 * n], coeffs) == blob(GU.pt(c["commit"]))
    qb, nz = C.witness_quotient_bytes(C.scalars_to_bytes(coeffs), n, GU.sc(c["x"]), GU.sc(c["y"]))
    assert not nz and C.msm_g1_raw(srs[: 96 * (n - 1)], qb, n - 1) == blob(GU.pt(c["witness"]))
    _, nz = C.witness_quotient_bytes(C.scalars_to_bytes(coeffs), n, GU.sc(c["x"]), GU.sc(c["wrong_y"]))
    assert nz
    e = g["eval"]
    d = e["d"]
    ev = [GU.sc(h) for h in e["evals"]]
    assert C.fft([GU.sc(h) for h in e["coeffs"]]) == ev
    lag = b"".join(blob(GU.pt(h)) for h in e["lagrange_compressed"])
    assert C.msm_g1(lag, ev) == blob(GU.pt(e["commit"]))
    num = [(v - ev[e["index"]]) % M.R for v in ev]
    q = C.div_by_omega_i_bytes(C.scalars_to_bytes(num), d, e["index"])
    assert C.msm_g1_raw(lag, q, d) == blob(GU.pt(e["witness"]))


def test_poly_helpers_vs_model():
    rng = random.Random(9)
    num = [rng.randrange(M.R) for _ in range(30)]
    den = [rng.randrange(M.R) for _ in range(7)]
    q, r = C.long_division(num, den)
    Q, Rm = M.Polynomial(num).long_division(M.Polynomial(den))
    assert q == Q.slice_coeffs() and r == Rm.coeffs[:6]
    x = rng.randrange(M.R)
    assert C.poly_eval(num, x) == M.Polynomial(num).eval(x)
    ev = M.EvaluationDomain.from_coeffs([rng.randrange(M.R) for _ in range(16)])
    got = C.bytes_to_scalars(C.div_by_omega_i_bytes(C.scalars_to_bytes(ev.coeffs), 16, 3))
    assert got == M.div_by_omega_i(ev, 3).coeffs
    for ln in range(0, 8):
        xs = [rng.randrange(M.R) for _ in range(1 << ln)]
        e = M.EvaluationDomain.from_coeffs(xs)
        e.fft()
        assert C.fft(xs) == e.coeffs


def test_fast_msm_of_the_timing_leg_equals_the_checker():
    """orc_msm_g1_fast (bench.py's cpu_baseline timing leg: signed 16-bit windows, batch-affine buckets, unrolled Montgomery
    multiplication) against the plain Pippenger the tests check with -- random and u64-valued scalars, the extremes, and the cases
    the affine formulas special-case: a bucket receiving the same point twice (doubling), a point and its negative (the bucket
    empties), identity points, n = 0 and 1."""
    rng = random.Random(77)
    n = 700
    blob_all = C.setup_g1(0x1234567, n)
    pts = [blob_all[96 * i:96 * i + 96] for i in range(n)]

    def neg(p):
        return p[:48] + ((M.Q - int.from_bytes(p[48:], "little")) % M.Q).to_bytes(48, "little")

    def same(ptsl, scal):
        pb, sb = b"".join(ptsl), C.scalars_to_bytes([s % M.R for s in scal])
        assert C.msm_g1_fast_raw(pb, sb, len(scal)) == C.msm_g1_raw(pb, sb, len(scal))

    same(pts, [rng.randrange(M.R) for _ in range(n)])
    same(pts, [rng.getrandbits(64) for _ in range(n)])
    same(pts[:200], [M.R - 1] * 200)
    same(pts[:50], [0] * 50)
    same([pts[5]] * 300, [rng.randrange(M.R) for _ in range(300)])
    same([pts[5]] * 300, [12345] * 300)
    same([pts[7], neg(pts[7])] * 60 + pts[:40], [999] * 120 + [rng.randrange(M.R) for _ in range(40)])
    same([pts[7], neg(pts[7]), pts[7], pts[7], neg(pts[7])] * 30, [999] * 150)
    same([bytes(96)] * 10 + pts[:10], [rng.randrange(M.R) for _ in range(20)])
    same(pts[:1], [1])
    same([], [])
