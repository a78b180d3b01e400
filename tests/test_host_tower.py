"""CPU unit tests of kzg_amd/csrc/tower.h (Fq2/Fq6/Fq12, G2, ate pairing) -- the same source hipcc compiles for
gfx950 -- against the python oracle (oracle/pairing_model.py)."""
import ctypes
import os
import random
import subprocess

import pytest

from oracle import kzg_model as M, pairing_model as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Q = M.Q


@pytest.fixture(scope="module")
def L(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("ht") / "libhosttower.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so,
                           os.path.join(ROOT, "tests", "host_tower.cpp")])
    return ctypes.CDLL(so)


def f2b(a):
    return a[0].to_bytes(48, "little") + a[1].to_bytes(48, "little")


def f2u(b):
    return (int.from_bytes(b[:48], "little"), int.from_bytes(b[48:96], "little"))


def f12b(a):
    return b"".join(f2b(c) for c in a)


def f12u(b):
    return tuple(f2u(b[96 * k:96 * k + 96]) for k in range(6))


def g1b(p):
    return bytes(96) if p is None else p[0].to_bytes(48, "little") + p[1].to_bytes(48, "little")


def g2b(p):
    return bytes(192) if p is None else f2b(p[0]) + f2b(p[1])


def g2u(b):
    p = (f2u(b[:96]), f2u(b[96:]))
    return None if p == ((0, 0), (0, 0)) else p


def out(f, n, *args):
    o = ctypes.create_string_buffer(n)
    f(*args, o)
    return o.raw


def rf2(rng):
    return (rng.randrange(Q), rng.randrange(Q))


def rf12(rng):
    return tuple(rf2(rng) for _ in range(6))


def test_fq2_ops(L):
    rng = random.Random(1)
    cases = [((0, 0), (5, 7)), ((Q - 1, Q - 1), (Q - 1, Q - 1)), ((1, 0), (0, 1)), ((0, 1), (0, 1))]
    cases += [(rf2(rng), rf2(rng)) for _ in range(100)]
    for a, c in cases:
        assert f2u(out(L.ht_f2_mul, 96, f2b(a), f2b(c))) == P.f2_mul(a, c)
        assert f2u(out(L.ht_f2_sqr, 96, f2b(a))) == P.f2_sqr(a)
        if a != (0, 0):
            assert f2u(out(L.ht_f2_inv, 96, f2b(a))) == P.f2_inv(a)


def test_fq12_ops(L):
    rng = random.Random(2)
    for it in range(20):
        a, c = rf12(rng), rf12(rng)
        if it == 0:
            a = P.F12_ONE
        assert f12u(out(L.ht_f12_mul, 576, f12b(a), f12b(c))) == P.f12_mul(a, c)
        assert f12u(out(L.ht_f12_sqr, 576, f12b(a))) == P.f12_sqr(a)
        assert f12u(out(L.ht_f12_inv, 576, f12b(c))) == P.f12_inv(c)
        assert f12u(out(L.ht_f12_frob, 576, f12b(a), 1)) == P.f12_frob(a) == P.f12_pow(a, Q) if it < 2 else True
        assert f12u(out(L.ht_f12_frob, 576, f12b(a), 2)) == P.f12_frob2(a)
    a = rf12(rng)
    assert P.f12_frob2(a) == P.f12_frob(P.f12_frob(a))


def test_g2_ops(L):
    rng = random.Random(3)
    assert g2u(out(L.ht_g2_generator, 192)) == P.G2
    assert L.ht_g2_on_curve(g2b(P.G2)) == 1 and L.ht_g2_on_curve(g2b(None)) == 1
    assert L.ht_g2_on_curve(g2b((P.G2[0], P.f2_add(P.G2[1], (1, 0))))) == 0
    for k in [0, 1, 2, 3, M.R - 1, M.R, rng.randrange(M.R), rng.randrange(1 << 64)]:
        assert g2u(out(L.ht_g2_mul, 192, g2b(P.G2), (k % (1 << 256)).to_bytes(32, "little"))) == P.g2_mul(P.G2, k % M.R)
    a, c = P.g2_mul(P.G2, 11), P.g2_mul(P.G2, 31)
    for x, y in [(a, c), (a, a), (a, P.g2_neg(a)), (a, None), (None, c), (None, None)]:
        assert g2u(out(L.ht_g2_add, 192, g2b(x), g2b(y))) == P.g2_add(x, y)


def test_pairing_matches_oracle_and_is_bilinear(L):
    rng = random.Random(4)
    a, c = rng.randrange(M.R), rng.randrange(M.R)
    Pa, Qc = M.g1_mul(M.G1, a), P.g2_mul(P.G2, c)
    f = f12u(out(L.ht_miller_loop, 576, g1b(Pa), g2b(Qc), 1))
    assert f == P.miller_loop([(Pa, Qc)])
    e = f12u(out(L.ht_final_exp, 576, f12b(f)))
    # cyclotomic squaring (used by the exponentiations by z) == plain squaring on the cyclotomic subgroup
    assert f12u(out(L.ht_f12_cyclotomic_sqr, 576, f12b(e))) == P.f12_sqr(e) == P.f12_cyclotomic_sqr(e)
    assert e == P.final_exponentiation(f) == P.f12_pow(P.pairing(M.G1, P.G2), a * c % M.R)
    # product checks: e(aG, cH) e(-acG, H) = 1; a wrong exponent is rejected; identity members contribute 1
    neg = M.g1_neg(M.g1_mul(M.G1, a * c % M.R))
    assert L.ht_pairing_product_is_one(g1b(Pa) + g1b(neg), g2b(Qc) + g2b(P.G2), 2) == 1
    bad = M.g1_neg(M.g1_mul(M.G1, (a * c + 1) % M.R))
    assert L.ht_pairing_product_is_one(g1b(Pa) + g1b(bad), g2b(Qc) + g2b(P.G2), 2) == 0
    assert L.ht_pairing_product_is_one(g1b(None) + g1b(Pa), g2b(Qc) + g2b(None), 2) == 1
    assert L.ht_pairing_product_is_one(g1b(Pa), g2b(Qc), 1) == 0
    two = f12u(out(L.ht_miller_loop, 576, g1b(Pa) + g1b(neg), g2b(Qc) + g2b(P.G2), 2))
    assert two == P.miller_loop([(Pa, Qc), (neg, P.G2)])
    # precomputed lines for a fixed Q give the same Miller value (pair 0 stored, pair 1 on the fly; then 3 pairs)
    assert f12u(out(L.ht_miller_loop_fixed, 576, g1b(Pa) + g1b(neg), g2b(Qc) + g2b(P.G2), 2)) == two
    three = [(Pa, Qc), (neg, P.G2), (M.g1_mul(M.G1, 5), P.g2_mul(P.G2, 7))]
    assert f12u(out(L.ht_miller_loop_fixed, 576, b"".join(g1b(p) for p, _ in three), b"".join(g2b(q) for _, q in three), 3)) \
        == P.miller_loop(three)
    assert f12u(out(L.ht_miller_loop_fixed, 576, g1b(Pa) + g1b(None), g2b(None) + g2b(Qc), 2)) == P.F12_ONE
