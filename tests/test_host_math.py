"""CPU unit tests of kzg_amd/csrc/{field,curve}.h -- the same arithmetic source hipcc compiles for
gfx950 -- against the oracle (python model + C restatement)."""
import ctypes
import os
import random
import subprocess

import pytest

from oracle import kzg_model as M, c_oracle as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("hm") / "libhostmath.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so,
                           os.path.join(ROOT, "tests", "host_math.cpp")])
    return ctypes.CDLL(so)


def b(x, n):
    return x.to_bytes(n, "little")


def call(f, *args, n=48):
    o = ctypes.create_string_buffer(n)
    f(*args, o)
    return int.from_bytes(o.raw, "little")


def pt(f, *args):
    o = ctypes.create_string_buffer(96)
    f(*args, o)
    return o.raw


def test_field_ops(L):
    rng = random.Random(3)
    Rq, Rr = M.FQ_MONT_R, M.FR_MONT_R
    edge_q = [(0, 5), (M.Q - 1, M.Q - 1), (1, 0), (M.Q - 1, 1)]
    edge_r = [(0, 5), (M.R - 1, M.R - 1), (1, 0), (M.R - 1, 1)]
    for it in range(300):
        a, c = edge_q[it] if it < 4 else (rng.randrange(M.Q), rng.randrange(M.Q))
        assert call(L.hm_fq_mul, b(a, 48), b(c, 48)) == a * c * pow(Rq, -1, M.Q) % M.Q
        assert call(L.hm_fq_add, b(a, 48), b(c, 48)) == (a + c) % M.Q
        assert call(L.hm_fq_sub, b(a, 48), b(c, 48)) == (a - c) % M.Q
        a, c = edge_r[it] if it < 4 else (rng.randrange(M.R), rng.randrange(M.R))
        assert call(L.hm_fr_mul, b(a, 32), b(c, 32), n=32) == a * c * pow(Rr, -1, M.R) % M.R
        assert call(L.hm_fr_add, b(a, 32), b(c, 32), n=32) == (a + c) % M.R
        assert call(L.hm_fr_sub, b(a, 32), b(c, 32), n=32) == (a - c) % M.R
        assert call(L.hm_fr_to_mont, b(a, 32), n=32) == a * Rr % M.R
        assert call(L.hm_fr_from_mont, b(a, 32), n=32) == a * pow(Rr, -1, M.R) % M.R
    # inv() = binary extended GCD, inv_fermat() = a^(p-2): both against python, incl. edge values
    for a in [1, 2, 3, M.Q - 1, M.Q - 2, (M.Q + 1) // 2, 1 << 380] + [rng.randrange(1, M.Q) for _ in range(40)]:
        want = pow(a, -1, M.Q) * Rq % M.Q
        assert call(L.hm_fq_inv, b(a * Rq % M.Q, 48)) == want
        assert call(L.hm_fq_inv_fermat, b(a * Rq % M.Q, 48)) == want
    for a in [1, 2, 3, M.R - 1, M.R - 2, (M.R + 1) // 2, 1 << 254] + [rng.randrange(1, M.R) for _ in range(40)]:
        want = pow(a, -1, M.R) * Rr % M.R
        assert call(L.hm_fr_inv, b(a * Rr % M.R, 32), n=32) == want
        assert call(L.hm_fr_inv_fermat, b(a * Rr % M.R, 32), n=32) == want
    assert call(L.hm_fq_inv, b(0, 48)) == 0 and call(L.hm_fr_inv, b(0, 32), n=32) == 0
    assert call(L.hm_fr_root_of_unity, n=32) == M.FR_ROOT_OF_UNITY


def test_g1_ops(L):
    rng = random.Random(4)
    assert pt(L.hm_g1_generator) == C.g1_generator()
    G, INF = C.g1_generator(), bytes(96)
    P = C.g1_mul(G, rng.randrange(M.R))
    Qp = C.g1_mul(G, rng.randrange(M.R))
    nP = C.point_to_blob(M.g1_neg(C.blob_to_point(P)))
    for f in (L.hm_g1_madd, L.hm_g1_add):
        assert pt(f, P, Qp) == C.g1_add(P, Qp)
        assert pt(f, P, P) == C.g1_add(P, P)          # doubling branch
        assert pt(f, P, nP) == INF                     # P + (-P)
        assert pt(f, INF, P) == P and pt(f, P, INF) == P and pt(f, INF, INF) == INF
    for k in (0, 1, 2, M.R - 1, rng.randrange(M.R)):
        assert pt(L.hm_g1_mul, P, b(k, 32)) == C.g1_mul(P, k)
        assert pt(L.hm_g1_jac_roundtrip, P, b(k, 32)) == C.g1_mul(P, k)
    assert L.hm_g1_on_curve(P) == 1 and L.hm_g1_on_curve(INF) == 1
    assert L.hm_g1_on_curve(P[:95] + bytes([P[95] ^ 1])) == 0
