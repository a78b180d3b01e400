"""Device field / curve arithmetic (kzg_amd/csrc/field.h, curve.h as compiled for gfx950) vs the oracle."""
import ctypes
import random

import pytest

from oracle import c_oracle as C
from oracle import kzg_model as M

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine(hooks_engine):  # noqa: F811  the unit-test hooks live in the -DKZG_TEST_HOOKS build of the library, not in the product
    return hooks_engine


def _call(engine, fname, n, *bufs, out_elem):
    out = ctypes.create_string_buffer(n * out_elem)
    f = getattr(engine.lib, fname)
    rc = f(engine.ctx, *bufs, n, out)
    assert rc == 0, engine.last_error()
    return out.raw


def test_fr_fq_mul(engine):
    rng = random.Random(11)
    n = 1000
    Rr, Rq = M.FR_MONT_R, M.FQ_MONT_R
    a = [rng.randrange(M.R) for _ in range(n)]
    b = [rng.randrange(M.R) for _ in range(n)]
    a[:4], b[:4] = [0, M.R - 1, 1, M.R - 1], [5, M.R - 1, 0, 1]
    out = _call(engine, "kzg_test_fr_mul", n, b"".join(x.to_bytes(32, "little") for x in a),
                b"".join(x.to_bytes(32, "little") for x in b), out_elem=32)
    rinv = pow(Rr, -1, M.R)
    for i in range(n):
        assert int.from_bytes(out[32 * i:32 * i + 32], "little") == a[i] * b[i] * rinv % M.R
    a = [rng.randrange(M.Q) for _ in range(n)]
    b = [rng.randrange(M.Q) for _ in range(n)]
    a[:4], b[:4] = [0, M.Q - 1, 1, M.Q - 1], [5, M.Q - 1, 0, 1]
    out = _call(engine, "kzg_test_fq_mul", n, b"".join(x.to_bytes(48, "little") for x in a),
                b"".join(x.to_bytes(48, "little") for x in b), out_elem=48)
    rinv = pow(Rq, -1, M.Q)
    for i in range(n):
        assert int.from_bytes(out[48 * i:48 * i + 48], "little") == a[i] * b[i] * rinv % M.Q


def test_fr_inv(engine):
    rng = random.Random(12)
    n = 300
    Rr = M.FR_MONT_R
    a = [rng.randrange(1, M.R) for _ in range(n)]
    a[0], a[1], a[2] = 1, M.R - 1, 0
    out = ctypes.create_string_buffer(n * 32)
    rc = engine.lib.kzg_test_fr_inv(engine.ctx, b"".join((x * Rr % M.R).to_bytes(32, "little") for x in a), n, out)
    assert rc == 0, engine.last_error()
    for i in range(n):
        want = 0 if a[i] == 0 else pow(a[i], -1, M.R) * Rr % M.R
        assert int.from_bytes(out.raw[32 * i:32 * i + 32], "little") == want


def test_g1_add_edge_cases(engine):
    """P+Q, P+P (doubling branch), P+(-P), inf+P, P+inf, inf+inf -- mixed and general addition."""
    rng = random.Random(13)
    G, INF = C.g1_generator(), bytes(96)
    pts = [C.g1_mul(G, rng.randrange(1, M.R)) for _ in range(20)]
    A, B = [], []
    for i in range(0, 20, 2):
        A.append(pts[i]); B.append(pts[i + 1])
    P = pts[0]
    nP = C.point_to_blob(M.g1_neg(C.blob_to_point(P)))
    A += [P, P, INF, P, INF]
    B += [P, nP, P, INF, INF]
    n = len(A)
    out = _call(engine, "kzg_test_g1_add", n, b"".join(A), b"".join(B), out_elem=96)
    for i in range(n):
        assert out[96 * i:96 * i + 96] == C.g1_add(A[i], B[i]), f"case {i}"


def test_g1_scalar_mul(engine):
    rng = random.Random(14)
    G = C.g1_generator()
    ks = [0, 1, 2, M.R - 1] + [rng.randrange(M.R) for _ in range(28)]
    P = C.g1_mul(G, 123456789)
    n = len(ks)
    out = _call(engine, "kzg_test_g1_mul", n, P * n, b"".join(k.to_bytes(32, "little") for k in ks), out_elem=96)
    for i, k in enumerate(ks):
        assert out[96 * i:96 * i + 96] == C.g1_mul(P, k)
