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
    # inv() = safegcd (Bernstein-Yang divsteps), inv_bgcd() = binary extended GCD, inv_fermat() = a^(p-2):
    # all three against python, incl. edge values and values around the 30-bit limb boundaries
    for a in [1, 2, 3, M.Q - 1, M.Q - 2, (M.Q + 1) // 2, 1 << 380, 2 ** 30, 2 ** 30 - 1, 2 ** 60 + 1] + \
            [rng.randrange(1, M.Q) for _ in range(400)]:
        want = pow(a, -1, M.Q) * Rq % M.Q
        assert call(L.hm_fq_inv, b(a * Rq % M.Q, 48)) == want
        if a % 10 < 2:
            assert call(L.hm_fq_inv_bgcd, b(a * Rq % M.Q, 48)) == want
            assert call(L.hm_fq_inv_fermat, b(a * Rq % M.Q, 48)) == want
    for a in [1, 2, 3, M.R - 1, M.R - 2, (M.R + 1) // 2, 1 << 254, 2 ** 30, 2 ** 30 - 1] + \
            [rng.randrange(1, M.R) for _ in range(400)]:
        want = pow(a, -1, M.R) * Rr % M.R
        assert call(L.hm_fr_inv, b(a * Rr % M.R, 32), n=32) == want
        if a % 10 < 2:
            assert call(L.hm_fr_inv_bgcd, b(a * Rr % M.R, 32), n=32) == want
            assert call(L.hm_fr_inv_fermat, b(a * Rr % M.R, 32), n=32) == want
    for am in list(range(1, 100)) + [M.Q - k for k in range(1, 30)]:   # small raw Montgomery residues
        a = am * pow(Rq, -1, M.Q) % M.Q
        assert call(L.hm_fq_inv, b(am, 48)) == pow(a, -1, M.Q) * Rq % M.Q
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


def test_emit_host_path(L):
    """emit.h compiled for the host -- what the library runs on the calling thread for a lone host-bound MSM result
    (to_affine + serialisation of a de-normalised XYZZ point in the signed 30-bit form) -- against the oracle's encoders:
    affine Montgomery, zcash compressed / uncompressed, Jacobian (compared as a point); identity included; both y signs."""
    rng = random.Random(41)
    G = C.g1_generator()
    rinv = pow(M.FQ_MONT_R, -1, M.Q)
    seen_sign = set()
    for k in [0, 1, 2, 3, M.R - 1] + [rng.randrange(M.R) for _ in range(40)]:
        want_blob = C.g1_mul(G, k)
        want = C.blob_to_point(want_blob)
        out = {}
        for fmt, nbytes in ((0, 96), (1, 144), (2, 96), (3, 48)):
            o = ctypes.create_string_buffer(nbytes)
            L.hm_emit(G, b(k, 32), fmt, o)
            out[fmt] = o.raw
        assert out[0] == want_blob, k
        assert out[2] == M.g1_to_uncompressed(want), k
        assert out[3] == M.g1_to_compressed(want), k
        X, Y, Z = (int.from_bytes(out[1][48 * i:48 * i + 48], "little") * rinv % M.Q for i in range(3))
        if want is None:
            assert Z == 0
        else:
            zi = pow(Z, -1, M.Q)
            assert (X * zi * zi % M.Q, Y * zi * zi * zi % M.Q) == want, k
            seen_sign.add(out[3][0] & 0x20)
    assert seen_sign == {0, 0x20}


def test_fq30_unsaturated_layer(L):
    """field30.h / curve30.h (the signed 13 x 30-bit representation used by k_accum_affine) vs the oracle."""
    rng = random.Random(30)
    Rq = M.FQ_MONT_R
    ri = pow(Rq, -1, M.Q)
    vals = [0, 1, M.Q - 1, M.Q - 2, (1 << 380) % M.Q] + [rng.randrange(M.Q) for _ in range(200)]
    for a in vals:
        assert call(L.hm_packunpack30, b(a, 48)) == a
        assert call(L.hm_roundtrip30, b(a, 48)) == a
    for i in range(200):
        a, c = vals[i % len(vals)], vals[(7 * i + 3) % len(vals)]
        assert call(L.hm_mul30, b(a, 48), b(c, 48)) == a * c * ri % M.Q
        assert call(L.hm_sqr30, b(a, 48)) == a * a * ri % M.Q
    G, INF = C.g1_generator(), bytes(96)

    def chain(ps, signs):
        o, o2 = ctypes.create_string_buffer(96), ctypes.create_string_buffer(96)
        L.hm_madd30_chain(b"".join(ps), len(ps), ctypes.c_uint64(signs), o)
        L.hm_madd30_chain_kernel_form(b"".join(ps), len(ps), ctypes.c_uint64(signs), o2)   # lazy accumulator, as k_accum_affine
        assert o.raw == o2.raw
        return o.raw

    def ref(ps, signs):
        acc = INF
        for i, p in enumerate(ps):
            q = C.point_to_blob(M.g1_neg(C.blob_to_point(p))) if (signs >> i) & 1 else p
            acc = C.g1_add(acc, q)
        return acc

    pts = [C.g1_mul(G, rng.randrange(1, M.R)) for _ in range(8)]
    P = pts[0]
    cases = [([P, P], 0), ([P, P], 2), ([P, P, P], 0), ([P, P, P, P], 0b0110), ([P, INF, P], 0), ([INF, P], 0),
             ([INF, INF, P, P], 0b1000), ([P, P, P], 0b010), ([P] + [pts[1]] * 5, 0), ([P, P, pts[2], P, P], 0b11000)]
    for ps, s in cases:            # doubling, inverse, identity points, infinity mid-chain
        assert chain(ps, s) == ref(ps, s), (len(ps), s)
    for trial in range(8):         # random chains (the lazy-reduction bounds are exercised repeatedly)
        k = rng.randrange(2, 40) if trial < 6 else 64
        ps = [rng.choice(pts) if rng.random() < 0.3 else C.g1_mul(G, rng.randrange(1, M.R)) for _ in range(k)]
        s = rng.getrandbits(k)
        assert chain(ps, s) == ref(ps, s)


def test_fq30_general_add_and_double(L):
    """g1_add30 / g1_dbl30 (tail kernels of the MSM) on de-normalised, lazily-reduced operands."""
    rng = random.Random(31)
    G, INF = C.g1_generator(), bytes(96)
    P = C.g1_mul(G, rng.randrange(1, M.R))
    Q = C.g1_mul(G, rng.randrange(1, M.R))
    nP = C.point_to_blob(M.g1_neg(C.blob_to_point(P)))
    for a, c in [(P, Q), (P, P), (P, nP), (INF, P), (P, INF), (INF, INF), (Q, P)]:
        assert pt(L.hm_add30, a, c) == C.g1_add(a, c)
    for k in (0, 1, 2, 3, M.R - 1, rng.randrange(M.R)):
        assert pt(L.hm_mul30_scalar, P, b(k, 32)) == C.g1_mul(P, k)


def test_fq30_raw_limb_bounds(L):
    """mul30 / sqr30 / muladd30 on raw balanced limbs, including the extreme digits -2^29 and 2^29 that the
    64-bit column accumulators must survive (26 products of 2^58 per column; the fused multiply-add sets the
    high part aside in the five columns that hold more than 30): results are exact Montgomery quotients,
    normalised, and inside the documented magnitude bound."""
    import struct
    rng = random.Random(3030)
    N, B = 13, 30
    R30, H = 1 << (N * B), 1 << (B - 1)

    def val(l):
        return sum(v << (B * i) for i, v in enumerate(l))

    def raw(l):
        return struct.pack("<13i", *l)

    def out(fn, *args):
        o = ctypes.create_string_buffer(52)
        fn(*[raw(a) for a in args], o)
        return list(struct.unpack("<13i", o.raw))

    def limbs(kind):
        if kind == "max":
            l = [H - 1] * 12
        elif kind == "min":
            l = [-H] * 12
        elif kind == "neg_of_min":          # limb-wise negation of a normalised value: +2^29 digits
            l = [H] * 12
        elif kind == "alt":
            l = [(-H if i & 1 else H - 1) for i in range(12)]
        else:
            l = [rng.randrange(-H, H) for _ in range(12)]
        return l + [rng.randrange(-(1 << 27), 1 << 27)]   # |value| < 2^387 ~ 80 q

    def check(r, num, bound_q):
        assert all(-H <= v < H for v in r[:12]), r                       # normalised
        x = val(r)
        assert (x * R30 - num) % M.Q == 0                                 # exact Montgomery quotient
        assert abs(x) * 1000 <= M.Q * int(bound_q * 1000), (abs(x) / M.Q, bound_q)

    kinds = ["max", "min", "neg_of_min", "alt", "rnd", "rnd", "rnd"]
    qr = M.Q / R30
    for ka in kinds:
        for kb in kinds:
            a, c = limbs(ka), limbs(kb)
            A, Cv = val(a), val(c)
            check(out(L.hm_mul30_raw, a, c), A * Cv, 0.5001 + abs(A * Cv) / M.Q / M.Q * qr)
            e, f = limbs(kb), limbs(ka)
            E, F = val(e), val(f)
            check(out(L.hm_muladd30_raw, a, c, e, f), A * Cv + E * F, 0.5001 + (abs(A * Cv) + abs(E * F)) / M.Q / M.Q * qr)
        if ka != "neg_of_min":                 # sqr30 doubles its operand: needs the normalised range
            a = limbs(ka)
            check(out(L.hm_sqr30_raw, a), val(a) ** 2, 0.5001 + val(a) ** 2 / M.Q / M.Q * qr)
    # merged subtractions (mul30_sub, sqr30_sub2): exact a*b/R - c and a^2/R - c - 2e as integers mod q, normalised output
    for ka in kinds:
        for kb in kinds:
            a, c, u = limbs(ka), limbs(kb), limbs(kb if ka == "rnd" else ka)
            r = out(L.hm_mul30_sub_raw, a, c, u)
            assert all(-H <= v < H for v in r[:12]), r
            assert ((val(r) + val(u)) * R30 - val(a) * val(c)) % M.Q == 0
            assert abs(val(r)) <= abs(val(a) * val(c)) // R30 + M.Q // 2 + abs(val(u)) + 2
            if ka != "neg_of_min":
                e = limbs(kb)
                r = out(L.hm_sqr30_sub2_raw, a, u, e)
                assert all(-H <= v < H for v in r[:12]), r
                assert ((val(r) + val(u) + 2 * val(e)) * R30 - val(a) ** 2) % M.Q == 0
                assert abs(val(r)) <= val(a) ** 2 // R30 + M.Q // 2 + abs(val(u)) + 2 * abs(val(e)) + 2
    # unsigned-digit outputs (mul30u, sqr30_sub2u) and their use as ONE operand of the next product: balanced x unsigned at the
    # extremes (every unsigned digit 2^30 - 1 against every balanced digit +-2^29: the column bound 2^62.93 of field30.h)
    U = (1 << B) - 1

    def ulimbs(kind):
        if kind == "umax":
            l = [U] * 12
        elif kind == "uzero":
            l = [0] * 12
        else:
            l = [rng.randrange(0, 1 << B) for _ in range(12)]
        return l + [rng.randrange(-(1 << 23), 1 << 23)]

    for ka in kinds:
        for ku in ("umax", "umax", "urnd", "urnd", "uzero"):
            a, u = limbs(ka), ulimbs(ku)
            for fn in (L.hm_mul30u_raw, L.hm_mul30_raw):            # unsigned operand into both output flavours
                r = out(fn, a, u)
                assert ((val(r)) * R30 - val(a) * val(u)) % M.Q == 0, (ka, ku)
                assert abs(val(r)) <= abs(val(a) * val(u)) // R30 + M.Q // 2 + 2
                if fn is L.hm_mul30u_raw:
                    assert all(0 <= v < (1 << B) for v in r[:12]), r
                else:
                    assert all(-H <= v < H for v in r[:12]), r
            r = out(L.hm_mul30_sub_raw, a, u, ulimbs("umax"))       # unsigned operand and unsigned subtrahend
            assert all(-H <= v < H for v in r[:12])
            if ka != "neg_of_min":
                c, e = limbs(ka), ulimbs(ku)
                r = out(L.hm_sqr30_sub2u_raw, a, c, e)
                assert all(0 <= v < (1 << B) for v in r[:12]), r
                assert ((val(r) + val(c) + 2 * val(e)) * R30 - val(a) ** 2) % M.Q == 0
    # same-sign worst case for every column at once
    a = [H] * 12 + [1 << 20]
    check(out(L.hm_muladd30_raw, a, a, a, a), 2 * val(a) ** 2, 0.5001 + 2 * val(a) ** 2 / M.Q / M.Q * qr)
    na = [-H] * 12 + [-(1 << 20)]
    check(out(L.hm_muladd30_raw, a, na, a, na), 2 * val(a) * val(na), 0.5001 + 2 * val(a) ** 2 / M.Q / M.Q * qr)
    # normalize30: any limbs below 2^31 - 2^29 in magnitude -> the unique normalised form of the same integer
    for _ in range(50):
        l = [rng.randrange(-3 * H + 1, 3 * H) for _ in range(12)] + [rng.randrange(-(1 << 20), 1 << 20)]
        r = out(L.hm_normalize30_raw, l)
        assert val(r) == val(l) and all(-H <= v < H for v in r[:12])
    # from30 on lazy values up to 256 q in magnitude, either sign: canonical x / R30 * R384
    Rq = M.FQ_MONT_R
    for _ in range(50):
        x = rng.randrange(-255 * M.Q, 255 * M.Q)
        l, t = [], x
        for _i in range(12):
            d = ((t + H) % (1 << B)) - H
            l.append(d)
            t = (t - d) >> B
        l.append(t)
        o = ctypes.create_string_buffer(48)
        L.hm_from30_raw(raw(l), o)
        assert int.from_bytes(o.raw, "little") == x * pow(R30, -1, M.Q) * Rq % M.Q


def test_fr29_ntt_arithmetic(L):
    """fr29.h: the unsaturated 9 x 29-bit Fr of the NTT kernels -- multiply against any 256-bit input, and up to
    12 consecutive lazy butterflies (the worst-case growth inside one LDS tile) followed by the closing multiply."""
    rng = random.Random(5)
    R, R256 = M.R, 1 << 256
    for it in range(300):
        x = rng.randrange(R256) if it % 3 else rng.randrange(R)
        w = rng.randrange(R)
        if it < 4:
            x, w = [0, R256 - 1, R - 1, 1][it], [5, R - 1, R - 1, 0][it]
        o = ctypes.create_string_buffer(32)
        L.hm_fr29_mul(b(x, 32), b(w * R256 % R, 32), o)
        assert int.from_bytes(o.raw, "little") == x * w % R
    for stages in (1, 2, 5, 10, 12):
        for _ in range(20):
            u, v, w = rng.randrange(R), rng.randrange(R), rng.randrange(R)
            ou, ov = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
            L.hm_fr29_butterflies(b(u, 32), b(v, 32), b(w * R256 % R, 32), stages, ou, ov)
            U, V = u, v
            for _s in range(stages):
                t = V * w % R
                U, V = (U + t) % R, (U - t) % R
            assert int.from_bytes(ou.raw, "little") == U and int.from_bytes(ov.raw, "little") == V


def test_fr29_quotient_kernel_thread(L):
    """poly.hip's quotient kernels keep unreduced sums of Shoup products in 29-bit limbs (kzg_amd/csrc/fr29.h primitives; the
    reference's long_division, src/polynomial.rs:193-227, reduces after every step): one thread's sequence -- Horner of eight raw
    256-bit coefficients, up to ten scan steps adding products of neighbours' values as large as the kernels can hold (25 r), the
    canonical value, one output step -- against big-integer arithmetic, at random and at the largest operands."""
    rng = random.Random(291)
    R, R256 = M.R, 1 << 256
    MASK = (1 << 29) - 1
    for it in range(300):
        big = it < 6
        a = [R256 - 1 if big else rng.randrange(R256) for _ in range(8)]
        x = [R - 1, 1, 0, R - 2, 2, R - 1][it] if big else rng.randrange(R)
        p = R - 1 if big else rng.randrange(R)
        m = 10 if big else rng.randrange(0, 11)
        nbv = [25 * R - 1 - i if big else rng.randrange(25 * R) for i in range(m)]
        nb = (ctypes.c_uint32 * (9 * max(m, 1)))()
        for i, v in enumerate(nbv):         # normalised: limbs below 2^29, the excess in the top limb
            for j in range(9):
                nb[9 * i + j] = (v >> (29 * j)) & MASK if j < 8 else v >> 232
        a_next = rng.randrange(R) if not big else R - 1
        o_scan, o_next, top = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32), ctypes.c_uint32()
        L.hm_fr29_quotient_thread(b"".join(b(v, 32) for v in a), b(x * R256 % R, 32), b(p * R256 % R, 32), nb, m, b(a_next, 32), o_scan, o_next,
                                  ctypes.byref(top))
        want = (sum(c * pow(x, k, R) for k, c in enumerate(a)) + p * sum(nbv)) % R
        assert int.from_bytes(o_scan.raw, "little") == want, it
        assert int.from_bytes(o_next.raw, "little") == (a_next + want * x) % R, it
        assert top.value < 64 * (R >> 232), it      # what fr29_reduce_below_2r accepts


def test_fr29_shoup_multiply_and_lazy_radix4(L):
    """fr29.h, the NTT's multiplication by constants (Shoup: w with wp = floor(w 2^261 / r)) and the lazy radix-4 butterflies:
    (w, wp) from the table builder against big-integer arithmetic; the product exact mod r and below 2r for normalised inputs,
    for unnormalised sums with every limb at the 1.5 * 2^30 bound, and for values up to 2^261 - 1; then six stage pairs (the 12
    stages of the largest LDS tile) along the never-multiplied element chain taking the fastest-growing output each time."""
    rng = random.Random(29)
    R, R256, B261 = M.R, 1 << 256, 1 << 261
    MASK = (1 << 29) - 1

    def limbs(v):
        return (ctypes.c_uint32 * 9)(*[(v >> (29 * i)) & MASK for i in range(9)])

    def val(ls):
        return sum(int(x) << (29 * i) for i, x in enumerate(ls))

    for it in range(400):
        w = rng.randrange(R) if it > 3 else [0, 1, R - 1, 7][it]
        out, wl, wpl = (ctypes.c_uint32 * 9)(), (ctypes.c_uint32 * 9)(), (ctypes.c_uint32 * 9)()
        kind = it % 4
        if kind == 0:      # normalised, below 2^256 (what a tile load produces)
            x = rng.randrange(R256)
            xl = limbs(x)
        elif kind == 1:    # any value below 2^261
            x = rng.randrange(B261) if it > 8 else B261 - 1
            xl = limbs(x)
        elif kind == 2:    # unnormalised: every limb up to 1.5 * 2^30 (u + 4r - t), value kept below 2^261
            raw = [rng.randrange(3 << 29) for _ in range(8)] + [rng.randrange(1 << 27)]
            if it < 12:
                raw = [(3 << 29) - 1] * 8 + [(1 << 27) - 1]
            xl = (ctypes.c_uint32 * 9)(*raw)
            x = val(raw)
            assert x < B261
        else:
            x = rng.randrange(R)
            xl = limbs(x)
        L.hm_fr29_shoup_raw(xl, b(w * R256 % R, 32), out, wl, wpl)
        assert val(wl) == w and val(wpl) == (w << 261) // R
        assert all(v <= MASK for v in out)
        got = val(out)
        assert got % R == x * w % R and got < 2 * R, (it, kind)
    for which in (3, 0, 1, 2):
        for pairs in (1, 5, 6):
            for _ in range(6):
                x0 = rng.randrange(R256)
                xs = [rng.randrange(R256) for _ in range(3 * pairs)]
                ws = [rng.randrange(R) for _ in range(3 * pairs)]
                o = ctypes.create_string_buffer(32)
                L.hm_fr29_radix4_chain(b(x0, 32), b"".join(b(v, 32) for v in xs), b"".join(b(v * R256 % R, 32) for v in ws), pairs, which, o)
                X0 = x0
                for p in range(pairs):
                    x1, x2, x3 = xs[3 * p: 3 * p + 3]
                    a, bb, c = ws[3 * p: 3 * p + 3]
                    t1, t3 = x1 * a, x3 * a
                    s0, y1, s2, y3 = X0 + t1, X0 - t1, x2 + t3, x2 - t3
                    t2, t3b = s2 * bb, y3 * c
                    X0 = [s0 + t2, y1 + t3b, s0 - t2, y1 - t3b][which] % R
                assert int.from_bytes(o.raw, "little") == X0


def test_naf18_recoding(L):
    """naf.h: the width-18 NAF digits of the positional tables.  k = sum d 2^p exactly, digits odd with |d| < 2^17, positions at
    least 18 apart and below 255, at most 15 of them, sign combined with the flip of a balanced scalar; edge patterns: 0, 1, runs
    of ones (a carry through the whole run), alternating bits, the largest balanced value, digits at the very top."""
    import struct
    rng = random.Random(1818)

    def recode(k, flip=0):
        out = (ctypes.c_uint32 * 15)()
        cnt = L.hm_naf18(k.to_bytes(32, "little"), flip, out)
        return [out[i] for i in range(cnt)]

    cases = [0, 1, 2, 3, (1 << 17) - 1, 1 << 17, (1 << 17) + 1, (1 << 18) - 1, 1 << 18, (1 << 254) - 1, (1 << 253) + 1, (1 << 254) - (1 << 200),
             int("aaaaaaaa" * 8, 16) >> 2, int("55555555" * 8, 16) >> 2, (1 << 64) - 1, (1 << 128) - 1, ((1 << 254) - 1) ^ ((1 << 100) - 1),
             (M.R - 1) // 2, M.R - (M.R >> 1), (1 << 236) * ((1 << 17) + 1), (1 << 237) * ((1 << 17) - 1)]
    cases += [rng.randrange(1 << 254) for _ in range(3000)] + [rng.getrandbits(64) for _ in range(300)]
    cases += [rng.getrandbits(rng.randrange(1, 254)) for _ in range(1000)]
    total = 0
    for k in cases:
        assert k < (1 << 254)
        for flip in (0, 1):
            recs = recode(k, flip)
            assert len(recs) <= 15
            val, last = 0, -18
            for r in recs:
                assert r >> 31 == 1
                idx, sign, p = r & 0x1ffff, (r >> 17) & 1, (r >> 18) & 0xff
                mag = 2 * idx + 1
                assert mag < (1 << 17) and p - last >= 18 and p <= 254
                last = p
                val += (-mag if sign ^ flip else mag) << p
            assert val == k, hex(k)
        total += len(recs)
    assert total / len(cases) < 14.2


def test_shared_arithmetic_is_free_of_undefined_behaviour(tmp_path_factory):
    """The field / curve / NTT-arithmetic headers are one source for the host and for gfx950 (signed 30-bit limbs, arithmetic right
    shifts of 64-bit accumulators, 29-bit lazy limbs ...).  GPU sanitizers are not available on the test pool, so the CPU build runs
    under UBSan (-fsanitize=undefined, no recovery: the first signed overflow, out-of-range shift or misaligned access aborts the
    child process) through every test of this module."""
    import sys
    so = str(tmp_path_factory.mktemp("hm_ubsan") / "libhostmath_ubsan.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-fsanitize=undefined", "-fno-sanitize-recover=undefined",
                           "-o", so, os.path.join(ROOT, "tests", "host_math.cpp")])
    code = ("import ctypes, inspect, sys; sys.path.insert(0, %r); import tests.test_host_math as T; L = ctypes.CDLL(%r); n = 0\n"
            "for name in sorted(dir(T)):\n"
            "    f = getattr(T, name)\n"
            "    if name.startswith('test_') and list(inspect.signature(f).parameters) == ['L']:\n"
            "        f(L); n += 1\n"
            "print('UBSAN-CLEAN', n)\n" % (ROOT, so))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "runtime error" not in r.stderr, r.stderr[-3000:]
    assert "UBSAN-CLEAN" in r.stdout and int(r.stdout.split()[-1]) >= 8
