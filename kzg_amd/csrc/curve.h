// curve.h -- BLS12-381 G1 (y^2 = x^3 + 4 over Fq) point arithmetic in extended Jacobian ("XYZZ")
// coordinates: x = X/ZZ, y = Y/ZZZ with ZZ^3 = ZZZ^2.  Mixed addition costs 8M + 2S, which is why
// the bucket accumulators use it.  Stands in for blstrs::{G1Affine, G1Projective} (external to the
// reference, Cargo.toml:27); results leave the engine only as canonical affine coordinates, so the
// internal representation cannot affect parity.
#pragma once
#include "field.h"

namespace kzg {

struct G1Affine {  // identity: x = y = 0 (blst_p1_affine convention)
    Fq x, y;
    KZG_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
    static KZG_HD G1Affine inf() {
        G1Affine p;
        p.x = Fq::zero();
        p.y = Fq::zero();
        return p;
    }
};

struct G1Xyzz {  // identity: zz = 0
    Fq x, y, zz, zzz;
    KZG_HD bool is_inf() const { return zz.is_zero(); }
    static KZG_HD G1Xyzz inf() {
        G1Xyzz p;
        p.x = Fq::zero();
        p.y = Fq::zero();
        p.zz = Fq::zero();
        p.zzz = Fq::zero();
        return p;
    }
    static KZG_HD G1Xyzz from_affine(const G1Affine &a) {
        G1Xyzz p;
        if (a.is_inf()) return inf();
        p.x = a.x;
        p.y = a.y;
        p.zz = Fq::one();
        p.zzz = Fq::one();
        return p;
    }
};

struct G1Jacobian {  // blst_p1: identity Z = 0
    Fq x, y, z;
};

KZG_HD G1Affine g1_neg(const G1Affine &a) {
    G1Affine r;
    r.x = a.x;
    r.y = a.y.is_zero() ? a.y : neg(a.y);
    return r;
}

// dbl-2008-s-1 (a = 0): 6M + 3S
KZG_HD G1Xyzz g1_dbl(const G1Xyzz &p) {
    if (p.is_inf() || p.y.is_zero()) return G1Xyzz::inf();
    Fq U = dbl(p.y);
    Fq V = sqr(U);
    Fq W = mul(U, V);
    Fq S = mul(p.x, V);
    Fq X2 = sqr(p.x);
    Fq M = add(dbl(X2), X2);
    G1Xyzz r;
    r.x = sub(sqr(M), dbl(S));
    r.y = sub(mul(M, sub(S, r.x)), mul(W, p.y));
    r.zz = mul(V, p.zz);
    r.zzz = mul(W, p.zzz);
    return r;
}

KZG_HD G1Xyzz g1_dbl_affine(const G1Affine &a) {
    return g1_dbl(G1Xyzz::from_affine(a));
}

// madd-2008-s: acc += a  (8M + 2S), all special cases handled
KZG_HD G1Xyzz g1_madd(const G1Xyzz &p, const G1Affine &a) {
    if (a.is_inf()) return p;
    if (p.is_inf()) return G1Xyzz::from_affine(a);
    Fq U2 = mul(a.x, p.zz);
    Fq S2 = mul(a.y, p.zzz);
    Fq Pp = sub(U2, p.x);
    Fq R = sub(S2, p.y);
    if (Pp.is_zero()) {
        if (R.is_zero()) return g1_dbl_affine(a);
        return G1Xyzz::inf();
    }
    Fq PP = sqr(Pp);
    Fq PPP = mul(Pp, PP);
    Fq Q = mul(p.x, PP);
    G1Xyzz r;
    r.x = sub(sub(sqr(R), PPP), dbl(Q));
    r.y = sub(mul(R, sub(Q, r.x)), mul(p.y, PPP));
    r.zz = mul(p.zz, PP);
    r.zzz = mul(p.zzz, PPP);
    return r;
}

// add-2008-s: 12M + 2S
KZG_HD G1Xyzz g1_add(const G1Xyzz &p, const G1Xyzz &q) {
    if (q.is_inf()) return p;
    if (p.is_inf()) return q;
    Fq U1 = mul(p.x, q.zz);
    Fq U2 = mul(q.x, p.zz);
    Fq S1 = mul(p.y, q.zzz);
    Fq S2 = mul(q.y, p.zzz);
    Fq Pp = sub(U2, U1);
    Fq R = sub(S2, S1);
    if (Pp.is_zero()) {
        if (R.is_zero()) return g1_dbl(p);
        return G1Xyzz::inf();
    }
    Fq PP = sqr(Pp);
    Fq PPP = mul(Pp, PP);
    Fq Q = mul(U1, PP);
    G1Xyzz r;
    r.x = sub(sub(sqr(R), PPP), dbl(Q));
    r.y = sub(mul(R, sub(Q, r.x)), mul(S1, PPP));
    r.zz = mul(mul(p.zz, q.zz), PP);
    r.zzz = mul(mul(p.zzz, q.zzz), PPP);
    return r;
}

// Curve::to_affine (src/coeff_form.rs:63,78,107; src/eval_form.rs:120,139): one Fq inversion.
// 1/ZZ = ZZ^2 / ZZZ^2 (since ZZ^3 = ZZZ^2), so a single inverse of ZZZ serves both coordinates.
KZG_HD G1Affine g1_to_affine(const G1Xyzz &p) {
    if (p.is_inf()) return G1Affine::inf();
    Fq izzz = inv(p.zzz);
    Fq izz = mul(sqr(p.zz), sqr(izzz));
    G1Affine r;
    r.x = mul(p.x, izz);
    r.y = mul(p.y, izzz);
    return r;
}

// to_affine given a precomputed 1/ZZZ (batch inversion)
KZG_HD G1Affine g1_to_affine_with_inv(const G1Xyzz &p, const Fq &izzz) {
    if (p.is_inf()) return G1Affine::inf();
    Fq izz = mul(sqr(p.zz), sqr(izzz));
    G1Affine r;
    r.x = mul(p.x, izz);
    r.y = mul(p.y, izzz);
    return r;
}

// XYZZ -> Jacobian without inversion: Z = ZZ*ZZZ gives Z^2 = ZZ^5, Z^3 = ZZZ^5.
KZG_HD G1Jacobian g1_to_jacobian(const G1Xyzz &p) {
    G1Jacobian r;
    if (p.is_inf()) {
        r.x = Fq::zero();
        r.y = Fq::zero();
        r.z = Fq::zero();
        return r;
    }
    Fq zz2 = sqr(p.zz), zzz2 = sqr(p.zzz);
    r.x = mul(p.x, sqr(zz2));
    r.y = mul(p.y, sqr(zzz2));
    r.z = mul(p.zz, p.zzz);
    return r;
}

KZG_HD G1Xyzz g1_from_jacobian(const G1Jacobian &p) {
    if (p.z.is_zero()) return G1Xyzz::inf();
    G1Xyzz r;
    r.x = p.x;
    r.y = p.y;
    r.zz = sqr(p.z);
    r.zzz = mul(r.zz, p.z);
    return r;
}

KZG_HD bool g1_on_curve(const G1Affine &a) {
    if (a.is_inf()) return true;
    Fq four = from_u64<FqParams>(4);
    return sqr(a.y) == add(mul(sqr(a.x), a.x), four);
}

KZG_HD G1Affine g1_generator() {
    constexpr uint32_t GX[12] = {0xdb22c6bbu, 0xfb3af00au, 0xf97a1aefu, 0x6c55e83fu, 0x171bac58u, 0xa14e3a3fu,
                                 0x9774b905u, 0xc3688c4fu, 0x4fa9ac0fu, 0x2695638cu, 0x3197d794u, 0x17f1d3a7u};
    constexpr uint32_t GY[12] = {0x46c5e7e1u, 0x0caa2329u, 0xa2888ae4u, 0xd03cc744u, 0x2c04b3edu, 0x00db18cbu,
                                 0xd5d00af6u, 0xfcf5e095u, 0x741d8ae4u, 0xa09e30edu, 0xe3aaa0f1u, 0x08b3f481u};
    Fq x, y;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        x.v[i] = GX[i];
        y.v[i] = GY[i];
    }
    G1Affine g;
    g.x = to_mont(x);
    g.y = to_mont(y);
    return g;
}

// [k]P for a canonical 256-bit k (8 x u32 LE), double-and-add (vartime); used off the hot path
KZG_HD G1Xyzz g1_scalar_mul(const G1Affine &p, const uint32_t k[8]) {
    G1Xyzz acc = G1Xyzz::inf();
    for (int i = 255; i >= 0; i--) {
        acc = g1_dbl(acc);
        if ((k[i >> 5] >> (i & 31)) & 1) acc = g1_madd(acc, p);
    }
    return acc;
}

}  // namespace kzg
