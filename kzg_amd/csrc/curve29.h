// curve29.h -- XYZZ mixed addition on the unsaturated 29-bit Fq representation (field29.h), the inner
// loop of the MSM accumulation kernel.  Same formulas as g1_madd (madd-2008-s, 8M + 2S); what changes is
// the bookkeeping: values are lazily reduced, every subtraction adds a multiple of q chosen from the
// proven bound of its subtrahend, and only the operands of multiplications are limb-normalised.
//
// Bounds (M = "Montgomery product", < 1.0001 q because R29 = 2^406 >> q):
//   table point x2, y2 < 2q;  -y2 = 4q - y2 < 4q
//   acc.zz, acc.zzz : M            acc.x < 19q            acc.y < 11q
//   P  = U2 - X1 + 32q < 35q       R  = S2 - Y1 + 16q < 19q
//   X3 = R^2 - (PPP + 2Q) + 16q < 19q      Q - X3 + 32q < 35q      Y3 = (R(Q - X3) + (16q - Y1)*PPP)/R29 < 1.0001q
// General addition / doubling (g1_add29, g1_dbl29) accept any X < 20q, Y < 12q (ZZ, ZZZ are always
// products) and return X, Y < 5.0001q, so every mix of the three operations stays inside the classes.
// All far below the 2^12 q limit of mul29.
#pragma once
#include "curve.h"
#include "field29.h"

namespace kzg {

struct G1Affine29 {  // table entry: x, y < 2q normalised; identity = all limbs zero
    Fq29 x, y;
    KZG_HD bool is_inf() const { return x.limbs_all_zero() && y.limbs_all_zero(); }
};

struct alignas(16) G1Xyzz29 {  // 240 B: what the MSM partial-sum buffers hold
    Fq29 x, y, zz, zzz;
    uint32_t inf;
    uint32_t pad[3];
    static KZG_HD G1Xyzz29 infinity() {
        G1Xyzz29 p;
#pragma unroll
        for (int i = 0; i < F29_N; i++) p.x.v[i] = p.y.v[i] = p.zz.v[i] = p.zzz.v[i] = 0;
        p.inf = 1;
        p.pad[0] = p.pad[1] = p.pad[2] = 0;
        return p;
    }
};

KZG_HD G1Affine29 g1_affine_to29(const G1Affine &a) {
    G1Affine29 r;
    if (a.is_inf()) {
#pragma unroll
        for (int i = 0; i < F29_N; i++) r.x.v[i] = r.y.v[i] = 0;
        return r;
    }
    r.x = to29(a.x);
    r.y = to29(a.y);
    return r;
}

KZG_HD Fq29 zero29() {
    Fq29 z;
#pragma unroll
    for (int i = 0; i < F29_N; i++) z.v[i] = 0;
    return z;
}

KZG_HD G1Xyzz29 g1_from_affine29(const G1Affine29 &a, bool negate) {
    G1Xyzz29 p;
    p.inf = a.is_inf() ? 1u : 0u;
    p.pad[0] = p.pad[1] = p.pad[2] = 0;
    p.x = a.x;
    p.y = negate ? sub29<4>(zero29(), a.y) : a.y;
    p.zz = one29();
    p.zzz = one29();
    return p;
}

KZG_HD G1Xyzz g1_xyzz_from29(const G1Xyzz29 &p) {
    if (p.inf) return G1Xyzz::inf();
    G1Xyzz r;
    r.x = from29(p.x);
    r.y = from29(p.y);
    r.zz = from29(p.zz);
    r.zzz = from29(p.zzz);
    if (r.zz.is_zero()) return G1Xyzz::inf();
    return r;
}

KZG_HD G1Xyzz29 g1_xyzz_to29(const G1Xyzz &p) {
    G1Xyzz29 r;
    r.inf = p.is_inf() ? 1u : 0u;
    r.pad[0] = r.pad[1] = r.pad[2] = 0;
    r.x = to29(p.x);
    r.y = to29(p.y);
    r.zz = to29(p.zz);
    r.zzz = to29(p.zzz);
    return r;
}

// a * 2 and a * 3 (limb-wise, then normalised)
KZG_HD Fq29 times2_29(const Fq29 &a) {
    Fq29 r;
#pragma unroll
    for (int i = 0; i < F29_N; i++) r.v[i] = a.v[i] << 1;
    return normalize29(r);
}
KZG_HD Fq29 times3_29(const Fq29 &a) {
    Fq29 r;
#pragma unroll
    for (int i = 0; i < F29_N; i++) r.v[i] = a.v[i] * 3u;
    return normalize29(r);
}

// dbl-2008-s-1 (a = 0): 6M + 3S.  In: X < 20q, Y < 12q.  Out: X, Y < 5.0001q.
KZG_HD G1Xyzz29 g1_dbl29(const G1Xyzz29 &p) {
    if (p.inf) return p;
    Fq29 U = times2_29(p.y);              // < 24q
    Fq29 V = sqr29(U);
    if (is_zero_mod_q_product(V)) return G1Xyzz29::infinity();  // y == 0: a point of order two
    Fq29 W = mul29(U, V);
    Fq29 S = mul29(p.x, V);
    Fq29 Mm = times3_29(sqr29(p.x));      // < 3.0003q
    G1Xyzz29 r;
    r.inf = 0;
    r.pad[0] = r.pad[1] = r.pad[2] = 0;
    r.x = sub29<4>(sqr29(Mm), times2_29(S));                       // < 5.0001q
    r.y = sub29<4>(mul29(Mm, sub29<8>(S, r.x)), mul29(W, p.y));    // < 5.0001q
    r.zz = mul29(V, p.zz);
    r.zzz = mul29(W, p.zzz);
    return r;
}

// add-2008-s: 12M + 2S.  In: X < 20q, Y < 12q on both sides.  Out: X, Y < 5.0001q.
KZG_HD G1Xyzz29 g1_add29(const G1Xyzz29 &p, const G1Xyzz29 &q) {
    if (q.inf) return p;
    if (p.inf) return q;
    Fq29 U1 = mul29(p.x, q.zz);
    Fq29 U2 = mul29(q.x, p.zz);
    Fq29 S1 = mul29(p.y, q.zzz);
    Fq29 S2 = mul29(q.y, p.zzz);
    Fq29 Pp = sub29<4>(U2, U1);           // < 5.0001q
    Fq29 R = sub29<4>(S2, S1);
    Fq29 PP = sqr29(Pp);
    if (is_zero_mod_q_product(PP)) {
        if (is_zero_mod_q_product(sqr29(R))) return g1_dbl29(p);
        return G1Xyzz29::infinity();
    }
    Fq29 PPP = mul29(Pp, PP);
    Fq29 Q = mul29(U1, PP);
    G1Xyzz29 r;
    r.inf = 0;
    r.pad[0] = r.pad[1] = r.pad[2] = 0;
    r.zz = mul29(mul29(p.zz, q.zz), PP);
    r.zzz = mul29(mul29(p.zzz, q.zzz), PPP);
    Fq29 Bm = mul29(S1, PPP);
    r.x = sub29<4>(sqr29(R), add2x29(PPP, Q));                     // t < 3.0003q; X3 < 5.0001q
    r.y = sub29<4>(mul29(R, sub29<8>(Q, r.x)), Bm);                // < 5.0001q
    return r;
}

// acc += (negate ? -a : a), split in two so the caller can re-use the registers of `a` for the next
// gather as soon as the two products that read it are done:
//   phase 1: U2 = x2 * ZZ1, S2 = (+-y2) * ZZZ1              (the only uses of the affine point)
//   phase 2: everything else; `reload` re-fetches the affine point in the rare doubling case.
struct Madd29Mid {
    Fq29 U2, S2;
};

KZG_HD Madd29Mid g1_madd29_phase1(const G1Xyzz29 &p, const G1Affine29 &a, bool negate) {
    Madd29Mid m;
    Fq29 y2 = negate ? sub29<4>(zero29(), a.y) : a.y;
    m.U2 = mul29(a.x, p.zz);
    m.S2 = mul29(y2, p.zzz);
    return m;
}

template <class Reload>
KZG_HD G1Xyzz29 g1_madd29_phase2(const G1Xyzz29 &p, const Madd29Mid &m, bool negate, Reload reload) {
    Fq29 Pp = sub29<32>(m.U2, p.x);
    Fq29 R = sub29<16>(m.S2, p.y);
    Fq29 PP = sqr29(Pp);
    if (is_zero_mod_q_product(PP)) {
        // same x: either the same point (double it) or its inverse (infinity)
        if (!is_zero_mod_q_product(sqr29(R))) return G1Xyzz29::infinity();
        return g1_dbl29(g1_from_affine29(reload(), negate));
    }
    Fq29 PPP = mul29(Pp, PP);
    Fq29 Q = mul29(p.x, PP);
    G1Xyzz29 r;
    r.inf = 0;
    r.pad[0] = r.pad[1] = r.pad[2] = 0;
    r.zz = mul29(p.zz, PP);
    r.zzz = mul29(p.zzz, PPP);
    Fq29 t = add2x29(PPP, Q);
    r.x = sub29<16>(sqr29(R), t);
    // Y3 = R (Q - X3) + (16q - Y1) PPP: one double-width accumulation, one reduction (< 1.0001 q)
    r.y = muladd29_inline(R, sub29<32>(Q, r.x), sub29<16>(zero29(), p.y), PPP);
    return r;
}

KZG_HD G1Xyzz29 g1_madd29(const G1Xyzz29 &p, const G1Affine29 &a, bool negate) {
    if (a.is_inf()) return p;
    if (p.inf) return g1_from_affine29(a, negate);
    Madd29Mid m = g1_madd29_phase1(p, a, negate);
    return g1_madd29_phase2(p, m, negate, [&]() { return a; });
}

}  // namespace kzg
