// curve29.h -- XYZZ mixed addition on the unsaturated 29-bit Fq representation (field29.h), the inner
// loop of the MSM accumulation kernel.  Same formulas as g1_madd (madd-2008-s, 8M + 2S); what changes is
// the bookkeeping: values are lazily reduced, every subtraction adds a multiple of q chosen from the
// proven bound of its subtrahend, and only the operands of multiplications are limb-normalised.
//
// Bounds (M = "Montgomery product", < 1.0001 q because R29 = 2^406 >> q):
//   table point x2, y2 < 2q;  -y2 = 4q - y2 < 4q
//   acc.zz, acc.zzz : M            acc.x < 19q            acc.y < 11q
//   P  = U2 - X1 + 32q < 35q       R  = S2 - Y1 + 16q < 19q
//   X3 = R^2 - (PPP + 2Q) + 16q < 19q      Q - X3 + 32q < 35q      Y3 = R(Q - X3) - Y1*PPP + 8q < 11q
// All far below the 2^12 q limit of mul29.
#pragma once
#include "curve.h"
#include "field29.h"

namespace kzg {

struct G1Affine29 {  // table entry: x, y < 2q normalised; identity = all limbs zero
    Fq29 x, y;
    KZG_HD bool is_inf() const { return x.limbs_all_zero() && y.limbs_all_zero(); }
};

struct G1Xyzz29 {
    Fq29 x, y, zz, zzz;
    bool inf;
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
    p.inf = a.is_inf();
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
    r.inf = p.is_inf();
    r.x = to29(p.x);
    r.y = to29(p.y);
    r.zz = to29(p.zz);
    r.zzz = to29(p.zzz);
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
        // same x: either the same point (double it) or its inverse (infinity).  Rare: done in the
        // saturated representation.
        Fq29 RR = sqr29(R);
        if (!is_zero_mod_q_product(RR)) {
            G1Xyzz29 r = p;
            r.inf = true;
            return r;
        }
        G1Affine29 a = reload();
        G1Affine s;
        s.x = from29(a.x);
        s.y = from29(negate ? sub29<4>(zero29(), a.y) : a.y);
        return g1_xyzz_to29(g1_dbl_affine(s));
    }
    Fq29 PPP = mul29(Pp, PP);
    Fq29 Q = mul29(p.x, PP);
    G1Xyzz29 r;
    r.inf = false;
    r.zz = mul29(p.zz, PP);
    r.zzz = mul29(p.zzz, PPP);
    Fq29 Bm = mul29(p.y, PPP);
    Fq29 t = add2x29(PPP, Q);
    r.x = sub29<16>(sqr29(R), t);
    r.y = sub29<8>(mul29(R, sub29<32>(Q, r.x)), Bm);
    return r;
}

KZG_HD G1Xyzz29 g1_madd29(const G1Xyzz29 &p, const G1Affine29 &a, bool negate) {
    if (a.is_inf()) return p;
    if (p.inf) return g1_from_affine29(a, negate);
    Madd29Mid m = g1_madd29_phase1(p, a, negate);
    return g1_madd29_phase2(p, m, negate, [&]() { return a; });
}

}  // namespace kzg
