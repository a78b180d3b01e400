// curve30.h -- G1 arithmetic in extended Jacobian "XYZZ" coordinates on the signed 13 x 30-bit Fq representation
// (field30.h): the inner loop of the MSM accumulation kernel and every kernel of its tail.  Same formulas as
// curve.h (madd-2008-s 8M + 2S, add-2008-s 12M + 2S, dbl-2008-s-1 6M + 3S); what changes is the bookkeeping:
// values are signed and lazily reduced, a subtraction is limb-wise followed by a carry pass, and there is no
// conditional subtraction anywhere.
//
// Bounds, in units of q (M = "a Montgomery product": |M| <= 0.5 + 0.0016 |a||b|, field30.h):
//   table point x2, y2: |.| < 0.51                   accumulator ZZ, ZZZ, Y: products, < 0.6
//   mixed add:  P = U2 - X1, R = S2 - Y1;  X3 = R^2 - PPP - 2Q: |X3| < 2.2;  |P|, |Q - X3| < 2.8;  Y3: one product
//   general add / double: inputs |X|, |Y| < 8 on both sides, outputs |X| < 2.2 (add) / 1.6 (double), |Y| < 1.2
// so every mix of the three operations stays inside the classes, every product has |a||b| < 70 (limit 300), and
// "is zero mod q" of a product is "all limbs zero".
#pragma once
#include "curve.h"
#include "field30.h"

namespace kzg {

// KZG_ROW_BYTES: size of a resident table entry.  128 = one cache line per gather.  112 (the two coordinates + 8 B, 7 x dwordx4)
// saves 12.5 % of the table's HBM but an entry then straddles two 128-B lines three times out of four: measured on the same box
// (profiles/traffic.json) 2.00 against 1.14 GB of FETCH_SIZE per launch, k_accum_affine 2.33 against 2.20 ms, batched
// throughput 406 against 420.5 commitments/s.
#ifndef KZG_ROW_BYTES
#define KZG_ROW_BYTES 128
#endif
struct alignas(16) G1Affine30 {  // table entry: x, y normalised; identity = all limbs zero
    Fq30 x, y;
    uint32_t pad[(KZG_ROW_BYTES - 104) / 4];
    KZG_HD bool is_inf() const { return x.limbs_all_zero() && y.limbs_all_zero(); }
    // the same test for a VALIDATED table entry, on y alone: the curve has odd order (no point of order two, so y != 0 on every
    // finite point), and a table coordinate is below q in magnitude, so y = 0 mod q means all limbs zero
    KZG_HD bool is_inf_table() const { return y.limbs_all_zero(); }
};

struct alignas(16) G1Xyzz30 {  // 224 B: what the MSM partial-sum buffers hold
    Fq30 x, y, zz, zzz;
    uint32_t inf;
    uint32_t pad[3];
    static KZG_HD G1Xyzz30 infinity() {
        G1Xyzz30 p;
#pragma unroll
        for (int i = 0; i < F30_N; i++) p.x.v[i] = p.y.v[i] = p.zz.v[i] = p.zzz.v[i] = 0;
        p.inf = 1;
        p.pad[0] = p.pad[1] = p.pad[2] = 0;
        return p;
    }
};

KZG_HD G1Affine30 g1_affine_to30(const G1Affine &a) {
    G1Affine30 r;
#pragma unroll
    for (int i = 0; i < (KZG_ROW_BYTES - 104) / 4; i++) r.pad[i] = 0;
    if (a.is_inf()) {
        r.x = zero30();
        r.y = zero30();
        return r;
    }
    r.x = to30(a.x);
    r.y = to30(a.y);
    return r;
}

KZG_HD G1Xyzz30 g1_from_affine30(const G1Affine30 &a, bool negate) {
    G1Xyzz30 p;
    p.inf = a.is_inf() ? 1u : 0u;
    p.pad[0] = p.pad[1] = p.pad[2] = 0;
    p.x = a.x;
    p.y = cneg30(a.y, negate);
    p.zz = one30();
    p.zzz = one30();
    return p;
}

KZG_HD G1Xyzz g1_xyzz_from30(const G1Xyzz30 &p) {
    if (p.inf) return G1Xyzz::inf();
    G1Xyzz r;
    r.x = from30(p.x);
    r.y = from30(p.y);
    r.zz = from30(p.zz);
    r.zzz = from30(p.zzz);
    if (r.zz.is_zero()) return G1Xyzz::inf();
    return r;
}

KZG_HD G1Xyzz30 g1_xyzz_to30(const G1Xyzz &p) {
    G1Xyzz30 r;
    r.inf = p.is_inf() ? 1u : 0u;
    r.pad[0] = r.pad[1] = r.pad[2] = 0;
    r.x = to30(p.x);
    r.y = to30(p.y);
    r.zz = to30(p.zz);
    r.zzz = to30(p.zzz);
    return r;
}

// a * 2 and a * 3 (limb-wise, then normalised; a normalised)
KZG_HD Fq30 times2_30(const Fq30 &a) {
    Fq30 r;
#pragma unroll
    for (int i = 0; i < F30_N; i++) r.v[i] = a.v[i] * 2;
    return normalize30(r);
}
KZG_HD Fq30 times3_30(const Fq30 &a) {
    Fq30 r;
#pragma unroll
    for (int i = 0; i < F30_N; i++) r.v[i] = a.v[i] * 3;
    return normalize30(r);
}

// dbl-2008-s-1 (a = 0): 6M + 3S.  In: |X|, |Y| < 8.  Out: |X| < 1.6, |Y| < 1.2.
KZG_HD G1Xyzz30 g1_dbl30(const G1Xyzz30 &p) {
    if (p.inf) return p;
    Fq30 U = times2_30(p.y);                // < 16 (limbs of y <= 2^29 in magnitude, normalised or negated)
    Fq30 V = sqr30(U);                      // 16^2 = 256 < 300: |V| < q, so the zero test below is exact
    if (is_zero30(V)) return G1Xyzz30::infinity();  // y == 0: a point of order two
    Fq30 W = mul30(U, V);
    Fq30 S = mul30(p.x, V);
    Fq30 Mm = times3_30(sqr30(p.x));        // < 1.9
    G1Xyzz30 r;
    r.inf = 0;
    r.pad[0] = r.pad[1] = r.pad[2] = 0;
    r.x = sub30(sqr30(Mm), times2_30(S));                        // < 1.6
    r.y = sub30(mul30(Mm, sub30(S, r.x)), mul30(W, p.y));        // < 1.2
    r.zz = mul30(V, p.zz);
    r.zzz = mul30(W, p.zzz);
    return r;
}

// add-2008-s: 12M + 2S.  In: |X|, |Y| < 8 on both sides.  Out: |X| < 2.2, |Y| < 1.1.
KZG_HD G1Xyzz30 g1_add30(const G1Xyzz30 &p, const G1Xyzz30 &q) {
    if (q.inf) return p;
    if (p.inf) return q;
    Fq30 U1 = mul30(p.x, q.zz);
    Fq30 U2 = mul30(q.x, p.zz);
    Fq30 S1 = mul30(p.y, q.zzz);
    Fq30 S2 = mul30(q.y, p.zzz);
    Fq30 Pp = sub30(U2, U1);              // < 1.1
    Fq30 R = sub30(S2, S1);
    Fq30 PP = sqr30(Pp);
    if (is_zero30(PP)) {
        if (is_zero30(sqr30(R))) return g1_dbl30(p);
        return G1Xyzz30::infinity();
    }
    Fq30 PPP = mul30(Pp, PP);
    Fq30 Q = mul30(U1, PP);
    G1Xyzz30 r;
    r.inf = 0;
    r.pad[0] = r.pad[1] = r.pad[2] = 0;
    r.zz = mul30(mul30(p.zz, q.zz), PP);
    r.zzz = mul30(mul30(p.zzz, q.zzz), PPP);
    Fq30 Bm = mul30(S1, PPP);
    r.x = sub30(sqr30(R), add2x30(PPP, Q));                      // < 2.2
    r.y = sub30(mul30(R, sub30(Q, r.x)), Bm);                    // < 1.1
    return r;
}

// acc += (negate ? -a : a), split in two so the caller can re-use the registers of `a` for the next
// gather as soon as the two products that read it are done:
//   phase 1: P = x2 * ZZ1 - X1, R = (+-y2) * ZZZ1 - Y1     (the only uses of the affine point; the subtractions are merged
//            into the products' output columns, field30.h mul30_sub: both come out normalised)
//   phase 2: everything else; `reload` re-fetches the affine point in the rare doubling case.
struct Madd30Mid {
    Fq30 P, R;
};

KZG_HD Madd30Mid g1_madd30_phase1(const G1Xyzz30 &p, const G1Affine30 &a, bool negate) {
    Madd30Mid m;
    m.P = mul30_sub(a.x, p.zz, p.x);
    m.R = mul30_sub(cneg30(a.y, negate), p.zzz, p.y);
    return m;
}

template <class Reload>
KZG_HD G1Xyzz30 g1_madd30_phase2(const G1Xyzz30 &p, const Madd30Mid &m, bool negate, Reload reload) {
    const Fq30 &Pp = m.P, &R = m.R;
    Fq30 PP = sqr30(Pp);
    if (is_zero30(PP)) {
        // same x: either the same point (double it) or its inverse (infinity)
        if (!is_zero30(sqr30(R))) return G1Xyzz30::infinity();
        return g1_dbl30(g1_from_affine30(reload(), negate));
    }
    // X3, ZZ3, ZZZ3 and Q come out with UNSIGNED digits (field30.h mul30u: no rounding add per digit).  Each of them only ever
    // meets a balanced partner again -- x2, y2, PP, PPP -- or a subtraction; P, R, PP, PPP, Y3 stay balanced.  A point that
    // leaves the accumulation loop is normalised first (g1_normalize30).
    Fq30 PPP = mul30(Pp, PP);
    Fq30 Q = mul30u(p.x, PP);
    G1Xyzz30 r;
    r.inf = 0;
    r.pad[0] = r.pad[1] = r.pad[2] = 0;
    r.zz = mul30u(p.zz, PP);
    r.zzz = mul30u(p.zzz, PPP);
    r.x = sqr30_sub2u(R, PPP, Q);  // X3 = R^2 - PPP - 2Q, one digit extraction
    // Y3 = R (Q - X3) + (-Y1) PPP: one double-width accumulation, one reduction
    r.y = muladd30(R, sub30(Q, r.x), neg30(p.y), PPP);
    return r;
}

// the accumulator of a madd chain back in the form every other operation expects (balanced digits)
KZG_HD G1Xyzz30 g1_normalize30(G1Xyzz30 p) {
    p.x = normalize30(p.x);
    p.zz = normalize30(p.zz);
    p.zzz = normalize30(p.zzz);
    return p;
}

KZG_HD G1Xyzz30 g1_madd30(const G1Xyzz30 &p, const G1Affine30 &a, bool negate) {
    if (a.is_inf()) return p;
    if (p.inf) return g1_from_affine30(a, negate);
    Madd30Mid m = g1_madd30_phase1(p, a, negate);
    return g1_normalize30(g1_madd30_phase2(p, m, negate, [&]() { return a; }));
}

}  // namespace kzg
