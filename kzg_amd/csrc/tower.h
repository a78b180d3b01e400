// tower.h -- Fq2 / Fq6 / Fq12 tower of BLS12-381, G2 (the sextic twist y^2 = x^3 + 4(1+u) over Fq2) and the
// optimal ate pairing, for the verifier half of the crate (KZGVerifier::verify_eval*, src/coeff_form.rs:126-182;
// KZGVerifierEvalForm::verify_eval*, src/eval_form.rs:173-217).  Stands in for blstrs::{G2Affine, G2Projective,
// Gt, pairing} (external to the reference, Cargo.toml:23,27).
//
//   Fq2 = Fq[u]/(u^2 + 1),  Fq6 = Fq2[v]/(v^3 - xi),  Fq12 = Fq6[w]/(w^2 - v),  xi = 1 + u.
//
// A pairing check is a long serial chain (~25 k Fq multiplies); it is not a throughput kernel of the prover.
// One thread evaluates one check, so a batch of openings verifies in one launch; every operation is an
// out-of-line function on operands in memory (an Fq12 alone is 144 dwords -- it cannot live in VGPRs), which
// keeps the code a few tens of KB.  The verifier exposes booleans only: the Gt value is fixed up to the r-coprime
// power 3 of the final-exponentiation chain used below, which no comparison can observe.
#pragma once
#include "curve.h"

#if defined(__HIPCC__)
#define KZG_NI static __host__ __device__ __noinline__
#else
#define KZG_NI static __attribute__((noinline))
#endif

namespace kzg {

#include "tower_consts.inc"

// ------------------------------------------------------------------------------------------------ Fq2
struct Fq2 {
    Fq c0, c1;
    KZG_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    KZG_HD bool operator==(const Fq2 &o) const { return c0 == o.c0 && c1 == o.c1; }
    static KZG_HD Fq2 zero() { return Fq2{Fq::zero(), Fq::zero()}; }
    static KZG_HD Fq2 one() { return Fq2{Fq::one(), Fq::zero()}; }
};

KZG_HD void f2_add(Fq2 &r, const Fq2 &a, const Fq2 &b) {
    r.c0 = add(a.c0, b.c0);
    r.c1 = add(a.c1, b.c1);
}
KZG_HD void f2_sub(Fq2 &r, const Fq2 &a, const Fq2 &b) {
    r.c0 = sub(a.c0, b.c0);
    r.c1 = sub(a.c1, b.c1);
}
KZG_HD void f2_neg(Fq2 &r, const Fq2 &a) {
    r.c0 = a.c0.is_zero() ? a.c0 : neg(a.c0);
    r.c1 = a.c1.is_zero() ? a.c1 : neg(a.c1);
}
KZG_HD void f2_conj(Fq2 &r, const Fq2 &a) {
    r.c0 = a.c0;
    r.c1 = a.c1.is_zero() ? a.c1 : neg(a.c1);
}
KZG_HD void f2_mul(Fq2 &r, const Fq2 &a, const Fq2 &b) {  // Karatsuba, 3 Fq multiplies
    Fq t0 = mul(a.c0, b.c0), t1 = mul(a.c1, b.c1);
    Fq s = mul(add(a.c0, a.c1), add(b.c0, b.c1));
    r.c0 = sub(t0, t1);
    r.c1 = sub(sub(s, t0), t1);
}
KZG_HD void f2_sqr(Fq2 &r, const Fq2 &a) {  // (a0+a1)(a0-a1), 2 a0 a1
    Fq t = mul(a.c0, a.c1);
    r.c0 = mul(add(a.c0, a.c1), sub(a.c0, a.c1));
    r.c1 = dbl(t);
}
KZG_HD void f2_mul_fq(Fq2 &r, const Fq2 &a, const Fq &k) {
    r.c0 = mul(a.c0, k);
    r.c1 = mul(a.c1, k);
}
KZG_HD void f2_mul_xi(Fq2 &r, const Fq2 &a) {  // (a0 + a1 u)(1 + u)
    Fq t = sub(a.c0, a.c1);
    r.c1 = add(a.c0, a.c1);
    r.c0 = t;
}
KZG_HD void f2_inv(Fq2 &r, const Fq2 &a) {
    Fq n = inv(add(sqr(a.c0), sqr(a.c1)));
    Fq m = mul(a.c1, n);
    r.c0 = mul(a.c0, n);
    r.c1 = m.is_zero() ? m : neg(m);
}
KZG_HD void f2_dbl(Fq2 &r, const Fq2 &a) { f2_add(r, a, a); }

// ------------------------------------------------------------------------------------------------ Fq6
struct Fq6 {
    Fq2 c0, c1, c2;
};

KZG_HD void f6_add(Fq6 &r, const Fq6 &a, const Fq6 &b) {
    f2_add(r.c0, a.c0, b.c0);
    f2_add(r.c1, a.c1, b.c1);
    f2_add(r.c2, a.c2, b.c2);
}
KZG_HD void f6_sub(Fq6 &r, const Fq6 &a, const Fq6 &b) {
    f2_sub(r.c0, a.c0, b.c0);
    f2_sub(r.c1, a.c1, b.c1);
    f2_sub(r.c2, a.c2, b.c2);
}
KZG_HD void f6_neg(Fq6 &r, const Fq6 &a) {
    f2_neg(r.c0, a.c0);
    f2_neg(r.c1, a.c1);
    f2_neg(r.c2, a.c2);
}
KZG_HD void f6_mul_v(Fq6 &r, const Fq6 &a) {  // (c0, c1, c2) v = (xi c2, c0, c1)
    Fq2 t;
    f2_mul_xi(t, a.c2);
    r.c2 = a.c1;
    r.c1 = a.c0;
    r.c0 = t;
}
KZG_NI void f6_mul(Fq6 &r, const Fq6 &a, const Fq6 &b) {  // Karatsuba, 6 Fq2 multiplies
    Fq2 t0, t1, t2, s, x, y, c0, c1, c2;
    f2_mul(t0, a.c0, b.c0);
    f2_mul(t1, a.c1, b.c1);
    f2_mul(t2, a.c2, b.c2);
    f2_add(x, a.c1, a.c2);
    f2_add(y, b.c1, b.c2);
    f2_mul(s, x, y);
    f2_sub(s, s, t1);
    f2_sub(s, s, t2);
    f2_mul_xi(s, s);
    f2_add(c0, t0, s);
    f2_add(x, a.c0, a.c1);
    f2_add(y, b.c0, b.c1);
    f2_mul(s, x, y);
    f2_sub(s, s, t0);
    f2_sub(s, s, t1);
    f2_mul_xi(x, t2);
    f2_add(c1, s, x);
    f2_add(x, a.c0, a.c2);
    f2_add(y, b.c0, b.c2);
    f2_mul(s, x, y);
    f2_sub(s, s, t0);
    f2_sub(s, s, t2);
    f2_add(c2, s, t1);
    r.c0 = c0;
    r.c1 = c1;
    r.c2 = c2;
}
KZG_NI void f6_inv(Fq6 &r, const Fq6 &a) {
    Fq2 c0, c1, c2, t, s;
    f2_sqr(c0, a.c0);
    f2_mul(t, a.c1, a.c2);
    f2_mul_xi(t, t);
    f2_sub(c0, c0, t);  // a0^2 - xi a1 a2
    f2_sqr(c1, a.c2);
    f2_mul_xi(c1, c1);
    f2_mul(t, a.c0, a.c1);
    f2_sub(c1, c1, t);  // xi a2^2 - a0 a1
    f2_sqr(c2, a.c1);
    f2_mul(t, a.c0, a.c2);
    f2_sub(c2, c2, t);  // a1^2 - a0 a2
    f2_mul(t, a.c2, c1);
    f2_mul(s, a.c1, c2);
    f2_add(t, t, s);
    f2_mul_xi(t, t);
    f2_mul(s, a.c0, c0);
    f2_add(t, t, s);
    f2_inv(t, t);
    f2_mul(r.c0, c0, t);
    f2_mul(r.c1, c1, t);
    f2_mul(r.c2, c2, t);
}

// ------------------------------------------------------------------------------------------------ Fq12
struct Fq12 {
    Fq6 c0, c1;  // c0 + c1 w; coefficient of w^k: k even -> c0.c{k/2}, k odd -> c1.c{(k-1)/2}
};

KZG_NI void f12_one(Fq12 &r) {
    Fq2 z = Fq2::zero();
    r.c0.c0 = Fq2::one();
    r.c0.c1 = z;
    r.c0.c2 = z;
    r.c1.c0 = z;
    r.c1.c1 = z;
    r.c1.c2 = z;
}
KZG_NI bool f12_is_one(const Fq12 &a) {
    return a.c0.c0 == Fq2::one() && a.c0.c1.is_zero() && a.c0.c2.is_zero() && a.c1.c0.is_zero() && a.c1.c1.is_zero() &&
           a.c1.c2.is_zero();
}
KZG_NI void f12_mul(Fq12 &r, const Fq12 &a, const Fq12 &b) {  // 3 Fq6 multiplies
    Fq6 t0, t1, x, y, s;
    f6_mul(t0, a.c0, b.c0);
    f6_mul(t1, a.c1, b.c1);
    f6_add(x, a.c0, a.c1);
    f6_add(y, b.c0, b.c1);
    f6_mul(s, x, y);
    f6_sub(s, s, t0);
    f6_sub(r.c1, s, t1);
    f6_mul_v(t1, t1);
    f6_add(r.c0, t0, t1);
}
KZG_NI void f12_sqr(Fq12 &r, const Fq12 &a) {  // complex squaring, 2 Fq6 multiplies
    Fq6 t, x, y, s;
    f6_mul(t, a.c0, a.c1);
    f6_add(x, a.c0, a.c1);
    f6_mul_v(y, a.c1);
    f6_add(y, y, a.c0);
    f6_mul(s, x, y);
    f6_sub(s, s, t);
    f6_mul_v(x, t);
    f6_sub(r.c0, s, x);
    f6_add(r.c1, t, t);
}
// (x + y s)^2 in Fq4 = Fq2[s]/(s^2 - xi): t0 = x^2 + xi y^2, t1 = 2 x y  (2 Fq2 products)
KZG_HD void f4_sqr(Fq2 &t0, Fq2 &t1, const Fq2 &x, const Fq2 &y) {
    Fq2 t, u, v;
    f2_mul(t, x, y);
    f2_add(u, x, y);
    f2_mul_xi(v, y);
    f2_add(v, v, x);
    f2_mul(u, u, v);
    f2_sub(u, u, t);
    f2_mul_xi(v, t);
    f2_sub(t0, u, v);
    f2_dbl(t1, t);
}
// a^2 for a in the cyclotomic subgroup (Granger-Scott): three Fq4 squarings, 18 Fq multiplies instead of 36
KZG_NI void f12_cyclotomic_sqr(Fq12 &r, const Fq12 &a) {
    Fq2 t0, t1, t2, t3, t4, t5, t;
    f4_sqr(t0, t1, a.c0.c0, a.c1.c1);
    f4_sqr(t2, t3, a.c1.c0, a.c0.c2);
    f4_sqr(t4, t5, a.c0.c1, a.c1.c2);
    Fq12 o;
    f2_sub(t, t0, a.c0.c0);
    f2_dbl(t, t);
    f2_add(o.c0.c0, t, t0);  // 3 t0 - 2 z0
    f2_add(t, t1, a.c1.c1);
    f2_dbl(t, t);
    f2_add(o.c1.c1, t, t1);  // 3 t1 + 2 z1
    f2_mul_xi(t5, t5);
    f2_add(t, t5, a.c1.c0);
    f2_dbl(t, t);
    f2_add(o.c1.c0, t, t5);  // 3 xi t5 + 2 z2
    f2_sub(t, t4, a.c0.c2);
    f2_dbl(t, t);
    f2_add(o.c0.c2, t, t4);  // 3 t4 - 2 z3
    f2_sub(t, t2, a.c0.c1);
    f2_dbl(t, t);
    f2_add(o.c0.c1, t, t2);  // 3 t2 - 2 z4
    f2_add(t, t3, a.c1.c2);
    f2_dbl(t, t);
    f2_add(o.c1.c2, t, t3);  // 3 t3 + 2 z5
    r = o;
}
KZG_NI void f12_conj(Fq12 &r, const Fq12 &a) {  // a^(q^6)
    r.c0 = a.c0;
    f6_neg(r.c1, a.c1);
}
KZG_NI void f12_inv(Fq12 &r, const Fq12 &a) {
    Fq6 t0, t1;
    f6_mul(t0, a.c0, a.c0);
    f6_mul(t1, a.c1, a.c1);
    f6_mul_v(t1, t1);
    f6_sub(t0, t0, t1);
    f6_inv(t0, t0);
    f6_mul(r.c0, a.c0, t0);
    f6_mul(t1, a.c1, t0);
    f6_neg(r.c1, t1);
}
KZG_HD Fq2 *f12_coeff(Fq12 &a, int k) {
    Fq6 &h = (k & 1) ? a.c1 : a.c0;
    return (k >> 1) == 0 ? &h.c0 : ((k >> 1) == 1 ? &h.c1 : &h.c2);
}
KZG_NI void f12_frob(Fq12 &r, const Fq12 &a) {  // sum conj(b_k) xi^(k(q-1)/6) w^k
    Fq12 t = a;
    for (int k = 0; k < 6; k++) {
        Fq2 *c = f12_coeff(t, k);
        Fq2 g{tower_frob1_limb(2 * k), tower_frob1_limb(2 * k + 1)}, x;
        f2_conj(x, *c);
        f2_mul(*c, x, g);
    }
    r = t;
}
KZG_NI void f12_frob2(Fq12 &r, const Fq12 &a) {  // sum b_k xi^(k(q^2-1)/6) w^k, coefficients in Fq
    Fq12 t = a;
    for (int k = 0; k < 6; k++) {
        Fq2 *c = f12_coeff(t, k);
        f2_mul_fq(*c, *c, tower_frob2_limb(k));
    }
    r = t;
}

// ------------------------------------------------------------------------------------------------ G2
struct G2Affine {  // identity: x = y = 0 (blst_p2_affine convention)
    Fq2 x, y;
    KZG_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
};
struct G2Jacobian {  // blst_p2: identity Z = 0
    Fq2 x, y, z;
    KZG_HD bool is_inf() const { return z.is_zero(); }
};

KZG_HD G2Affine g2_generator() {
    G2Affine g;
    g.x.c0 = g2_generator_limb(0);
    g.x.c1 = g2_generator_limb(1);
    g.y.c0 = g2_generator_limb(2);
    g.y.c1 = g2_generator_limb(3);
    return g;
}
KZG_HD void g2_set_inf(G2Jacobian &p) {
    p.x = Fq2::zero();
    p.y = Fq2::zero();
    p.z = Fq2::zero();
}
KZG_NI void g2_from_affine(G2Jacobian &r, const G2Affine &a) {
    if (a.is_inf()) {
        g2_set_inf(r);
        return;
    }
    r.x = a.x;
    r.y = a.y;
    r.z = Fq2::one();
}
KZG_NI bool g2_on_curve(const G2Affine &a) {
    if (a.is_inf()) return true;
    Fq2 l, rr, b;
    f2_sqr(l, a.y);
    f2_sqr(rr, a.x);
    f2_mul(rr, rr, a.x);
    b.c0 = from_u64<FqParams>(4);
    b.c1 = b.c0;
    f2_add(rr, rr, b);
    return l == rr;
}
// dbl-2009-l (a = 0)
KZG_NI void g2_dbl(G2Jacobian &r, const G2Jacobian &p) {
    if (p.is_inf() || p.y.is_zero()) {
        g2_set_inf(r);
        return;
    }
    Fq2 A, B, C, D, E, F, t, z3;
    f2_sqr(A, p.x);
    f2_sqr(B, p.y);
    f2_sqr(C, B);
    f2_add(t, p.x, B);
    f2_sqr(t, t);
    f2_sub(t, t, A);
    f2_sub(t, t, C);
    f2_dbl(D, t);
    f2_dbl(E, A);
    f2_add(E, E, A);
    f2_sqr(F, E);
    f2_mul(z3, p.y, p.z);
    f2_dbl(z3, z3);
    f2_dbl(t, D);
    f2_sub(r.x, F, t);
    f2_sub(t, D, r.x);
    f2_mul(t, E, t);
    f2_dbl(C, C);
    f2_dbl(C, C);
    f2_dbl(C, C);
    f2_sub(r.y, t, C);
    r.z = z3;
}
// add-2007-bl with the doubling / inverse cases
KZG_NI void g2_add(G2Jacobian &r, const G2Jacobian &p, const G2Jacobian &q) {
    if (p.is_inf()) {
        r = q;
        return;
    }
    if (q.is_inf()) {
        r = p;
        return;
    }
    Fq2 z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t;
    f2_sqr(z1z1, p.z);
    f2_sqr(z2z2, q.z);
    f2_mul(u1, p.x, z2z2);
    f2_mul(u2, q.x, z1z1);
    f2_mul(s1, p.y, q.z);
    f2_mul(s1, s1, z2z2);
    f2_mul(s2, q.y, p.z);
    f2_mul(s2, s2, z1z1);
    f2_sub(h, u2, u1);
    f2_sub(rr, s2, s1);
    if (h.is_zero()) {
        if (rr.is_zero()) {
            g2_dbl(r, p);
        } else {
            g2_set_inf(r);
        }
        return;
    }
    f2_dbl(rr, rr);
    f2_dbl(i, h);
    f2_sqr(i, i);
    f2_mul(j, h, i);
    f2_mul(v, u1, i);
    G2Jacobian o;
    f2_sqr(o.x, rr);
    f2_sub(o.x, o.x, j);
    f2_sub(o.x, o.x, v);
    f2_sub(o.x, o.x, v);
    f2_sub(t, v, o.x);
    f2_mul(t, rr, t);
    f2_mul(s1, s1, j);
    f2_dbl(s1, s1);
    f2_sub(o.y, t, s1);
    f2_add(t, p.z, q.z);
    f2_sqr(t, t);
    f2_sub(t, t, z1z1);
    f2_sub(t, t, z2z2);
    f2_mul(o.z, t, h);
    r = o;
}
KZG_NI void g2_neg_affine(G2Affine &r, const G2Affine &a) {
    r.x = a.x;
    f2_neg(r.y, a.y);
}
KZG_NI void g2_to_affine(G2Affine &r, const G2Jacobian &p) {
    if (p.is_inf()) {
        r.x = Fq2::zero();
        r.y = Fq2::zero();
        return;
    }
    Fq2 zi, zi2;
    f2_inv(zi, p.z);
    f2_sqr(zi2, zi);
    f2_mul(r.x, p.x, zi2);
    f2_mul(zi2, zi2, zi);
    f2_mul(r.y, p.y, zi2);
}
// [k]P, k canonical 8 x u32 LE; double-and-add (vartime)
KZG_NI void g2_scalar_mul(G2Jacobian &r, const G2Affine &p, const uint32_t k[8]) {
    G2Jacobian acc, pj;
    g2_set_inf(acc);
    g2_from_affine(pj, p);
    for (int i = 255; i >= 0; i--) {
        g2_dbl(acc, acc);
        if ((k[i >> 5] >> (i & 31)) & 1) g2_add(acc, acc, pj);
    }
    r = acc;
}

// ------------------------------------------------------------------------------------------------ pairing
constexpr uint64_t BLS_Z_ABS = 0xd201000000010000ull;  // the curve parameter is -BLS_Z_ABS

constexpr int MILLER_LINES = 68;  // 63 doublings + 5 additions (hamming weight of |z| minus one)

// One Miller-loop step for T (doubling when S == nullptr, else the chord through T and *S): the line's slope and
// constant term  (lam, lam xT - yT)  and the update of T.
KZG_NI void miller_step(G2Affine &T, const G2Affine *S, Fq2 &lam, Fq2 &cst) {
    Fq2 t, x3;
    if (!S) {
        f2_sqr(lam, T.x);
        f2_dbl(t, lam);
        f2_add(lam, lam, t);
        f2_dbl(t, T.y);
    } else {
        f2_sub(lam, S->y, T.y);
        f2_sub(t, S->x, T.x);
    }
    f2_inv(t, t);
    f2_mul(lam, lam, t);
    f2_mul(cst, lam, T.x);
    f2_sub(cst, cst, T.y);
    f2_sqr(x3, lam);
    f2_sub(x3, x3, T.x);
    f2_sub(x3, x3, S ? S->x : T.x);
    f2_sub(t, T.x, x3);
    f2_mul(t, lam, t);
    f2_sub(T.y, t, T.y);
    T.x = x3;
}

// f *= the line (lam, cst) evaluated at P, scaled by w^3 (killed by the final exponentiation):
//   yP w^3 - lam xP w^2 + cst   ->  c0 = (cst, -lam xP, 0), c1 = (0, yP, 0)
KZG_NI void f12_mul_line(Fq12 &f, const Fq2 &lam, const Fq2 &cst, const G1Affine &P) {
    Fq12 l;
    Fq2 z = Fq2::zero(), t;
    l.c0.c0 = cst;
    f2_mul_fq(t, lam, P.x);
    f2_neg(l.c0.c1, t);
    l.c0.c2 = z;
    l.c1.c0 = z;
    l.c1.c1.c0 = P.y;
    l.c1.c1.c1 = Fq::zero();
    l.c1.c2 = z;
    f12_mul(f, f, l);
}

// The MILLER_LINES (lam, cst) pairs of a FIXED Q, in loop order: tab[2k], tab[2k+1].  The verifier's second argument
// is almost always one of two SRS points (hs[0], hs[1]); with their lines stored once a check needs no G2 arithmetic.
KZG_NI void g2_precompute_lines(const G2Affine &Q, Fq2 *tab) {
    G2Affine T = Q;
    int k = 0;
    for (int b = 62; b >= 0; b--) {
        if (Q.is_inf()) break;
        miller_step(T, nullptr, tab[2 * k], tab[2 * k + 1]);
        k++;
        if ((BLS_Z_ABS >> b) & 1) {
            miller_step(T, &Q, tab[2 * k], tab[2 * k + 1]);
            k++;
        }
    }
    for (; k < MILLER_LINES; k++) {
        tab[2 * k] = Fq2::zero();
        tab[2 * k + 1] = Fq2::zero();
    }
}

// prod_i f_{z,Q_i}(P_i).  Pair i uses the stored lines tabs[i] when that pointer is non-null, otherwise T_i is kept
// affine and updated on the fly (one Fq inversion per step: cheap next to the Fq12 work and free of projective line
// formulas).  Pairs with an identity member contribute 1.
KZG_NI void miller_loop(Fq12 &f, const G1Affine *Ps, const G2Affine *Qs, G2Affine *Ts, int np, const Fq2 *const *tabs = nullptr) {
    f12_one(f);
    for (int i = 0; i < np; i++) Ts[i] = Qs[i];
    int k = 0;
    for (int b = 62; b >= 0; b--) {
        const bool add = ((BLS_Z_ABS >> b) & 1) != 0;
        f12_sqr(f, f);
        for (int i = 0; i < np; i++) {
            if (Ps[i].is_inf() || Qs[i].is_inf()) continue;
            const Fq2 *tab = tabs ? tabs[i] : nullptr;
            Fq2 lam, cst;
            if (tab) {
                f12_mul_line(f, tab[2 * k], tab[2 * k + 1], Ps[i]);
                if (add) f12_mul_line(f, tab[2 * k + 2], tab[2 * k + 3], Ps[i]);
            } else {
                miller_step(Ts[i], nullptr, lam, cst);
                f12_mul_line(f, lam, cst, Ps[i]);
                if (add) {
                    miller_step(Ts[i], &Qs[i], lam, cst);
                    f12_mul_line(f, lam, cst, Ps[i]);
                }
            }
        }
        k += add ? 2 : 1;
    }
    f12_conj(f, f);  // z < 0
}

// a^z for a in the cyclotomic subgroup (inverse = conjugate)
KZG_NI void f12_exp_z(Fq12 &r, const Fq12 &a) {
    Fq12 acc = a;
    for (int b = 62; b >= 0; b--) {
        f12_cyclotomic_sqr(acc, acc);
        if ((BLS_Z_ABS >> b) & 1) f12_mul(acc, acc, a);
    }
    f12_conj(r, acc);
}

// f^(3 (q^12 - 1)/r):  easy part (q^6 - 1)(q^2 + 1), hard part 3(q^4 - q^2 + 1)/r = (z-1)^2 (z+q)(z^2+q^2-1) + 3
KZG_NI void final_exponentiation(Fq12 &r, const Fq12 &f) {
    Fq12 f1, f2, a, b, c, t;
    f12_inv(t, f);
    f12_conj(f1, f);
    f12_mul(f1, f1, t);
    f12_frob2(f2, f1);
    f12_mul(f2, f2, f1);
    f12_exp_z(a, f2);
    f12_conj(t, f2);
    f12_mul(a, a, t);
    f12_exp_z(b, a);
    f12_conj(t, a);
    f12_mul(a, b, t);
    f12_exp_z(b, a);
    f12_frob(t, a);
    f12_mul(b, b, t);
    f12_exp_z(c, b);
    f12_exp_z(c, c);
    f12_frob2(t, b);
    f12_mul(c, c, t);
    f12_conj(t, b);
    f12_mul(c, c, t);
    f12_sqr(t, f2);
    f12_mul(t, t, f2);
    f12_mul(r, c, t);
}

// prod_i e(P_i, Q_i) == 1
KZG_NI bool pairing_product_is_one(const G1Affine *Ps, const G2Affine *Qs, G2Affine *Ts, int np,
                                   const Fq2 *const *tabs = nullptr) {
    Fq12 f, g;
    miller_loop(f, Ps, Qs, Ts, np, tabs);
    final_exponentiation(g, f);
    return f12_is_one(g);
}

}  // namespace kzg
