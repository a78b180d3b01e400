// Host build (g++) of kzg_amd/csrc/tower.h -- the same source hipcc compiles for gfx950 -- so tests/ can check
// the tower fields, G2 and the pairing against the python oracle without a GPU.  Test infrastructure only.
#include "../kzg_amd/csrc/tower.h"
using namespace kzg;

static Fq load_fq(const uint8_t *p) {  // canonical little-endian 48 B -> Montgomery
    Fq a;
    memcpy(a.v, p, 48);
    return to_mont(a);
}
static void store_fq(uint8_t *p, const Fq &a) {
    Fq c = from_mont(a);
    memcpy(p, c.v, 48);
}
static Fq2 load_f2(const uint8_t *p) { return Fq2{load_fq(p), load_fq(p + 48)}; }
static void store_f2(uint8_t *p, const Fq2 &a) {
    store_fq(p, a.c0);
    store_fq(p + 48, a.c1);
}
static void load_f12(Fq12 &a, const uint8_t *p) {  // 6 Fq2 coefficients of w^0..w^5
    for (int k = 0; k < 6; k++) *f12_coeff(a, k) = load_f2(p + 96 * k);
}
static void store_f12(uint8_t *p, Fq12 &a) {
    for (int k = 0; k < 6; k++) store_f2(p + 96 * k, *f12_coeff(a, k));
}
static G1Affine load_g1(const uint8_t *p) { return G1Affine{load_fq(p), load_fq(p + 48)}; }
static G2Affine load_g2(const uint8_t *p) { return G2Affine{load_f2(p), load_f2(p + 96)}; }

extern "C" {
void ht_f2_mul(const uint8_t *a, const uint8_t *b, uint8_t *o) {
    Fq2 r;
    f2_mul(r, load_f2(a), load_f2(b));
    store_f2(o, r);
}
void ht_f2_sqr(const uint8_t *a, uint8_t *o) {
    Fq2 r = load_f2(a);
    f2_sqr(r, r);
    store_f2(o, r);
}
void ht_f2_inv(const uint8_t *a, uint8_t *o) {
    Fq2 r = load_f2(a);
    f2_inv(r, r);
    store_f2(o, r);
}
void ht_f12_mul(const uint8_t *a, const uint8_t *b, uint8_t *o) {
    Fq12 x, y;
    load_f12(x, a);
    load_f12(y, b);
    f12_mul(x, x, y);
    store_f12(o, x);
}
void ht_f12_sqr(const uint8_t *a, uint8_t *o) {
    Fq12 x;
    load_f12(x, a);
    f12_sqr(x, x);
    store_f12(o, x);
}
void ht_f12_cyclotomic_sqr(const uint8_t *a, uint8_t *o) {
    Fq12 x;
    load_f12(x, a);
    f12_cyclotomic_sqr(x, x);
    store_f12(o, x);
}
void ht_f12_inv(const uint8_t *a, uint8_t *o) {
    Fq12 x;
    load_f12(x, a);
    f12_inv(x, x);
    store_f12(o, x);
}
void ht_f12_frob(const uint8_t *a, int power, uint8_t *o) {
    Fq12 x;
    load_f12(x, a);
    if (power == 1) f12_frob(x, x); else f12_frob2(x, x);
    store_f12(o, x);
}
void ht_g2_generator(uint8_t *o) {
    G2Affine g = g2_generator();
    store_f2(o, g.x);
    store_f2(o + 96, g.y);
}
int ht_g2_on_curve(const uint8_t *p) { return g2_on_curve(load_g2(p)) ? 1 : 0; }
void ht_g2_mul(const uint8_t *p, const uint8_t *k, uint8_t *o) {  // affine canonical 192 B, k canonical LE 32 B
    uint32_t kk[8];
    memcpy(kk, k, 32);
    G2Jacobian r;
    g2_scalar_mul(r, load_g2(p), kk);
    G2Affine a;
    g2_to_affine(a, r);
    store_f2(o, a.x);
    store_f2(o + 96, a.y);
}
void ht_g2_add(const uint8_t *p, const uint8_t *q, uint8_t *o) {
    G2Jacobian a, b;
    g2_from_affine(a, load_g2(p));
    g2_from_affine(b, load_g2(q));
    g2_add(a, a, b);
    G2Affine r;
    g2_to_affine(r, a);
    store_f2(o, r.x);
    store_f2(o + 96, r.y);
}
void ht_miller_loop(const uint8_t *ps, const uint8_t *qs, int np, uint8_t *o) {
    G1Affine P[4];
    G2Affine Q[4], T[4];
    for (int i = 0; i < np; i++) {
        P[i] = load_g1(ps + 96 * i);
        Q[i] = load_g2(qs + 192 * i);
    }
    Fq12 f;
    miller_loop(f, P, Q, T, np);
    store_f12(o, f);
}
// same product with the lines of every Q precomputed (the verifier's fixed-argument path)
void ht_miller_loop_fixed(const uint8_t *ps, const uint8_t *qs, int np, uint8_t *o) {
    G1Affine P[4];
    G2Affine Q[4], T[4];
    static Fq2 tabs[4][2 * MILLER_LINES];
    const Fq2 *tp[4];
    for (int i = 0; i < np; i++) {
        P[i] = load_g1(ps + 96 * i);
        Q[i] = load_g2(qs + 192 * i);
        g2_precompute_lines(Q[i], tabs[i]);
        tp[i] = (i & 1) ? nullptr : tabs[i];  // mix stored and on-the-fly pairs
    }
    Fq12 f;
    miller_loop(f, P, Q, T, np, tp);
    store_f12(o, f);
}
void ht_final_exp(const uint8_t *a, uint8_t *o) {
    Fq12 x, y;
    load_f12(x, a);
    final_exponentiation(y, x);
    store_f12(o, y);
}
int ht_pairing_product_is_one(const uint8_t *ps, const uint8_t *qs, int np) {
    G1Affine P[4];
    G2Affine Q[4], T[4];
    for (int i = 0; i < np; i++) {
        P[i] = load_g1(ps + 96 * i);
        Q[i] = load_g2(qs + 192 * i);
    }
    return pairing_product_is_one(P, Q, T, np) ? 1 : 0;
}
}
