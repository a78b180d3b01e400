/*
 * ORACLE (test infrastructure, NOT product code) -- plain-C CPU restatement of the proxima-one/kzg
 * commit/open hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (libkzg_mi355x.so) never links or calls it.
 *
 * PARITY STATUS: see oracle/kzg_model.py header.  Fr-polynomial layer pinned by the reference's
 * literal tests (src/polynomial.rs:494-690); G1 layer "parity unpinned" by the reference (its
 * arithmetic is the un-vendored blstrs rev b98fc83 / blst; Cargo.toml:27) and pinned here by the
 * public BLS12-381 definition + known-tau identities + agreement with the independent python
 * big-int model (tests/test_oracle_c.py).
 *
 * Formats at this boundary (little-endian bytes):
 *   scalar  : 32 B canonical (value < r)              == Scalar::to_bytes_le
 *   G1 point: 96 B = x(48 B LE) || y(48 B LE), Montgomery form (blst_p1_affine), identity = all 0
 *
 * Build: make -C oracle   (gcc -O2 -shared -fPIC) -> oracle/libkzg_oracle.so
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;

/* ------------------------------------------------------------------------------------------ */
/* generic N-limb Montgomery arithmetic                                                        */
/* ------------------------------------------------------------------------------------------ */
#define DEFINE_FIELD(NAME, N)                                                                      \
    typedef struct { u64 v[N]; } NAME;                                                             \
    static const u64 NAME##_P[N];                                                                  \
    static const u64 NAME##_INV;                                                                   \
    static const NAME NAME##_R1; /* R mod p  (Montgomery one) */                                   \
    static const NAME NAME##_R2; /* R^2 mod p */                                                   \
    static inline int NAME##_is_zero(const NAME *a) {                                              \
        u64 t = 0; for (int i = 0; i < N; i++) t |= a->v[i]; return t == 0; }                      \
    static inline int NAME##_eq(const NAME *a, const NAME *b) {                                    \
        u64 t = 0; for (int i = 0; i < N; i++) t |= a->v[i] ^ b->v[i]; return t == 0; }            \
    static inline int NAME##_geq_p(const u64 *a) {                                                 \
        for (int i = N - 1; i >= 0; i--) { if (a[i] > NAME##_P[i]) return 1;                       \
            if (a[i] < NAME##_P[i]) return 0; } return 1; }                                        \
    static inline void NAME##_sub_p(u64 *a) {                                                      \
        u64 br = 0; for (int i = 0; i < N; i++) { u128 d = (u128)a[i] - NAME##_P[i] - br;          \
            a[i] = (u64)d; br = (u64)(d >> 64) & 1; } }                                            \
    static inline void NAME##_add(NAME *r, const NAME *a, const NAME *b) {                         \
        u64 c = 0; u64 t[N]; for (int i = 0; i < N; i++) { u128 s = (u128)a->v[i] + b->v[i] + c;   \
            t[i] = (u64)s; c = (u64)(s >> 64); }                                                   \
        if (c || NAME##_geq_p(t)) NAME##_sub_p(t); memcpy(r->v, t, sizeof t); }                    \
    static inline void NAME##_sub(NAME *r, const NAME *a, const NAME *b) {                         \
        u64 br = 0; u64 t[N]; for (int i = 0; i < N; i++) { u128 d = (u128)a->v[i] - b->v[i] - br; \
            t[i] = (u64)d; br = (u64)(d >> 64) & 1; }                                              \
        if (br) { u64 c = 0; for (int i = 0; i < N; i++) { u128 s = (u128)t[i] + NAME##_P[i] + c;  \
            t[i] = (u64)s; c = (u64)(s >> 64); } } memcpy(r->v, t, sizeof t); }                    \
    static inline void NAME##_neg(NAME *r, const NAME *a) {                                        \
        NAME z; memset(&z, 0, sizeof z); NAME##_sub(r, &z, a); }                                   \
    static inline void NAME##_mul(NAME *r, const NAME *a, const NAME *b) {                         \
        u64 t[N + 2]; memset(t, 0, sizeof t);                                                      \
        for (int i = 0; i < N; i++) {                                                              \
            u64 c = 0;                                                                             \
            for (int j = 0; j < N; j++) { u128 x = (u128)a->v[j] * b->v[i] + t[j] + c;             \
                t[j] = (u64)x; c = (u64)(x >> 64); }                                               \
            u128 x = (u128)t[N] + c; t[N] = (u64)x; t[N + 1] = (u64)(x >> 64);                     \
            u64 m = t[0] * NAME##_INV;                                                             \
            x = (u128)m * NAME##_P[0] + t[0]; c = (u64)(x >> 64);                                  \
            for (int j = 1; j < N; j++) { x = (u128)m * NAME##_P[j] + t[j] + c;                    \
                t[j - 1] = (u64)x; c = (u64)(x >> 64); }                                           \
            x = (u128)t[N] + c; t[N - 1] = (u64)x; t[N] = t[N + 1] + (u64)(x >> 64);               \
        }                                                                                          \
        if (t[N] || NAME##_geq_p(t)) NAME##_sub_p(t); memcpy(r->v, t, N * sizeof(u64)); }          \
    static inline void NAME##_sqr(NAME *r, const NAME *a) { NAME##_mul(r, a, a); }                 \
    static inline void NAME##_to_mont(NAME *r, const NAME *a) { NAME##_mul(r, a, &NAME##_R2); }    \
    static inline void NAME##_from_mont(NAME *r, const NAME *a) {                                  \
        NAME one; memset(&one, 0, sizeof one); one.v[0] = 1; NAME##_mul(r, a, &one); }             \
    /* r = a^e, e given as nlimbs u64 little-endian (pow_vartime) */                               \
    static void NAME##_pow(NAME *r, const NAME *a, const u64 *e, int nlimbs) {                     \
        NAME acc = NAME##_R1; int started = 0;                                                     \
        for (int i = nlimbs * 64 - 1; i >= 0; i--) {                                               \
            if (started) NAME##_sqr(&acc, &acc);                                                   \
            if ((e[i / 64] >> (i % 64)) & 1) { NAME##_mul(&acc, &acc, a); started = 1; } }         \
        *r = acc; }                                                                                \
    static void NAME##_inv(NAME *r, const NAME *a) { /* Fermat: a^(p-2) */                         \
        u64 e[N]; memcpy(e, NAME##_P, sizeof e); e[0] -= 2; NAME##_pow(r, a, e, N); }

DEFINE_FIELD(fq, 6)
DEFINE_FIELD(fr, 4)

static const u64 fq_P[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                            0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
static const u64 fq_INV = 0x89f3fffcfffcfffdULL;
static const fq fq_R1 = {{0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL,
                          0x77ce585370525745ULL, 0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL}};
static const fq fq_R2 = {{0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL, 0x8de5476c4c95b6d5ULL,
                          0x67eb88a9939d83c0ULL, 0x9a793e85b519952dULL, 0x11988fe592cae3aaULL}};

static const u64 fr_P[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL,
                            0x73eda753299d7d48ULL};
static const u64 fr_INV = 0xfffffffeffffffffULL;
static const fr fr_R1 = {{0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL,
                          0x1824b159acc5056fULL}};
static const fr fr_R2 = {{0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL,
                          0x0748d9d99f59ff11ULL}};

/* ff::PrimeField::root_of_unity() = 7^((r-1)/2^32), canonical (src/ft.rs:73) */
static const u64 FR_ROOT_OF_UNITY[4] = {0x3829971f439f0d2bULL, 0xb63683508c2280b9ULL,
                                        0xd09b681922c813b4ULL, 0x16a2a19edfe81f20ULL};
#define FR_S 32

static void fr_load(fr *r, const uint8_t *b) { fr t; memcpy(t.v, b, 32); fr_to_mont(r, &t); }
static void fr_store(uint8_t *b, const fr *a) { fr t; fr_from_mont(&t, a); memcpy(b, t.v, 32); }
static void fr_from_u64(fr *r, u64 x) { fr t = {{x, 0, 0, 0}}; fr_to_mont(r, &t); }

/* ------------------------------------------------------------------------------------------ */
/* G1 (y^2 = x^3 + 4), Jacobian coordinates, Montgomery form                                   */
/* ------------------------------------------------------------------------------------------ */
typedef struct { fq x, y; } g1a;        /* affine; identity = (0,0) */
typedef struct { fq x, y, z; } g1j;     /* Jacobian; identity z = 0 */

static int g1a_is_inf(const g1a *p) { return fq_is_zero(&p->x) && fq_is_zero(&p->y); }
static void g1j_set_inf(g1j *p) { memset(p, 0, sizeof *p); }

static void g1j_double(g1j *r, const g1j *p) {
    if (fq_is_zero(&p->z) || fq_is_zero(&p->y)) { g1j_set_inf(r); return; }
    fq A, B, C, D, E, F, t;
    fq_sqr(&A, &p->x); fq_sqr(&B, &p->y); fq_sqr(&C, &B);
    fq_add(&t, &p->x, &B); fq_sqr(&t, &t); fq_sub(&t, &t, &A); fq_sub(&t, &t, &C); fq_add(&D, &t, &t);
    fq_add(&E, &A, &A); fq_add(&E, &E, &A);
    fq_sqr(&F, &E);
    fq Z3; fq_mul(&Z3, &p->y, &p->z); fq_add(&Z3, &Z3, &Z3);
    fq X3; fq_sub(&X3, &F, &D); fq_sub(&X3, &X3, &D);
    fq Y3; fq_sub(&t, &D, &X3); fq_mul(&Y3, &E, &t);
    fq c8; fq_add(&c8, &C, &C); fq_add(&c8, &c8, &c8); fq_add(&c8, &c8, &c8);
    fq_sub(&Y3, &Y3, &c8);
    r->x = X3; r->y = Y3; r->z = Z3;
}

static void g1j_add_affine(g1j *r, const g1j *p, const g1a *q) {
    if (g1a_is_inf(q)) { *r = *p; return; }
    if (fq_is_zero(&p->z)) { r->x = q->x; r->y = q->y; r->z = fq_R1; return; }
    fq Z1Z1, U2, S2, H, Rr, HH, HHH, V, t;
    fq_sqr(&Z1Z1, &p->z);
    fq_mul(&U2, &q->x, &Z1Z1);
    fq_mul(&S2, &q->y, &p->z); fq_mul(&S2, &S2, &Z1Z1);
    if (fq_eq(&U2, &p->x)) {
        if (fq_eq(&S2, &p->y)) { g1j_double(r, p); return; }
        g1j_set_inf(r); return;
    }
    fq_sub(&H, &U2, &p->x); fq_sub(&Rr, &S2, &p->y);
    fq_sqr(&HH, &H); fq_mul(&HHH, &H, &HH); fq_mul(&V, &p->x, &HH);
    fq X3; fq_sqr(&X3, &Rr); fq_sub(&X3, &X3, &HHH); fq_sub(&X3, &X3, &V); fq_sub(&X3, &X3, &V);
    fq Y3; fq_sub(&t, &V, &X3); fq_mul(&Y3, &Rr, &t); fq_mul(&t, &p->y, &HHH); fq_sub(&Y3, &Y3, &t);
    fq Z3; fq_mul(&Z3, &p->z, &H);
    r->x = X3; r->y = Y3; r->z = Z3;
}

static void g1j_add(g1j *r, const g1j *p, const g1j *q) {
    if (fq_is_zero(&p->z)) { *r = *q; return; }
    if (fq_is_zero(&q->z)) { *r = *p; return; }
    fq Z1Z1, Z2Z2, U1, U2, S1, S2, H, Rr, HH, HHH, V, t;
    fq_sqr(&Z1Z1, &p->z); fq_sqr(&Z2Z2, &q->z);
    fq_mul(&U1, &p->x, &Z2Z2); fq_mul(&U2, &q->x, &Z1Z1);
    fq_mul(&S1, &p->y, &q->z); fq_mul(&S1, &S1, &Z2Z2);
    fq_mul(&S2, &q->y, &p->z); fq_mul(&S2, &S2, &Z1Z1);
    if (fq_eq(&U1, &U2)) {
        if (fq_eq(&S1, &S2)) { g1j_double(r, p); return; }
        g1j_set_inf(r); return;
    }
    fq_sub(&H, &U2, &U1); fq_sub(&Rr, &S2, &S1);
    fq_sqr(&HH, &H); fq_mul(&HHH, &H, &HH); fq_mul(&V, &U1, &HH);
    fq X3; fq_sqr(&X3, &Rr); fq_sub(&X3, &X3, &HHH); fq_sub(&X3, &X3, &V); fq_sub(&X3, &X3, &V);
    fq Y3; fq_sub(&t, &V, &X3); fq_mul(&Y3, &Rr, &t); fq_mul(&t, &S1, &HHH); fq_sub(&Y3, &Y3, &t);
    fq Z3; fq_mul(&Z3, &p->z, &q->z); fq_mul(&Z3, &Z3, &H);
    r->x = X3; r->y = Y3; r->z = Z3;
}

/* Curve::to_affine (call sites src/coeff_form.rs:63,78,107; src/eval_form.rs:120,139) */
static void g1j_to_affine(g1a *r, const g1j *p) {
    if (fq_is_zero(&p->z)) { memset(r, 0, sizeof *r); return; }
    fq zi, zi2, zi3;
    fq_inv(&zi, &p->z); fq_sqr(&zi2, &zi); fq_mul(&zi3, &zi2, &zi);
    fq_mul(&r->x, &p->x, &zi2); fq_mul(&r->y, &p->y, &zi3);
}

/* batch to_affine with one inversion (Montgomery trick) */
static void g1j_batch_to_affine(g1a *out, const g1j *in, size_t n) {
    fq *pre = (fq *)malloc((n + 1) * sizeof(fq));
    fq acc = fq_R1;
    for (size_t i = 0; i < n; i++) { pre[i] = acc; if (!fq_is_zero(&in[i].z)) fq_mul(&acc, &acc, &in[i].z); }
    fq inv; fq_inv(&inv, &acc);
    for (size_t i = n; i-- > 0;) {
        if (fq_is_zero(&in[i].z)) { memset(&out[i], 0, sizeof(g1a)); continue; }
        fq zi, zi2, zi3; fq_mul(&zi, &inv, &pre[i]); fq_mul(&inv, &inv, &in[i].z);
        fq_sqr(&zi2, &zi); fq_mul(&zi3, &zi2, &zi);
        fq_mul(&out[i].x, &in[i].x, &zi2); fq_mul(&out[i].y, &in[i].y, &zi3);
    }
    free(pre);
}

static const g1a G1_GEN_CANON = {
    {{0xfb3af00adb22c6bbULL, 0x6c55e83ff97a1aefULL, 0xa14e3a3f171bac58ULL, 0xc3688c4f9774b905ULL,
      0x2695638c4fa9ac0fULL, 0x17f1d3a73197d794ULL}},
    {{0x0caa232946c5e7e1ULL, 0xd03cc744a2888ae4ULL, 0x00db18cb2c04b3edULL, 0xfcf5e095d5d00af6ULL,
      0xa09e30ed741d8ae4ULL, 0x08b3f481e3aaa0f1ULL}}};

static void g1_generator(g1a *g) { fq_to_mont(&g->x, &G1_GEN_CANON.x); fq_to_mont(&g->y, &G1_GEN_CANON.y); }

/* [k]P, k canonical 4x u64 */
static void g1_scalar_mul(g1j *r, const g1a *p, const u64 k[4]) {
    g1j acc; g1j_set_inf(&acc);
    for (int i = 255; i >= 0; i--) {
        g1j_double(&acc, &acc);
        if ((k[i / 64] >> (i % 64)) & 1) g1j_add_affine(&acc, &acc, p);
    }
    *r = acc;
}

/* ------------------------------------------------------------------------------------------ */
/* exported: G1                                                                                */
/* ------------------------------------------------------------------------------------------ */
/* G1Projective::multi_exp(points, scalars).to_affine()  -- external blst Pippenger; restated as the
 * textbook bucket method (call sites src/coeff_form.rs:61,78,102; src/eval_form.rs:118,136).
 * Single-threaded, like the reference. */
void orc_msm_g1(const uint8_t *points, const uint8_t *scalars, size_t n, uint8_t *out) {
    const g1a *pts = (const g1a *)points;
    const u64 *sc = (const u64 *)scalars;
    int c = 3;
    if (n >= 32) { c = 0; size_t t = n; while (t >>= 1) c++; c = c > 4 ? c - 3 : 2; if (c > 16) c = 16; }
    int nwin = (255 + c) / c;
    size_t nb = ((size_t)1 << c) - 1;
    g1j *buckets = (g1j *)malloc(nb * sizeof(g1j));
    g1j total; g1j_set_inf(&total);
    for (int w = nwin - 1; w >= 0; w--) {
        for (int i = 0; i < c; i++) g1j_double(&total, &total);
        for (size_t b = 0; b < nb; b++) g1j_set_inf(&buckets[b]);
        int bit = w * c;
        for (size_t i = 0; i < n; i++) {
            const u64 *s = sc + 4 * i;
            u64 d = s[bit / 64] >> (bit % 64);
            if ((bit % 64) + c > 64 && bit / 64 + 1 < 4) d |= s[bit / 64 + 1] << (64 - bit % 64);
            d &= nb;
            if (d) g1j_add_affine(&buckets[d - 1], &buckets[d - 1], &pts[i]);
        }
        g1j run, sum; g1j_set_inf(&run); g1j_set_inf(&sum);
        for (size_t b = nb; b-- > 0;) { g1j_add(&run, &run, &buckets[b]); g1j_add(&sum, &sum, &run); }
        g1j_add(&total, &total, &sum);
    }
    free(buckets);
    g1j_to_affine((g1a *)out, &total);
}

/* ------------------------------------------------------------------------------------------ */
/* TIMING LEG ONLY (bench.py's cpu_baseline): a faster single-threaded Pippenger.  The checker of   */
/* every test stays orc_msm_g1 above; this one is checked against it (tests/test_oracle_c.py).      */
/* What a tuned CPU library does that the textbook loop above does not: signed 16-bit window digits  */
/* (half the buckets), bucket accumulation in AFFINE coordinates with the field inversions of a      */
/* whole batch of additions shared (Montgomery's trick: ~6 multiplications per addition instead of   */
/* the 11 of a mixed Jacobian one), and an unrolled no-carry CIOS Montgomery multiplication.         */
/* ------------------------------------------------------------------------------------------ */
#define FQ_MAC(hi, lo, a, b, c, d) do { u128 _x = (u128)(a) * (b) + (c) + (d); lo = (u64)_x; hi = (u64)(_x >> 64); } while (0)
static inline void fq_mulf(fq *r, const fq *a, const fq *b) {
    /* the modulus' top limb (0x1a01...) leaves spare bits: the carry of a round fits the top word */
    u64 t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
    const u64 *p = fq_P;
#define FQ_ROUND(bi) do { u64 A, C, m, _d; \
        FQ_MAC(A, t0, a->v[0], bi, t0, 0); m = t0 * fq_INV; FQ_MAC(C, _d, m, p[0], t0, 0); (void)_d; \
        FQ_MAC(A, t1, a->v[1], bi, t1, A); FQ_MAC(C, t0, m, p[1], t1, C); \
        FQ_MAC(A, t2, a->v[2], bi, t2, A); FQ_MAC(C, t1, m, p[2], t2, C); \
        FQ_MAC(A, t3, a->v[3], bi, t3, A); FQ_MAC(C, t2, m, p[3], t3, C); \
        FQ_MAC(A, t4, a->v[4], bi, t4, A); FQ_MAC(C, t3, m, p[4], t4, C); \
        FQ_MAC(A, t5, a->v[5], bi, t5, A); FQ_MAC(C, t4, m, p[5], t5, C); \
        t5 = C + A; } while (0)
    FQ_ROUND(b->v[0]); FQ_ROUND(b->v[1]); FQ_ROUND(b->v[2]); FQ_ROUND(b->v[3]); FQ_ROUND(b->v[4]); FQ_ROUND(b->v[5]);
#undef FQ_ROUND
    u64 t[6] = {t0, t1, t2, t3, t4, t5};
    if (fq_geq_p(t)) fq_sub_p(t);
    memcpy(r->v, t, sizeof t);
}

/* Jacobian helpers on the fast multiplication (same formulas as above) */
static void g1j_add_affine_f(g1j *r, const g1j *p, const g1a *q) {
    if (g1a_is_inf(q)) { *r = *p; return; }
    if (fq_is_zero(&p->z)) { r->x = q->x; r->y = q->y; r->z = fq_R1; return; }
    fq Z1Z1, U2, S2, H, Rr, HH, HHH, V, t;
    fq_mulf(&Z1Z1, &p->z, &p->z);
    fq_mulf(&U2, &q->x, &Z1Z1);
    fq_mulf(&S2, &q->y, &p->z); fq_mulf(&S2, &S2, &Z1Z1);
    if (fq_eq(&U2, &p->x)) {
        if (fq_eq(&S2, &p->y)) { g1j_double(r, p); return; }
        g1j_set_inf(r); return;
    }
    fq_sub(&H, &U2, &p->x); fq_sub(&Rr, &S2, &p->y);
    fq_mulf(&HH, &H, &H); fq_mulf(&HHH, &H, &HH); fq_mulf(&V, &p->x, &HH);
    fq X3; fq_mulf(&X3, &Rr, &Rr); fq_sub(&X3, &X3, &HHH); fq_sub(&X3, &X3, &V); fq_sub(&X3, &X3, &V);
    fq Y3; fq_sub(&t, &V, &X3); fq_mulf(&Y3, &Rr, &t); fq_mulf(&t, &p->y, &HHH); fq_sub(&Y3, &Y3, &t);
    fq Z3; fq_mulf(&Z3, &p->z, &H);
    r->x = X3; r->y = Y3; r->z = Z3;
}
static void g1j_add_f(g1j *r, const g1j *p, const g1j *q) {
    if (fq_is_zero(&p->z)) { *r = *q; return; }
    if (fq_is_zero(&q->z)) { *r = *p; return; }
    fq Z1Z1, Z2Z2, U1, U2, S1, S2, H, Rr, HH, HHH, V, t;
    fq_mulf(&Z1Z1, &p->z, &p->z); fq_mulf(&Z2Z2, &q->z, &q->z);
    fq_mulf(&U1, &p->x, &Z2Z2); fq_mulf(&U2, &q->x, &Z1Z1);
    fq_mulf(&S1, &p->y, &q->z); fq_mulf(&S1, &S1, &Z2Z2);
    fq_mulf(&S2, &q->y, &p->z); fq_mulf(&S2, &S2, &Z1Z1);
    if (fq_eq(&U1, &U2)) {
        if (fq_eq(&S1, &S2)) { g1j_double(r, p); return; }
        g1j_set_inf(r); return;
    }
    fq_sub(&H, &U2, &U1); fq_sub(&Rr, &S2, &S1);
    fq_mulf(&HH, &H, &H); fq_mulf(&HHH, &H, &HH); fq_mulf(&V, &U1, &HH);
    fq X3; fq_mulf(&X3, &Rr, &Rr); fq_sub(&X3, &X3, &HHH); fq_sub(&X3, &X3, &V); fq_sub(&X3, &X3, &V);
    fq Y3; fq_sub(&t, &V, &X3); fq_mulf(&Y3, &Rr, &t); fq_mulf(&t, &S1, &HHH); fq_sub(&Y3, &Y3, &t);
    fq Z3; fq_mulf(&Z3, &p->z, &q->z); fq_mulf(&Z3, &Z3, &H);
    r->x = X3; r->y = Y3; r->z = Z3;
}

#define FAST_C 16                 /* window width: 16 windows cover 255 bits + the carry of the signed recoding */
#define FAST_NWIN 16
#define FAST_BATCH 1024           /* affine additions that share one field inversion */
typedef struct { uint32_t bucket; uint32_t idx; } fast_job;   /* idx: point index, top bit = negated */

/* one batch of affine additions bucket[b] += pt, all to DIFFERENT non-empty buckets: one inversion for all of them */
static void fast_flush(g1a *buckets, uint8_t *busy, const g1a *pts, fast_job *jobs, size_t m, fq *den, fq *pre, g1a *tmp) {
    if (!m) return;
    fq acc = fq_R1;
    for (size_t k = 0; k < m; k++) {          /* denominators: x2 - x1, or 2 y for a doubling, or 1 for P + (-P) */
        g1a *B = &buckets[jobs[k].bucket];
        tmp[k] = pts[jobs[k].idx & 0x7fffffffu];
        if (jobs[k].idx >> 31) fq_neg(&tmp[k].y, &tmp[k].y);
        fq_sub(&den[k], &tmp[k].x, &B->x);
        if (fq_is_zero(&den[k])) {
            if (fq_eq(&tmp[k].y, &B->y)) fq_add(&den[k], &B->y, &B->y);   /* same point (y != 0 on this curve's prime-order group) */
            else den[k] = fq_R1;                                                  /* opposite points: the sum is the identity */
        }
        pre[k] = acc;
        fq_mulf(&acc, &acc, &den[k]);
    }
    fq inv; fq_inv(&inv, &acc);
    for (size_t k = m; k-- > 0;) {
        g1a *B = &buckets[jobs[k].bucket];
        const g1a *P = &tmp[k];
        fq dinv; fq_mulf(&dinv, &inv, &pre[k]); fq_mulf(&inv, &inv, &den[k]);
        fq num, lam, x3, y3, t;
        fq_sub(&t, &P->x, &B->x);
        if (fq_is_zero(&t)) {
            if (!fq_eq(&P->y, &B->y)) { memset(B, 0, sizeof *B); busy[jobs[k].bucket] = 0; continue; }   /* identity: bucket empty again */
            fq_mulf(&num, &B->x, &B->x); fq_add(&t, &num, &num); fq_add(&num, &t, &num);                  /* 3 x^2 (a = 0) */
        } else {
            fq_sub(&num, &P->y, &B->y);
        }
        fq_mulf(&lam, &num, &dinv);
        fq_mulf(&x3, &lam, &lam); fq_sub(&x3, &x3, &B->x); fq_sub(&x3, &x3, &P->x);
        fq_sub(&t, &B->x, &x3); fq_mulf(&y3, &lam, &t); fq_sub(&y3, &y3, &B->y);
        B->x = x3; B->y = y3;
        busy[jobs[k].bucket] = 0;
    }
}

void orc_msm_g1_fast(const uint8_t *points, const uint8_t *scalars, size_t n, uint8_t *out) {
    const g1a *pts = (const g1a *)points;
    const u64 *sc = (const u64 *)scalars;
    const size_t nb = (size_t)1 << (FAST_C - 1);
    int16_t *dig = (int16_t *)malloc(n * FAST_NWIN * sizeof(int16_t) + 16);
    for (size_t i = 0; i < n; i++) {          /* signed digits in [-2^15, 2^15): d_w - 2^16 and a carry when d_w >= 2^15 */
        const u64 *s = sc + 4 * i;
        int carry = 0;
        for (int w = 0; w < FAST_NWIN; w++) {
            int d = (int)((s[w / 4] >> (16 * (w % 4))) & 0xffff) + carry;
            carry = d >= 0x8000;
            dig[i * FAST_NWIN + w] = (int16_t)(d - (carry << 16));
        }                                      /* canonical scalars are below 2^255: the last digit never carries */
    }
    g1a *buckets = (g1a *)malloc(nb * sizeof(g1a));
    uint8_t *busy = (uint8_t *)calloc(nb, 1);
    uint8_t *full = (uint8_t *)malloc(nb);
    fast_job *jobs = (fast_job *)malloc(FAST_BATCH * sizeof(fast_job));
    fast_job *defer = (fast_job *)malloc((n + 1) * sizeof(fast_job)), *defer2 = (fast_job *)malloc((n + 1) * sizeof(fast_job));
    fq *den = (fq *)malloc(FAST_BATCH * sizeof(fq)), *pre = (fq *)malloc(FAST_BATCH * sizeof(fq));
    g1a *tmp = (g1a *)malloc(FAST_BATCH * sizeof(g1a));
    g1j total; g1j_set_inf(&total);
    for (int w = FAST_NWIN - 1; w >= 0; w--) {
        for (int i = 0; i < FAST_C; i++) g1j_double(&total, &total);
        memset(full, 0, nb);
        size_t m = 0, nd = 0;
        for (size_t i = 0; i < n; i++) {
            int d = dig[i * FAST_NWIN + w];
            if (!d || g1a_is_inf(&pts[i])) continue;
            fast_job j;
            j.bucket = (uint32_t)((d < 0 ? -d : d) - 1);
            j.idx = (uint32_t)i | (d < 0 ? 0x80000000u : 0u);
            if (busy[j.bucket]) { defer[nd++] = j; continue; }                                  /* already in this batch: later */
            if (!full[j.bucket] || g1a_is_inf(&buckets[j.bucket])) {                            /* first point of the bucket (or emptied by P + (-P)) */
                buckets[j.bucket] = pts[i];
                if (d < 0) fq_neg(&buckets[j.bucket].y, &buckets[j.bucket].y);
                full[j.bucket] = 1; continue; }
            __builtin_prefetch(&buckets[j.bucket]);                                             /* the flush finds it in cache */
            busy[j.bucket] = 1;
            jobs[m++] = j;
            if (m == FAST_BATCH) { fast_flush(buckets, busy, pts, jobs, m, den, pre, tmp); m = 0; }
        }
        fast_flush(buckets, busy, pts, jobs, m, den, pre, tmp);
        m = 0;
        while (nd) {                             /* the deferred additions, in rounds: each round takes one per bucket */
            size_t nd2 = 0;
            for (size_t k = 0; k < nd; k++) {
                fast_job *j = &defer[k];
                if (busy[j->bucket]) { defer2[nd2++] = *j; continue; }
                if (g1a_is_inf(&buckets[j->bucket])) {                                           /* emptied by P + (-P) */
                    buckets[j->bucket] = pts[j->idx & 0x7fffffffu];
                    if (j->idx >> 31) fq_neg(&buckets[j->bucket].y, &buckets[j->bucket].y);
                    continue; }
                busy[j->bucket] = 1;
                jobs[m++] = *j;
                if (m == FAST_BATCH) { fast_flush(buckets, busy, pts, jobs, m, den, pre, tmp); m = 0; }
            }
            fast_flush(buckets, busy, pts, jobs, m, den, pre, tmp);
            m = 0;
            fast_job *tq = defer; defer = defer2; defer2 = tq;
            nd = nd2;
        }
        g1j run, sum; g1j_set_inf(&run); g1j_set_inf(&sum);
        for (size_t b = nb; b-- > 0;) {
            if (full[b] && !g1a_is_inf(&buckets[b])) g1j_add_affine_f(&run, &run, &buckets[b]);
            g1j_add_f(&sum, &sum, &run);
        }
        g1j_add_f(&total, &total, &sum);
    }
    free(dig); free(buckets); free(busy); free(full); free(jobs); free(defer); free(defer2); free(den); free(pre); free(tmp);
    g1j_to_affine((g1a *)out, &total);
}

/* naive sum_i [s_i]P_i (independent of the bucket method; cross-check for orc_msm_g1) */
void orc_msm_g1_naive(const uint8_t *points, const uint8_t *scalars, size_t n, uint8_t *out) {
    g1j total; g1j_set_inf(&total);
    for (size_t i = 0; i < n; i++) {
        g1j t; g1_scalar_mul(&t, (const g1a *)points + i, (const u64 *)scalars + 4 * i);
        g1j_add(&total, &total, &t);
    }
    g1j_to_affine((g1a *)out, &total);
}

void orc_g1_generator(uint8_t *out) { g1_generator((g1a *)out); }

void orc_g1_mul(const uint8_t *point, const uint8_t *scalar, uint8_t *out) {
    g1j t; g1_scalar_mul(&t, (const g1a *)point, (const u64 *)scalar); g1j_to_affine((g1a *)out, &t);
}

void orc_g1_add(const uint8_t *a, const uint8_t *b, uint8_t *out) {
    g1j t; const g1a *pa = (const g1a *)a;
    if (g1a_is_inf(pa)) g1j_set_inf(&t); else { t.x = pa->x; t.y = pa->y; t.z = fq_R1; }
    g1j_add_affine(&t, &t, (const g1a *)b); g1j_to_affine((g1a *)out, &t);
}

int orc_g1_on_curve(const uint8_t *a) {
    const g1a *p = (const g1a *)a; if (g1a_is_inf(p)) return 1;
    fq l, r, four, t; fq_sqr(&l, &p->y); fq_sqr(&r, &p->x); fq_mul(&r, &r, &p->x);
    fq c4 = {{4, 0, 0, 0, 0, 0}}; fq_to_mont(&four, &c4); fq_add(&t, &r, &four); return fq_eq(&l, &t);
}

/* 96 B affine-Montgomery <-> 96 B zcash uncompressed (big-endian canonical x||y, 0x40 flag = inf) */
void orc_g1_to_uncompressed(const uint8_t *a, uint8_t *out) {
    const g1a *p = (const g1a *)a; memset(out, 0, 96);
    if (g1a_is_inf(p)) { out[0] = 0x40; return; }
    fq x, y; fq_from_mont(&x, &p->x); fq_from_mont(&y, &p->y);
    for (int i = 0; i < 48; i++) { out[47 - i] = (uint8_t)(x.v[i / 8] >> (8 * (i % 8)));
                                   out[95 - i] = (uint8_t)(y.v[i / 8] >> (8 * (i % 8))); }
}
void orc_g1_from_uncompressed(const uint8_t *in, uint8_t *out) {
    g1a *p = (g1a *)out; memset(p, 0, sizeof *p);
    if (in[0] & 0x40) return;
    fq x, y; memset(&x, 0, sizeof x); memset(&y, 0, sizeof y);
    for (int i = 0; i < 48; i++) { x.v[i / 8] |= (u64)in[47 - i] << (8 * (i % 8));
                                   y.v[i / 8] |= (u64)in[95 - i] << (8 * (i % 8)); }
    fq_to_mont(&p->x, &x); fq_to_mont(&p->y, &y);
}

/* setup(): gs[i] = gs[i-1] * s  (src/lib.rs:38-47), G1 half.  Implemented as gs[i] = [s^i]G with an
 * 8-bit fixed-base table so 2^16..2^18 points are feasible; identical group elements. */
void orc_setup_g1(const uint8_t *tau, size_t n, uint8_t *out) {
    if (n == 0) return;
    g1a gen; g1_generator(&gen);
    /* table[j][d-1] = d * 2^(8j) * G, j < 32, d in 1..255 */
    g1j *tj = (g1j *)malloc(32 * 255 * sizeof(g1j));
    g1a *ta = (g1a *)malloc(32 * 255 * sizeof(g1a));
    g1j base; base.x = gen.x; base.y = gen.y; base.z = fq_R1;
    for (int j = 0; j < 32; j++) {
        g1j acc = base;
        for (int d = 1; d <= 255; d++) { tj[j * 255 + d - 1] = acc; g1j_add(&acc, &acc, &base); }
        base = acc; /* 256 * base */
    }
    g1j_batch_to_affine(ta, tj, 32 * 255);
    free(tj);
    fr t, pw = fr_R1; fr_load(&t, tau);
    g1j *res = (g1j *)malloc(n * sizeof(g1j));
    for (size_t i = 0; i < n; i++) {
        uint8_t e[32]; fr_store(e, &pw);
        g1j acc; g1j_set_inf(&acc);
        for (int j = 0; j < 32; j++) if (e[j]) g1j_add_affine(&acc, &acc, &ta[j * 255 + e[j] - 1]);
        res[i] = acc;
        fr_mul(&pw, &pw, &t);
    }
    g1j_batch_to_affine((g1a *)out, res, n);
    free(res); free(ta);
}

/* ------------------------------------------------------------------------------------------ */
/* exported: Fr / polynomial / NTT                                                             */
/* ------------------------------------------------------------------------------------------ */
/* EvaluationDomain::compute_omega (src/ft.rs:55-76): returns 0 ok, 2 = PolynomialDegreeTooLarge */
int orc_compute_omega(u64 d, u64 *m_out, uint32_t *exp_out, uint8_t *omega_out) {
    u64 m = 1; uint32_t exp = 0;
    while (m < d) { m *= 2; exp += 1; if (exp >= FR_S) return 2; }
    fr root, w; fr t; memcpy(t.v, FR_ROOT_OF_UNITY, 32); fr_to_mont(&root, &t);
    u64 e[1] = {1ULL << (FR_S - exp)};
    fr_pow(&w, &root, e, 1);
    *m_out = m; *exp_out = exp; fr_store(omega_out, &w);
    return 0;
}

static uint32_t bitreverse(uint32_t n, uint32_t l) {
    uint32_t r = 0; for (uint32_t i = 0; i < l; i++) { r = (r << 1) | (n & 1); n >>= 1; } return r; }

/* fr_mul unrolled (no-carry CIOS: r's top limb 0x73ed... leaves a spare bit): the same product, ~2x faster -- the full-size NTT
 * parity tests spend their time in the two multiplications per butterfly below */
static inline void fr_mulf(fr *r, const fr *a, const fr *b) {
    u64 t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    const u64 *p = fr_P;
#define FR_MAC(hi, lo, x, y, c, d) do { u128 _x = (u128)(x) * (y) + (c) + (d); lo = (u64)_x; hi = (u64)(_x >> 64); } while (0)
#define FR_ROUND(bi) do { u64 A, C, m, _d; \
        FR_MAC(A, t0, a->v[0], bi, t0, 0); m = t0 * fr_INV; FR_MAC(C, _d, m, p[0], t0, 0); (void)_d; \
        FR_MAC(A, t1, a->v[1], bi, t1, A); FR_MAC(C, t0, m, p[1], t1, C); \
        FR_MAC(A, t2, a->v[2], bi, t2, A); FR_MAC(C, t1, m, p[2], t2, C); \
        FR_MAC(A, t3, a->v[3], bi, t3, A); FR_MAC(C, t2, m, p[3], t3, C); \
        t3 = C + A; } while (0)
    FR_ROUND(b->v[0]); FR_ROUND(b->v[1]); FR_ROUND(b->v[2]); FR_ROUND(b->v[3]);
#undef FR_ROUND
#undef FR_MAC
    u64 t[4] = {t0, t1, t2, t3};
    if (fr_geq_p(t)) fr_sub_p(t);
    memcpy(r->v, t, sizeof t);
}

/* serial_fft (src/ft.rs:291-333) on Montgomery-form data */
static void serial_fft_mont(fr *a, const fr *omega, uint32_t log_n) {
    /* The reference's loop nest, butterfly for butterfly.  With OpenMP (the Makefile asks for it when the compiler has it) the
     * independent butterflies of a stage are spread over the host's cores -- whole groups while there are many, chunks of one
     * group's j-range (each chunk starting from w_m^j0 instead of reaching it by repeated multiplication) once there are few.
     * Field elements are canonical, so the results are the same bits; tests/test_oracle_c.py compares with the python model. */
    uint32_t n = 1u << log_n;
#pragma omp parallel for schedule(static) if (log_n >= 16)
    for (uint32_t k = 0; k < n; k++) { uint32_t rk = bitreverse(k, log_n);
        if (k < rk) { fr t = a[rk]; a[rk] = a[k]; a[k] = t; } }
    uint32_t m = 1;
    for (uint32_t s = 0; s < log_n; s++) {
        u64 e[1] = {n / (2 * m)}; fr w_m; fr_pow(&w_m, omega, e, 1);
        const uint32_t groups = n / (2 * m);
        if (groups >= 256 || log_n < 16) {
#pragma omp parallel for schedule(static) if (log_n >= 16)
            for (uint32_t g = 0; g < groups; g++) {
                const uint32_t k = g * 2 * m;
                fr w = fr_R1;
                for (uint32_t j = 0; j < m; j++) {
                    fr t; fr_mulf(&t, &a[k + j + m], &w);
                    fr tmp; fr_sub(&tmp, &a[k + j], &t); a[k + j + m] = tmp;
                    fr_add(&a[k + j], &a[k + j], &t);
                    fr_mulf(&w, &w, &w_m);
                }
            }
        } else {
            const uint32_t chunk = m >= 4096 ? 4096 : m;   /* m / chunk chunks per group */
            const uint32_t per = m / chunk;
#pragma omp parallel for schedule(static)
            for (uint32_t c = 0; c < groups * per; c++) {
                const uint32_t k = (c / per) * 2 * m, j0 = (c % per) * chunk;
                u64 ej[1] = {j0}; fr w; fr_pow(&w, &w_m, ej, 1);
                for (uint32_t j = j0; j < j0 + chunk; j++) {
                    fr t; fr_mulf(&t, &a[k + j + m], &w);
                    fr tmp; fr_sub(&tmp, &a[k + j], &t); a[k + j + m] = tmp;
                    fr_add(&a[k + j], &a[k + j], &t);
                    fr_mulf(&w, &w, &w_m);
                }
            }
        }
        m *= 2;
    }
}

void orc_fft(uint8_t *data, uint32_t log_n, int inverse) {
    size_t n = (size_t)1 << log_n;
    u64 m; uint32_t exp; uint8_t wb[32];
    orc_compute_omega(n, &m, &exp, wb);
    fr omega; fr_load(&omega, wb);
    if (inverse) fr_inv(&omega, &omega);
    fr *a = (fr *)malloc(n * sizeof(fr));
    for (size_t i = 0; i < n; i++) fr_load(&a[i], data + 32 * i);
    serial_fft_mont(a, &omega, log_n);
    if (inverse) { fr minv; fr_from_u64(&minv, n); fr_inv(&minv, &minv);
        for (size_t i = 0; i < n; i++) fr_mul(&a[i], &a[i], &minv); }
    for (size_t i = 0; i < n; i++) fr_store(data + 32 * i, &a[i]);
    free(a);
}

/* Polynomial::eval (src/polynomial.rs:156-165) over coeffs[0..n) */
void orc_poly_eval(const uint8_t *coeffs, size_t n, const uint8_t *x, uint8_t *out) {
    fr xx, res; fr_load(&xx, x); fr_load(&res, coeffs + 32 * (n - 1));
    for (size_t i = n - 1; i-- > 0;) { fr c; fr_load(&c, coeffs + 32 * i); fr_mul(&res, &res, &xx); fr_add(&res, &res, &c); }
    fr_store(out, &res);
}

/* Polynomial::long_division (src/polynomial.rs:193-227) for num of exact degree n-1 (>= m-1) and
 * divisor of exact degree m-1 (lead != 0).  quot gets n-m+1 coeffs, rem gets m-1 coeffs.
 * Returns 1 if the remainder is non-zero (reference: Some(remainder)), 0 if exact (None). */
int orc_poly_long_division(const uint8_t *num, size_t n, const uint8_t *den, size_t m, uint8_t *quot, uint8_t *rem) {
    fr *r = (fr *)malloc(n * sizeof(fr)); fr *d = (fr *)malloc(m * sizeof(fr));
    for (size_t i = 0; i < n; i++) fr_load(&r[i], num + 32 * i);
    for (size_t i = 0; i < m; i++) fr_load(&d[i], den + 32 * i);
    fr lead_inv; fr_inv(&lead_inv, &d[m - 1]);
    for (size_t i = n - m + 1; i-- > 0;) {     /* i = remainder.degree - divisor.degree */
        fr factor; fr_mul(&factor, &r[i + m - 1], &lead_inv);
        fr_store(quot + 32 * i, &factor);
        if (fr_is_zero(&factor)) continue;     /* reference: shrink_degree skipped this slot */
        for (size_t j = 0; j < m; j++) { fr t; fr_mul(&t, &d[j], &factor); fr_sub(&r[i + j], &r[i + j], &t); }
    }
    int nz = 0;
    for (size_t i = 0; i + 1 < m; i++) { fr_store(rem + 32 * i, &r[i]); nz |= !fr_is_zero(&r[i]); }
    free(r); free(d);
    return nz;
}

/* the quotient used by KZGProver::create_witness (src/coeff_form.rs:66-81):
 * (p - y) / (X - x) by long_division; returns 1 (-> PointNotOnPolynomial) if remainder != 0. */
int orc_witness_quotient(const uint8_t *coeffs, size_t n, const uint8_t *x, const uint8_t *y, uint8_t *quot) {
    uint8_t *num = (uint8_t *)malloc(32 * n); memcpy(num, coeffs, 32 * n);
    fr c0, yy; fr_load(&c0, num); fr_load(&yy, y); fr_sub(&c0, &c0, &yy); fr_store(num, &c0);
    uint8_t den[64]; fr xx, one = fr_R1; fr_load(&xx, x); fr_neg(&xx, &xx); fr_store(den, &xx); fr_store(den + 32, &one);
    uint8_t rem[32];
    int nz = orc_poly_long_division(num, n, den, 2, quot, rem);
    free(num); return nz;
}

/* div_by_omega_i (src/eval_form.rs:58-84) in its closed form q_j = f_j/(w^j-w^m),
 * q_m = -sum_{i!=m} q_i w^(i-m)  (same field elements), with one batch inversion.
 * evals: d canonical scalars (already minus y); out: d scalars. */
void orc_div_by_omega_i(const uint8_t *evals, size_t d, size_t m, uint8_t *out) {
    u64 mm; uint32_t exp; uint8_t wb[32]; orc_compute_omega(d, &mm, &exp, wb);
    fr w; fr_load(&w, wb);
    fr *pw = (fr *)malloc(d * sizeof(fr)), *den = (fr *)malloc(d * sizeof(fr)), *pre = (fr *)malloc(d * sizeof(fr));
    pw[0] = fr_R1; for (size_t i = 1; i < d; i++) fr_mul(&pw[i], &pw[i - 1], &w);
    fr acc = fr_R1;
    for (size_t j = 0; j < d; j++) { pre[j] = acc; if (j != m) { fr_sub(&den[j], &pw[j], &pw[m]); fr_mul(&acc, &acc, &den[j]); } }
    fr inv; if (d > 1) fr_inv(&inv, &acc); else inv = fr_R1;
    fr qm; memset(&qm, 0, sizeof qm);
    for (size_t j = d; j-- > 0;) {
        if (j == m) continue;
        fr dinv; fr_mul(&dinv, &inv, &pre[j]); fr_mul(&inv, &inv, &den[j]);
        fr f, q; fr_load(&f, evals + 32 * j); fr_mul(&q, &f, &dinv); fr_store(out + 32 * j, &q);
        fr t; fr_mul(&t, &q, &pw[(j + d - m) % d]); fr_sub(&qm, &qm, &t);
    }
    fr_store(out + 32 * m, &qm);
    free(pw); free(den); free(pre);
}

/* Fr helpers for tests */
void orc_fr_mul(const uint8_t *a, const uint8_t *b, uint8_t *out) { fr x, y; fr_load(&x, a); fr_load(&y, b); fr_mul(&x, &x, &y); fr_store(out, &x); }
void orc_fr_inv(const uint8_t *a, uint8_t *out) { fr x; fr_load(&x, a); fr_inv(&x, &x); fr_store(out, &x); }
void orc_fr_to_mont(const uint8_t *a, uint8_t *out) { fr x; fr_load(&x, a); memcpy(out, x.v, 32); }
void orc_fr_from_mont(const uint8_t *a, uint8_t *out) { fr x; memcpy(x.v, a, 32); fr_store(out, &x); }
