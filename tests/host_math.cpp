// Host build of kzg_amd/csrc/{field,curve}.h for CPU-side unit tests (tests/test_host_math.py).
// The same headers are compiled by hipcc for gfx950; this exercises the identical arithmetic source.
#include "../kzg_amd/csrc/curve.h"
using namespace kzg;
extern "C" {
void hm_fq_mul(const uint32_t *a, const uint32_t *b, uint32_t *o) { Fq x, y; memcpy(x.v, a, 48); memcpy(y.v, b, 48); Fq z = mul(x, y); memcpy(o, z.v, 48); }
void hm_fq_add(const uint32_t *a, const uint32_t *b, uint32_t *o) { Fq x, y; memcpy(x.v, a, 48); memcpy(y.v, b, 48); Fq z = add(x, y); memcpy(o, z.v, 48); }
void hm_fq_sub(const uint32_t *a, const uint32_t *b, uint32_t *o) { Fq x, y; memcpy(x.v, a, 48); memcpy(y.v, b, 48); Fq z = sub(x, y); memcpy(o, z.v, 48); }
void hm_fq_inv(const uint32_t *a, uint32_t *o) { Fq x; memcpy(x.v, a, 48); Fq z = inv(x); memcpy(o, z.v, 48); }
void hm_fr_mul(const uint32_t *a, const uint32_t *b, uint32_t *o) { Fr x, y; memcpy(x.v, a, 32); memcpy(y.v, b, 32); Fr z = mul(x, y); memcpy(o, z.v, 32); }
void hm_fr_add(const uint32_t *a, const uint32_t *b, uint32_t *o) { Fr x, y; memcpy(x.v, a, 32); memcpy(y.v, b, 32); Fr z = add(x, y); memcpy(o, z.v, 32); }
void hm_fr_sub(const uint32_t *a, const uint32_t *b, uint32_t *o) { Fr x, y; memcpy(x.v, a, 32); memcpy(y.v, b, 32); Fr z = sub(x, y); memcpy(o, z.v, 32); }
void hm_fr_inv(const uint32_t *a, uint32_t *o) { Fr x; memcpy(x.v, a, 32); Fr z = inv(x); memcpy(o, z.v, 32); }
void hm_fr_to_mont(const uint32_t *a, uint32_t *o) { Fr x; memcpy(x.v, a, 32); Fr z = to_mont(x); memcpy(o, z.v, 32); }
void hm_fr_from_mont(const uint32_t *a, uint32_t *o) { Fr x; memcpy(x.v, a, 32); Fr z = from_mont(x); memcpy(o, z.v, 32); }
void hm_fr_root_of_unity(uint32_t *o) { Fr z = from_mont(fr_root_of_unity()); memcpy(o, z.v, 32); }
void hm_g1_generator(uint32_t *o) { G1Affine g = g1_generator(); memcpy(o, &g, 96); }
// acc(affine a) + b via madd, then via add(xyzz,xyzz), then dbl; all returned affine
void hm_g1_madd(const uint32_t *a, const uint32_t *b, uint32_t *o) { G1Affine x, y; memcpy(&x, a, 96); memcpy(&y, b, 96);
    G1Affine r = g1_to_affine(g1_madd(G1Xyzz::from_affine(x), y)); memcpy(o, &r, 96); }
void hm_g1_add(const uint32_t *a, const uint32_t *b, uint32_t *o) { G1Affine x, y; memcpy(&x, a, 96); memcpy(&y, b, 96);
    // de-normalise both operands first so the general add sees non-trivial ZZ/ZZZ
    G1Xyzz p = g1_dbl(G1Xyzz::from_affine(x)); p = g1_madd(p, g1_neg(x));
    G1Xyzz q = g1_dbl(G1Xyzz::from_affine(y)); q = g1_madd(q, g1_neg(y));
    G1Affine r = g1_to_affine(g1_add(p, q)); memcpy(o, &r, 96); }
void hm_g1_mul(const uint32_t *a, const uint32_t *k, uint32_t *o) { G1Affine x; memcpy(&x, a, 96);
    G1Affine r = g1_to_affine(g1_scalar_mul(x, k)); memcpy(o, &r, 96); }
void hm_g1_jac_roundtrip(const uint32_t *a, const uint32_t *k, uint32_t *o) { G1Affine x; memcpy(&x, a, 96);
    G1Xyzz p = g1_scalar_mul(x, k); G1Jacobian j = g1_to_jacobian(p); G1Affine r = g1_to_affine(g1_from_jacobian(j)); memcpy(o, &r, 96); }
int hm_g1_on_curve(const uint32_t *a) { G1Affine x; memcpy(&x, a, 96); return g1_on_curve(x); }
}
extern "C" {
void hm_fq_inv_fermat(const uint32_t *a, uint32_t *o) { Fq x; memcpy(x.v, a, 48); Fq z = inv_fermat(x); memcpy(o, z.v, 48); }
void hm_fr_inv_fermat(const uint32_t *a, uint32_t *o) { Fr x; memcpy(x.v, a, 32); Fr z = inv_fermat(x); memcpy(o, z.v, 32); }
}
#include "../kzg_amd/csrc/curve30.h"
#include "../kzg_amd/csrc/naf.h"
extern "C" {
// Fq30: x*R384 (48 B) -> to30 -> mul30 -> from30 -> 48 B, must equal the saturated Montgomery product
void hm_mul30(const uint32_t *a, const uint32_t *b, uint32_t *o) { Fq x, y; memcpy(x.v, a, 48); memcpy(y.v, b, 48);
    Fq z = from30(mul30(to30(x), to30(y))); memcpy(o, z.v, 48); }
void hm_sqr30(const uint32_t *a, uint32_t *o) { Fq x; memcpy(x.v, a, 48); Fq z = from30(sqr30(to30(x))); memcpy(o, z.v, 48); }
void hm_roundtrip30(const uint32_t *a, uint32_t *o) { Fq x; memcpy(x.v, a, 48); Fq z = from30(to30(x)); memcpy(o, z.v, 48); }
void hm_packunpack30(const uint32_t *a, uint32_t *o) { Fq x; memcpy(x.v, a, 48); Fq z = pack30(unpack30(x)); memcpy(o, z.v, 48); }
// raw limb interfaces (13 x int32 each): the exact integers going in and out, for the overflow / bound tests
void hm_mul30_raw(const int32_t *a, const int32_t *b, int32_t *o) { Fq30 x, y; memcpy(x.v, a, 52); memcpy(y.v, b, 52);
    Fq30 z = mul30(x, y); memcpy(o, z.v, 52); }
void hm_sqr30_raw(const int32_t *a, int32_t *o) { Fq30 x; memcpy(x.v, a, 52); Fq30 z = sqr30(x); memcpy(o, z.v, 52); }
void hm_muladd30_raw(const int32_t *a, const int32_t *b, const int32_t *c, const int32_t *d, int32_t *o) {
    Fq30 x, y, u, w; memcpy(x.v, a, 52); memcpy(y.v, b, 52); memcpy(u.v, c, 52); memcpy(w.v, d, 52);
    Fq30 z = muladd30_inline(x, y, u, w); memcpy(o, z.v, 52); }
void hm_mul30_sub_raw(const int32_t *a, const int32_t *b, const int32_t *c, int32_t *o) { Fq30 x, y, u; memcpy(x.v, a, 52); memcpy(y.v, b, 52);
    memcpy(u.v, c, 52); Fq30 z = mul30_sub(x, y, u); memcpy(o, z.v, 52); }
void hm_sqr30_sub2_raw(const int32_t *a, const int32_t *c, const int32_t *e, int32_t *o) { Fq30 x, u, w; memcpy(x.v, a, 52); memcpy(u.v, c, 52);
    memcpy(w.v, e, 52); Fq30 z = sqr30_sub2(x, u, w); memcpy(o, z.v, 52); }
// width-18 NAF recoding (naf.h): k as 8 limbs (below 2^254) -> up to 15 digit records
int hm_naf18(const uint32_t *k, uint32_t flip, uint32_t *out) { uint32_t L[12]; for (int i = 0; i < 8; i++) L[i] = k[i]; L[8] = L[9] = L[10] = L[11] = 0;
    return naf18_digits(L, flip, out); }
void hm_mul30u_raw(const int32_t *a, const int32_t *b, int32_t *o) { Fq30 x, y; memcpy(x.v, a, 52); memcpy(y.v, b, 52);
    Fq30 z = mul30u(x, y); memcpy(o, z.v, 52); }
void hm_sqr30_sub2u_raw(const int32_t *a, const int32_t *c, const int32_t *e, int32_t *o) { Fq30 x, u, w; memcpy(x.v, a, 52); memcpy(u.v, c, 52);
    memcpy(w.v, e, 52); Fq30 z = sqr30_sub2u(x, u, w); memcpy(o, z.v, 52); }
void hm_normalize30_raw(const int32_t *a, int32_t *o) { Fq30 x; memcpy(x.v, a, 52); Fq30 z = normalize30(x); memcpy(o, z.v, 52); }
void hm_from30_raw(const int32_t *a, uint32_t *o) { Fq30 x; memcpy(x.v, a, 52); Fq z = from30(x); memcpy(o, z.v, 48); }
// chain of n mixed additions in the 30-bit representation: acc = first; acc += pts[i] (sign bit i of `signs`)
void hm_madd30_chain(const uint32_t *pts, int n, uint64_t signs, uint32_t *o) {
    const G1Affine *p = (const G1Affine *)pts;
    G1Xyzz30 acc = g1_from_affine30(g1_affine_to30(p[0]), signs & 1);
    for (int i = 1; i < n; i++) acc = g1_madd30(acc, g1_affine_to30(p[i]), (signs >> i) & 1);
    G1Affine r = g1_to_affine(g1_xyzz_from30(acc)); memcpy(o, &r, 96); }
}
extern "C" {
// the same chain the way k_accum_affine runs it: phase 1 / phase 2 with the accumulator left in its lazy form (unsigned digits
// in X, ZZ, ZZZ) across iterations, identity / restart handled as in the kernel, normalised once at the end
void hm_madd30_chain_kernel_form(const uint32_t *pts, int n, uint64_t signs, uint32_t *o) {
    const G1Affine *p = (const G1Affine *)pts;
    G1Affine30 first = g1_affine_to30(p[0]);
    G1Xyzz30 acc = g1_from_affine30(first, signs & 1);
    for (int i = 1; i < n; i++) {
        const G1Affine30 cur = g1_affine_to30(p[i]);
        const bool neg = (signs >> i) & 1;
        if (cur.is_inf()) continue;
        if (acc.inf) { acc = g1_from_affine30(cur, neg); continue; }
        Madd30Mid mid = g1_madd30_phase1(acc, cur, neg);
        acc = g1_madd30_phase2(acc, mid, neg, [&]() { return cur; });
    }
    G1Affine r = g1_to_affine(g1_xyzz_from30(g1_normalize30(acc))); memcpy(o, &r, 96); }
// general 30-bit addition / doubling on de-normalised operands built from madd30 / dbl30 chains
void hm_add30(const uint32_t *a, const uint32_t *b, uint32_t *o) { G1Affine x, y; memcpy(&x, a, 96); memcpy(&y, b, 96);
    G1Affine30 x30 = g1_affine_to30(x), y30 = g1_affine_to30(y);
    // p = 2x - x (via dbl30 + madd30 of -x), q = (y + y) - y: non-trivial ZZ/ZZZ and lazy coordinates
    G1Xyzz30 p = g1_madd30(g1_dbl30(g1_from_affine30(x30, false)), x30, true);
    G1Xyzz30 q = g1_madd30(g1_madd30(g1_from_affine30(y30, false), y30, false), y30, true);
    G1Affine r = g1_to_affine(g1_xyzz_from30(g1_add30(p, q))); memcpy(o, &r, 96); }
// [k]P by double-and-add with dbl30 / add30 only (k: 8 x u32)
void hm_mul30_scalar(const uint32_t *a, const uint32_t *k, uint32_t *o) { G1Affine x; memcpy(&x, a, 96);
    G1Xyzz30 base = g1_from_affine30(g1_affine_to30(x), false), acc = G1Xyzz30::infinity();
    for (int i = 255; i >= 0; i--) { acc = g1_dbl30(acc); if ((k[i >> 5] >> (i & 31)) & 1) acc = g1_add30(acc, base); }
    G1Affine r = g1_to_affine(g1_xyzz_from30(acc)); memcpy(o, &r, 96); }
}
extern "C" {
void hm_fq_inv_bgcd(const uint32_t *a, uint32_t *o) { Fq x; memcpy(x.v, a, 48); Fq z = inv_bgcd(x); memcpy(o, z.v, 48); }
void hm_fr_inv_bgcd(const uint32_t *a, uint32_t *o) { Fr x; memcpy(x.v, a, 32); Fr z = inv_bgcd(x); memcpy(o, z.v, 32); }
}
#include "../kzg_amd/csrc/fr29.h"
extern "C" {
// x (any 256-bit integer < 2^256), w_mont = w*2^256 mod r  ->  canonical (x * w) mod r via the 29-bit path
void hm_fr29_mul(const uint32_t *x, const uint32_t *w_mont, uint32_t *o) { Fr a, w; memcpy(a.v, x, 32); memcpy(w.v, w_mont, 32);
    Fr z = fr29_pack_canonical(mul29r(fr29_unpack(a), fr29_twiddle_from_mont(w))); memcpy(o, z.v, 32); }
// `stages` lazy butterflies in a row on (u, v) with twiddle w, then a final multiplication by one: returns both canonical
void hm_fr29_butterflies(const uint32_t *u, const uint32_t *v, const uint32_t *w_mont, int stages, uint32_t *ou, uint32_t *ov) {
    Fr a, b, w; memcpy(a.v, u, 32); memcpy(b.v, v, 32); memcpy(w.v, w_mont, 32);
    Fr29 U = fr29_unpack(a), V = fr29_unpack(b), W = fr29_twiddle_from_mont(w);
    for (int s = 0; s < stages; s++) { Fr29 t = mul29r(V, W); fr29_butterfly(U, V, t); }
    Fr zu = fr29_pack_canonical(mul29r(U, fr29_one())), zv = fr29_pack_canonical(mul29r(V, fr29_one()));
    memcpy(ou, zu.v, 32); memcpy(ov, zv.v, 32); }
}
extern "C" {
// the Shoup product of the NTT kernels: x (9 raw limbs, possibly unnormalised: any u32 values the caller supplies) times the
// constant w (Montgomery form in), through fr29_shoup_from_twiddle -> (w, wp) -> mulshoup29; returns the 9 raw result limbs and (w, wp)
void hm_fr29_shoup_raw(const uint32_t *x_limbs, const uint32_t *w_mont, uint32_t *out_limbs, uint32_t *w_out, uint32_t *wp_out) {
    Fr w; memcpy(w.v, w_mont, 32);
    Fr29 x, W, WP; memcpy(x.v, x_limbs, 36);
    fr29_shoup_from_twiddle(fr29_twiddle_from_mont(w), W, WP);
    Fr29 r = mulshoup29(x, W, WP);
    memcpy(out_limbs, r.v, 36); memcpy(w_out, W.v, 36); memcpy(wp_out, WP.v, 36); }
// `pairs` radix-4 stage pairs of the LDS transform on the element chain that is never multiplied (role x0 of every pair) with
// the other three inputs fresh each time: exactly lds_ntt_stages29's register code -- lazy butterflies, products of unnormalised
// sums, one normalisation per pair; then the closing canonicalisation of pass 2.  x0..x3: 256-bit integers; which = which of the
// four outputs continues as the next pair's x0 (0..3: 3 = z3, the fastest-growing).  Returns the canonical chain value.
void hm_fr29_radix4_chain(const uint32_t *x0, const uint32_t *xs, const uint32_t *w_mont, int pairs, int which, uint32_t *o) {
    Fr a; memcpy(a.v, x0, 32);
    Fr29 X0 = fr29_unpack(a);
    for (int p = 0; p < pairs; p++) {
        Fr b1, b2, b3, w1, w2, w3;
        memcpy(b1.v, xs + 24 * p, 32); memcpy(b2.v, xs + 24 * p + 8, 32); memcpy(b3.v, xs + 24 * p + 16, 32);
        memcpy(w1.v, w_mont + 24 * p, 32); memcpy(w2.v, w_mont + 24 * p + 8, 32); memcpy(w3.v, w_mont + 24 * p + 16, 32);
        Fr29 A, AP, B, BP, Cw, CP;
        fr29_shoup_from_twiddle(fr29_twiddle_from_mont(w1), A, AP);
        fr29_shoup_from_twiddle(fr29_twiddle_from_mont(w2), B, BP);
        fr29_shoup_from_twiddle(fr29_twiddle_from_mont(w3), Cw, CP);
        Fr29 x1 = fr29_unpack(b1), x2 = fr29_unpack(b2), x3 = fr29_unpack(b3);
        Fr29 t1 = mulshoup29(x1, A, AP), t3 = mulshoup29(x3, A, AP);
        Fr29 s0, y1, s2, y3;
        fr29_butterfly_lazy(X0, t1, s0, y1);
        fr29_butterfly_lazy(x2, t3, s2, y3);
        Fr29 t2 = mulshoup29(s2, B, BP), t3b = mulshoup29(y3, Cw, CP);
        Fr29 z0, z2, z1, z3;
        fr29_butterfly_lazy(s0, t2, z0, z2);
        fr29_butterfly_lazy(y1, t3b, z1, z3);
        X0 = fr29_normalize(which == 0 ? z0 : which == 1 ? z1 : which == 2 ? z2 : z3);
    }
    Fr z = fr29_pack_canonical(fr29_reduce_below_2r(X0)); memcpy(o, z.v, 32); }
// The arithmetic of one thread of the quotient kernels (poly.hip: k_horner_partials / k_horner_scan / k_quotient_apply), same
// primitives in the same order: Horner of eight raw 256-bit coefficients by Shoup products with lazy sums, `m` scan steps each adding
// the product of a neighbour's value (nine raw limbs as read from LDS: normalised, up to 25 r) with a step constant, the canonical
// value written out, then one output step of k_quotient_apply: coefficient + canonical(product) by the saturated modular addition.
// a: 8 x 8 words; x_mont, p_mont: Montgomery form; nb: m x 9 limbs.  o_scan: canonical value after the scan; o_next: the next output.
void hm_fr29_quotient_thread(const uint32_t *a, const uint32_t *x_mont, const uint32_t *p_mont, const uint32_t *nb, int m, const uint32_t *a_next,
                             uint32_t *o_scan, uint32_t *o_next, uint32_t *top_limb) {
    Fr xm, pm; memcpy(xm.v, x_mont, 32); memcpy(pm.v, p_mont, 32);
    Fr29 X, XP, P, PP;
    fr29_shoup_from_twiddle(fr29_twiddle_from_mont(xm), X, XP);
    fr29_shoup_from_twiddle(fr29_twiddle_from_mont(pm), P, PP);
    Fr c[8]; memcpy(c, a, 256);
    Fr29 v = fr29_unpack(c[7]);
    for (int k = 6; k >= 0; k--) v = fr29_add_lazy(mulshoup29(v, X, XP), fr29_unpack(c[k]));
    v = fr29_normalize(v);
    for (int i = 0; i < m; i++) {
        Fr29 o; memcpy(o.v, nb + 9 * i, 36);
        v = fr29_normalize(fr29_add_lazy(v, mulshoup29(o, P, PP)));
    }
    *top_limb = v.v[8];
    Fr out = fr29_canonical(v); memcpy(o_scan, out.v, 32);
    Fr an; memcpy(an.v, a_next, 32);
    Fr nx = add(an, fr29_pack_canonical(mulshoup29(fr29_unpack(out), X, XP))); memcpy(o_next, nx.v, 32); }
}
#include "../kzg_amd/csrc/emit.h"
extern "C" {
// emit.h on the host (what capi.hip runs for a lone host-bound MSM result): k * P as a de-normalised XYZZ point in the signed
// 30-bit form -> `fmt`; k == 0 gives the identity
void hm_emit(const uint32_t *a, const uint32_t *k, int fmt, uint8_t *o) { G1Affine x; memcpy(&x, a, 96);
    G1Xyzz p = g1_scalar_mul(x, k);
    G1Xyzz30 q = g1_xyzz_to30(p);
    alignas(16) uint8_t buf[144];
    emit_one(q, buf, fmt);
    memcpy(o, buf, fmt == KZG_G1_JACOBIAN_MONT_144 ? 144 : fmt == KZG_G1_ZCASH_COMPRESSED_48 ? 48 : 96); }
}
