// fr29.h -- Fr in the UNSATURATED 9 x 29-bit representation (Montgomery radix 2^261) used inside the NTT
// kernels.  Same idea as field30.h (there with signed limbs): a product-scanning multiply whose columns (<= 18 products < 2^58) sum
// in one 64-bit accumulator, so every partial product is a bare v_mad_u64_u32 (162 per multiply instead of
// 128 mads + 128 carry folds), and lazily reduced butterflies: u + t and u - t + 2r are limb-wise adds.
//
// The NTT is linear, so data needs NO conversion multiply: the 256-bit integer found in memory (canonical or
// blst_fr Montgomery form, either way some residue) is just unpacked into 29-bit limbs; the twiddles are
// stored as w * 2^261 mod r, and mul29r(x, w*2^261) = x*w keeps whatever form x was in.
// Bounds: a twiddle product is < 1.4 r for any x below 26 r (2^261 = 70 r), values grow by at most 2 r per
// butterfly stage, 12 stages at most -> everything stays below 26 r < 2^261.  The last multiplication of a pass
// (inter-pass twiddle, d^-1 scale, or one) brings the value below 1.4 r and one conditional subtraction
// makes it canonical before it is packed back into 8 x 32-bit words.
#pragma once
#include "field.h"

namespace kzg {

#include "fr29_consts.inc"

constexpr uint32_t F29_MASK = (1u << 29) - 1u;

constexpr int R29_N = 9;

struct Fr29 {
    uint32_t v[R29_N];
};

KZG_HD Fr29 fr29_normalize(Fr29 a) {
#pragma unroll
    for (int i = 0; i < R29_N - 1; i++) {
        a.v[i + 1] += a.v[i] >> 29;
        a.v[i] &= F29_MASK;
    }
    return a;
}

// a*b/2^261 mod r (lazy).  Limbs of b < 2^29 (twiddle), limbs of a < 2^30.
KZG_HD Fr29 mul29r_inline(const Fr29 &a, const Fr29 &b) {
    uint32_t m[R29_N];
    Fr29 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < R29_N; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * Fr29Consts::mod(k - i);
        m[k] = ((uint32_t)acc * Fr29Consts::INV) & F29_MASK;
        acc += (uint64_t)m[k] * Fr29Consts::mod(0);
        acc >>= 29;
    }
#pragma unroll
    for (int k = R29_N; k < 2 * R29_N - 1; k++) {
#pragma unroll
        for (int i = k - R29_N + 1; i < R29_N; i++) {
            acc += (uint64_t)a.v[i] * b.v[k - i];
            acc += (uint64_t)m[i] * Fr29Consts::mod(k - i);
        }
        r.v[k - R29_N] = (uint32_t)acc & F29_MASK;
        acc >>= 29;
    }
    r.v[R29_N - 1] = (uint32_t)acc;
    return r;
}

#if defined(__HIP_DEVICE_COMPILE__) && defined(KZG_OOL_MUL29R)
// optional: one out-of-line body (smaller code; measured 2-10 % slower than inlining)
typedef uint32_t u32x9 __attribute__((ext_vector_type(9)));
__device__ __noinline__ u32x9 mul29r_ool(u32x9 a, u32x9 b) {
    Fr29 x, y;
#pragma unroll
    for (int i = 0; i < R29_N; i++) {
        x.v[i] = a[i];
        y.v[i] = b[i];
    }
    Fr29 z = mul29r_inline(x, y);
    u32x9 r;
#pragma unroll
    for (int i = 0; i < R29_N; i++) r[i] = z.v[i];
    return r;
}
KZG_HD Fr29 mul29r(const Fr29 &a, const Fr29 &b) {
    u32x9 x, y;
#pragma unroll
    for (int i = 0; i < R29_N; i++) {
        x[i] = a.v[i];
        y[i] = b.v[i];
    }
    u32x9 z = mul29r_ool(x, y);
    Fr29 r;
#pragma unroll
    for (int i = 0; i < R29_N; i++) r.v[i] = z[i];
    return r;
}
#elif defined(__HIP_DEVICE_COMPILE__) && !defined(KZG_C_MUL29R)
// default on the device: generated single-chain version (tools/gen_mul30.py, see field30.h): 2^20 NTT 0.145 -> 0.138 ms
#include "mul29r_gfx950.inc"
KZG_HD Fr29 mul29r(const Fr29 &a, const Fr29 &b) { return mul29r_asm(a, b); }
#else
KZG_HD Fr29 mul29r(const Fr29 &a, const Fr29 &b) { return mul29r_inline(a, b); }
#endif

// ---- multiplication by a CONSTANT (Shoup / Harvey): the NTT's twiddles are all constants ------------------------------------
// A twiddle is stored as the pair (w, wp): w < r in plain form (no Montgomery factor) and wp = floor(w 2^261 / r), nine 29-bit
// limbs each.  x * w mod r for any x < 2^261 with limbs below 1.5 * 2^30 (unnormalised sums of a butterfly are fine):
//   q~ = floor(high part of x * wp / 2^261): columns 7..16 of the product scan only -- the dropped columns 0..6 are worth less than
//        2^-24 of a unit of q, so q~ is floor(x w / r) or one less (x w'/2^261 > x w / r - x / 2^261);
//   result = low 261 bits of x * w - q~ * r, computed in ONE signed column chain: the true value lies in [0, (1 + x / 2^261) r)
//        < 2r, far below 2^261, so the low nine limbs ARE the value.
// 53 + 45 + 45 = 143 multiply-adds, 19 shifts and 18 masks; the Montgomery product above needs 154 + 9 adds by r_0 = 1, 17 shifts,
// 18 masks and nine quotient digits (negate + mask).  Column bounds: 9 * (1.5 * 2^30) * 2^29 = 2^62.75 (unsigned and signed chain).
// The data keeps whatever form it is in (canonical or blst_fr Montgomery): x * w is the plain product.
KZG_HD Fr29 mulshoup29_inline(const Fr29 &x, const Fr29 &w, const Fr29 &wp) {
    uint32_t q[R29_N];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 7; k < 2 * R29_N - 1; k++) {
#pragma unroll
        for (int i = 0; i < R29_N; i++)
            if (k - i >= 0 && k - i < R29_N) acc += (uint64_t)x.v[i] * wp.v[k - i];
        if (k >= R29_N) q[k - R29_N] = (uint32_t)acc & F29_MASK;
        acc >>= 29;
    }
    q[R29_N - 1] = (uint32_t)acc;
    Fr29 r;
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < R29_N; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) {
            s += (int64_t)((uint64_t)x.v[i] * w.v[k - i]);
            s -= (int64_t)((uint64_t)q[i] * Fr29Consts::mod(k - i));
        }
        r.v[k] = (uint32_t)s & F29_MASK;
        s >>= 29;
    }
    return r;
}

#if defined(__HIP_DEVICE_COMPILE__) && !defined(KZG_C_MUL29R) && !defined(KZG_OOL_MUL29R)
KZG_HD Fr29 mulshoup29(const Fr29 &x, const Fr29 &w, const Fr29 &wp) { return mulshoup29_asm(x, w, wp); }
// two independent products as one instruction stream, the two accumulator chains interleaved multiply-add by multiply-add (tools/gen_mul30.py)
KZG_HD void mulshoup29x2(Fr29 &x, const Fr29 &w, const Fr29 &wp, Fr29 &y, const Fr29 &w2, const Fr29 &wp2) {
#ifdef KZG_NTT_SINGLE_CHAIN
    x = mulshoup29_asm(x, w, wp);
    y = mulshoup29_asm(y, w2, wp2);
#else
    Fr29 rx, ry;
    mulshoup29x2_asm(x, w, wp, y, w2, wp2, rx, ry);
    x = rx;
    y = ry;
#endif
}
#else
KZG_HD Fr29 mulshoup29(const Fr29 &x, const Fr29 &w, const Fr29 &wp) { return mulshoup29_inline(x, w, wp); }
KZG_HD void mulshoup29x2(Fr29 &x, const Fr29 &w, const Fr29 &wp, Fr29 &y, const Fr29 &w2, const Fr29 &wp2) {
    x = mulshoup29_inline(x, w, wp);
    y = mulshoup29_inline(y, w2, wp2);
}
#endif

// low 261 bits of a * b (table construction only)
KZG_HD Fr29 fr29_lowmul(const Fr29 &a, const Fr29 &b) {
    Fr29 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < R29_N; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
        r.v[k] = (uint32_t)acc & F29_MASK;
        acc >>= 29;
    }
    return r;
}

// radix-2 butterfly on a Shoup product t (< 2r, limbs < 2^29): s = u + t, d = u - t + 4r.  NOT normalised: limbs grow by at most
// 2^29 (s) / 2^30 (d), values by 2r / 4r; a value that is multiplied next is reduced by the product, the others are normalised
// when they are stored (fr29_normalize), once per radix-4 stage pair instead of after every butterfly.
KZG_HD void fr29_butterfly_lazy(const Fr29 &u, const Fr29 &t, Fr29 &s_out, Fr29 &d_out) {
#pragma unroll
    for (int i = 0; i < R29_N; i++) {
        const uint32_t ui = u.v[i], ti = t.v[i];
        s_out.v[i] = ui + ti;
        d_out.v[i] = ui + Fr29Consts::sub4(i) - ti;
    }
}
// the same with 8r: t is an unmultiplied, NORMALISED sum of two raw 256-bit inputs (below 4.5 r): the first stage pair of a tile,
// whose stage-one twiddle is 1
KZG_HD void fr29_butterfly_lazy8(const Fr29 &u, const Fr29 &t, Fr29 &s_out, Fr29 &d_out) {
#pragma unroll
    for (int i = 0; i < R29_N; i++) {
        const uint32_t ui = u.v[i], ti = t.v[i];
        s_out.v[i] = ui + ti;
        d_out.v[i] = ui + Fr29Consts::sub8(i) - ti;
    }
}

// butterfly: (u, t) -> (u + t, u - t + 2r), t a twiddle product (normalised, < 1.4 r); outputs normalised
KZG_HD void fr29_butterfly(Fr29 &u, Fr29 &v_out, const Fr29 &t) {
    Fr29 s, d;
#pragma unroll
    for (int i = 0; i < R29_N; i++) {
        s.v[i] = u.v[i] + t.v[i];
        d.v[i] = u.v[i] + Fr29Consts::sub2(i) - t.v[i];
    }
    u = fr29_normalize(s);
    v_out = fr29_normalize(d);
}

// 8 x 32-bit words (any 256-bit integer) <-> 9 x 29-bit limbs
KZG_HD Fr29 fr29_unpack(const Fr &a) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < R29_N; i++) {
        int bit = 29 * i, w = bit >> 5, sh = bit & 31;
        uint32_t lo = a.v[w] >> sh;
        uint32_t hi = (sh > 3 && w + 1 < 8) ? (a.v[w + 1] << (32 - sh)) : 0u;
        r.v[i] = (lo | hi) & F29_MASK;
    }
    return r;
}

// x < 2r, normalised  ->  canonical residue packed into 8 x 32-bit words
KZG_HD Fr fr29_pack_canonical(const Fr29 &x) {
    Fr r = Fr::zero();
#pragma unroll
    for (int i = 0; i < R29_N; i++) {
        int bit = 29 * i, w = bit >> 5, sh = bit & 31;
        if (w < 8) r.v[w] |= x.v[i] << sh;
        if (sh > 3 && w + 1 < 8) r.v[w + 1] |= x.v[i] >> (32 - sh);
    }
    // the value is < 2r < 2^256, so it fits; one conditional subtraction of r
    Fr t = r;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t dd = (uint64_t)t.v[i] - FrParams::mod(i) - borrow;
        t.v[i] = (uint32_t)dd;
        borrow = (uint32_t)(dd >> 63);
    }
    return borrow ? r : t;
}

// x < 2^256, normalised -> 8 x 32-bit words, as it is (the scratch between the two passes: any representative will do)
KZG_HD Fr fr29_pack_raw(const Fr29 &x) {
    Fr r = Fr::zero();
#pragma unroll
    for (int i = 0; i < R29_N; i++) {
        int bit = 29 * i, w = bit >> 5, sh = bit & 31;
        if (w < 8) r.v[w] |= x.v[i] << sh;
        if (sh > 3 && w + 1 < 8) r.v[w + 1] |= x.v[i] >> (32 - sh);
    }
    return r;
}

// x < 64 r, normalised (limbs below 2^29, the excess in the top limb)  ->  the same residue below 2r.
// q = floor(top / (r_top + 1)) with top = x >> 232 never exceeds floor(x / r) and falls short of it by at most 1
// ((top + r_top + 1) / (r_top (r_top + 1)) < 9e-6 for top <= 64 r_top), so x - q r lies in [0, 2r): ~50 instructions instead of a
// multiplication by one.
KZG_HD Fr29 fr29_reduce_below_2r(const Fr29 &x) {
    constexpr uint32_t R_TOP = 0x73eda7u;  // r >> 232
    const uint32_t q = x.v[R29_N - 1] / (R_TOP + 1);
    Fr29 r;
    int64_t carry = 0;
#pragma unroll
    for (int i = 0; i < R29_N; i++) {
        int64_t t = (int64_t)x.v[i] - (int64_t)((uint64_t)q * Fr29Consts::mod(i)) + carry;
        if (i < R29_N - 1) {
            r.v[i] = (uint32_t)t & F29_MASK;
            carry = t >> 29;
        } else {
            r.v[i] = (uint32_t)t;
        }
    }
    return r;
}

// limb-wise sum, not normalised (the caller keeps the limbs below the bound of the next operation)
KZG_HD Fr29 fr29_add_lazy(const Fr29 &a, const Fr29 &b) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < R29_N; i++) r.v[i] = a.v[i] + b.v[i];
    return r;
}
KZG_HD Fr29 fr29_zero() {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < R29_N; i++) r.v[i] = 0u;
    return r;
}
// any normalised value below 64 r -> the canonical residue in 8 x 32-bit words
KZG_HD Fr fr29_canonical(const Fr29 &v) { return fr29_pack_canonical(fr29_reduce_below_2r(v)); }

// (w, wp) of a twiddle from its Montgomery-29 form t = w 2^261 mod r (canonical limbs: what fr29_twiddle_from_mont returns):
// w = t / 2^261 mod r (a Montgomery product with the integer 1), and since w 2^261 = wp r + t, wp = (-t) r^-1 mod 2^261.
KZG_HD void fr29_shoup_from_twiddle(const Fr29 &t, Fr29 &w, Fr29 &wp) {
    Fr29 one;
#pragma unroll
    for (int i = 0; i < R29_N; i++) one.v[i] = i == 0 ? 1u : 0u;
    w = fr29_unpack(fr29_pack_canonical(mul29r(t, one)));
    Fr29 nr;
#pragma unroll
    for (int i = 0; i < R29_N; i++) nr.v[i] = Fr29Consts::neg_rinv(i);
    wp = fr29_lowmul(t, nr);
}

KZG_HD Fr29 fr29_one() {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < R29_N; i++) r.v[i] = Fr29Consts::one(i);
    return r;
}

// twiddle conversion: w * 2^256 (saturated Montgomery form, canonical) -> w * 2^261 in 29-bit limbs (< 1.1 r; then
// reduced below r so that it can serve as the small operand everywhere)
KZG_HD Fr29 fr29_twiddle_from_mont(const Fr &w_mont) {
    Fr29 k;
#pragma unroll
    for (int i = 0; i < R29_N; i++) k.v[i] = Fr29Consts::k_to29(i);
    Fr29 t = mul29r(fr29_unpack(w_mont), k);
    return fr29_unpack(fr29_pack_canonical(t));
}

}  // namespace kzg
