// field30.h -- Fq in an UNSATURATED, SIGNED representation: 13 balanced limbs of 30 bits, Montgomery radix R30 = 2^390.
//
// Why: with saturated 32-bit limbs every partial product of a multi-precision multiply needs a carry fold
// (v_mad_u64_u32 + v_addc_co_u32; 288 of each per Fq multiply).  With limbs in [-2^29, 2^29) a whole column of a
// 13 x 13 product-scanning Montgomery multiply (<= 26 products of magnitude <= 2^58) sums in ONE signed 64-bit
// accumulator with no carry handling at all: acc = a*b + acc is a bare v_mad_i64_i32.  338 mads + ~110 other
// instructions.  An unsigned 13 x 30-bit form would overflow the accumulator (26 x 2^60), which is why the first
// version of this layer used 14 x 29-bit limbs (392 mads); balanced digits buy the 14th limb back.
//
// Signed values also make the lazy-reduction bookkeeping trivial: a - b is a limb-wise subtraction (values may be
// negative, no "+ c*q" bias constants), and a Montgomery product r = (a*b + m*q)/R30 with the balanced quotient
// |m| <= R30/2 satisfies |r| <= |a*b|/R30 + q/2: for |a|*|b| < 300 q^2 that is |r| < q (q/R30 = 2^-9.3), so a
// product is zero mod q iff all its limbs are zero.  Everything in curve30.h stays below 8q in magnitude.
//
// "Normalised" = limbs 0..11 in [-2^29, 2^29), limb 12 holds the (small, signed) rest; this form is unique.
// Multiplication operands must have |limb| <= 2^29 (normalised values and their limb-wise negations qualify).
// Used only inside the MSM kernels and the resident SRS table; canonical encodings are produced by from30().
#pragma once
#include "field.h"

namespace kzg {

#include "fq30_consts.inc"

constexpr int F30_N = 13;
constexpr int F30_B = 30;
constexpr int32_t F30_HALF = 1 << (F30_B - 1);
constexpr uint32_t F30_MASK = (1u << F30_B) - 1u;

struct Fq30 {
    int32_t v[F30_N];
    KZG_HD bool limbs_all_zero() const {
        int32_t t = 0;
#pragma unroll
        for (int i = 0; i < F30_N; i++) t |= v[i];
        return t == 0;
    }
};

// the low 30 bits of x as a balanced digit in [-2^29, 2^29)   (v_bfe_i32)
KZG_HD int32_t sext30(uint32_t x) { return (int32_t)(x << 2) >> 2; }

// carry-propagate into the unique normalised form.  Requires |limb| < 2^31 - 2^29 on entry.
KZG_HD Fq30 normalize30(Fq30 a) {
#pragma unroll
    for (int i = 0; i < F30_N - 1; i++) {
        int32_t c = (a.v[i] + F30_HALF) >> F30_B;
        a.v[i] = sext30((uint32_t)a.v[i]);
        a.v[i + 1] += c;
    }
    return a;
}

KZG_HD uint64_t mac30(uint64_t acc, int32_t a, int32_t b) { return acc + (uint64_t)((int64_t)a * (int64_t)b); }
KZG_HD uint64_t sar30(uint64_t acc) { return (uint64_t)((int64_t)acc >> F30_B); }

// Montgomery product a*b/R30 mod q, balanced: |result| <= |a*b|/R30 + q/2.  Operand limbs |.| <= 2^29.
// Result normalised.
// UNSIGNED_OUT: the output digits 0..11 come out in [0, 2^30) (floor carries) instead of balanced -- one instruction less per
// digit (no rounding add).  Such a value may be ONE operand of a later product whose other operand is balanced: a column then
// holds at most 12 products below 2^59 plus the m*q part, 2^29 * sum |q_j| = 2^60.6 for this q: 2^62.93 in all (host-checked at
// the extremes, tests/test_host_math.py); two unsigned operands would overflow.  Subtraction operands may be either.
template <bool UNSIGNED_OUT = false>
KZG_HD Fq30 mul30_inline(const Fq30 &a, const Fq30 &b) {
    int32_t m[F30_N];
    Fq30 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < F30_N; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc = mac30(acc, a.v[i], b.v[k - i]);
#pragma unroll
        for (int i = 0; i < k; i++) acc = mac30(acc, m[i], Fq30Consts::mod(k - i));
        m[k] = sext30((uint32_t)acc * Fq30Consts::INV);
        acc = mac30(acc, m[k], Fq30Consts::mod(0));
        acc = sar30(acc);  // exact: the low 30 bits are zero
    }
#pragma unroll
    for (int k = F30_N; k < 2 * F30_N - 1; k++) {
#pragma unroll
        for (int i = k - F30_N + 1; i < F30_N; i++) {
            acc = mac30(acc, a.v[i], b.v[k - i]);
            acc = mac30(acc, m[i], Fq30Consts::mod(k - i));
        }
        if (UNSIGNED_OUT) {
            r.v[k - F30_N] = (int32_t)((uint32_t)acc & F30_MASK);
            acc = sar30(acc);
        } else {
            r.v[k - F30_N] = sext30((uint32_t)acc);
            acc = sar30(acc + (uint64_t)F30_HALF);
        }
    }
    r.v[F30_N - 1] = (int32_t)acc;
    return r;
}

// a*b/R30 - c with the subtraction MERGED into the product: c's limb j enters output column 13 + j of the scan (one
// multiply-add by the constant -1), so the difference comes out of the digit extraction already normalised -- no limb-wise
// subtraction, no separate carry pass (13 multiply-adds instead of ~61 instructions).  |result| <= |a*b|/R30 + q/2 + |c|.
// c: any limbs below 2^31 in magnitude.
KZG_HD Fq30 mul30_sub_inline(const Fq30 &a, const Fq30 &b, const Fq30 &c) {
    int32_t m[F30_N];
    Fq30 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < F30_N; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc = mac30(acc, a.v[i], b.v[k - i]);
#pragma unroll
        for (int i = 0; i < k; i++) acc = mac30(acc, m[i], Fq30Consts::mod(k - i));
        m[k] = sext30((uint32_t)acc * Fq30Consts::INV);
        acc = mac30(acc, m[k], Fq30Consts::mod(0));
        acc = sar30(acc);
    }
#pragma unroll
    for (int k = F30_N; k < 2 * F30_N - 1; k++) {
#pragma unroll
        for (int i = k - F30_N + 1; i < F30_N; i++) {
            acc = mac30(acc, a.v[i], b.v[k - i]);
            acc = mac30(acc, m[i], Fq30Consts::mod(k - i));
        }
        acc = mac30(acc, c.v[k - F30_N], -1);
        r.v[k - F30_N] = sext30((uint32_t)acc);
        acc = sar30(acc + (uint64_t)F30_HALF);
    }
    acc = mac30(acc, c.v[F30_N - 1], -1);
    r.v[F30_N - 1] = (int32_t)acc;
    return r;
}

// (a*b + c*d)/R30 with ONE reduction.  A column of the fused scan holds up to 39 products of magnitude 2^58, which
// would overflow the signed accumulator; in the five columns with more than 30 products the multiple of 2^30
// accumulated so far is set aside before the c*d products go in and rejoins the carry afterwards.
// |result| <= (|a*b| + |c*d|)/R30 + q/2.  Inlined at its (single) call site.
KZG_HD Fq30 muladd30_inline(const Fq30 &a, const Fq30 &b, const Fq30 &c, const Fq30 &d) {
    int32_t m[F30_N];
    Fq30 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < F30_N; k++) {
        const bool split = 3 * (k + 1) > 30;
        uint64_t hi = 0;
#pragma unroll
        for (int i = 0; i <= k; i++) acc = mac30(acc, a.v[i], b.v[k - i]);
#pragma unroll
        for (int i = 0; i < k; i++) acc = mac30(acc, m[i], Fq30Consts::mod(k - i));
        if (split) {
            hi = sar30(acc);
            acc &= (uint64_t)F30_MASK;
        }
#pragma unroll
        for (int i = 0; i <= k; i++) acc = mac30(acc, c.v[i], d.v[k - i]);
        m[k] = sext30((uint32_t)acc * Fq30Consts::INV);
        acc = mac30(acc, m[k], Fq30Consts::mod(0));
        acc = sar30(acc) + hi;
    }
#pragma unroll
    for (int k = F30_N; k < 2 * F30_N - 1; k++) {
        const bool split = 3 * (2 * F30_N - 1 - k) > 30;
        uint64_t hi = 0;
#pragma unroll
        for (int i = k - F30_N + 1; i < F30_N; i++) {
            acc = mac30(acc, a.v[i], b.v[k - i]);
            acc = mac30(acc, m[i], Fq30Consts::mod(k - i));
        }
        if (split) {
            hi = sar30(acc);
            acc &= (uint64_t)F30_MASK;
        }
#pragma unroll
        for (int i = k - F30_N + 1; i < F30_N; i++) acc = mac30(acc, c.v[i], d.v[k - i]);
        r.v[k - F30_N] = sext30((uint32_t)acc);
        acc = sar30(acc + (uint64_t)F30_HALF) + hi;
    }
    r.v[F30_N - 1] = (int32_t)acc;
    return r;
}

// Montgomery square: the 78 cross products are formed once against a doubled copy of a (|limb| <= 2^30; a column
// holds at most 6 of them plus one square: 13 x 2^58 again), 91 + 169 mads instead of 338.
KZG_HD Fq30 sqr30_inline(const Fq30 &a) {
    int32_t m[F30_N], d[F30_N];
    Fq30 r;
#pragma unroll
    for (int i = 0; i < F30_N; i++) d[i] = a.v[i] * 2;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < F30_N; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) acc = mac30(acc, a.v[i], d[k - i]);
        if ((k & 1) == 0) acc = mac30(acc, a.v[k / 2], a.v[k / 2]);
#pragma unroll
        for (int i = 0; i < k; i++) acc = mac30(acc, m[i], Fq30Consts::mod(k - i));
        m[k] = sext30((uint32_t)acc * Fq30Consts::INV);
        acc = mac30(acc, m[k], Fq30Consts::mod(0));
        acc = sar30(acc);
    }
#pragma unroll
    for (int k = F30_N; k < 2 * F30_N - 1; k++) {
#pragma unroll
        for (int i = k - F30_N + 1; 2 * i < k; i++) acc = mac30(acc, a.v[i], d[k - i]);
        if ((k & 1) == 0) acc = mac30(acc, a.v[k / 2], a.v[k / 2]);
#pragma unroll
        for (int i = k - F30_N + 1; i < F30_N; i++) acc = mac30(acc, m[i], Fq30Consts::mod(k - i));
        r.v[k - F30_N] = sext30((uint32_t)acc);
        acc = sar30(acc + (uint64_t)F30_HALF);
    }
    r.v[F30_N - 1] = (int32_t)acc;
    return r;
}

// a^2/R30 - c - 2e, both subtrahends merged into the square's output columns (X3 = R^2 - PPP - 2Q of the mixed addition).
template <bool UNSIGNED_OUT = false>
KZG_HD Fq30 sqr30_sub2_inline(const Fq30 &a, const Fq30 &c, const Fq30 &e) {
    int32_t m[F30_N], d[F30_N];
    Fq30 r;
#pragma unroll
    for (int i = 0; i < F30_N; i++) d[i] = a.v[i] * 2;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < F30_N; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) acc = mac30(acc, a.v[i], d[k - i]);
        if ((k & 1) == 0) acc = mac30(acc, a.v[k / 2], a.v[k / 2]);
#pragma unroll
        for (int i = 0; i < k; i++) acc = mac30(acc, m[i], Fq30Consts::mod(k - i));
        m[k] = sext30((uint32_t)acc * Fq30Consts::INV);
        acc = mac30(acc, m[k], Fq30Consts::mod(0));
        acc = sar30(acc);
    }
#pragma unroll
    for (int k = F30_N; k < 2 * F30_N - 1; k++) {
#pragma unroll
        for (int i = k - F30_N + 1; 2 * i < k; i++) acc = mac30(acc, a.v[i], d[k - i]);
        if ((k & 1) == 0) acc = mac30(acc, a.v[k / 2], a.v[k / 2]);
#pragma unroll
        for (int i = k - F30_N + 1; i < F30_N; i++) acc = mac30(acc, m[i], Fq30Consts::mod(k - i));
        acc = mac30(acc, c.v[k - F30_N], -1);
        acc = mac30(acc, e.v[k - F30_N], -2);
        if (UNSIGNED_OUT) {
            r.v[k - F30_N] = (int32_t)((uint32_t)acc & F30_MASK);
            acc = sar30(acc);
        } else {
            r.v[k - F30_N] = sext30((uint32_t)acc);
            acc = sar30(acc + (uint64_t)F30_HALF);
        }
    }
    acc = mac30(acc, c.v[F30_N - 1], -1);
    acc = mac30(acc, e.v[F30_N - 1], -2);
    r.v[F30_N - 1] = (int32_t)acc;
    return r;
}

#if defined(__HIP_DEVICE_COMPILE__) && defined(KZG_OOL_MUL30)
// Optional (-DKZG_OOL_MUL30): ONE out-of-line body per operation (operands travel in VGPR tuples; struct arguments
// would go through scratch): a mixed addition is ~25 KB of code instead of ~80 KB.  Measured slower than inlining
// on gfx950 (k_accum_affine 2.76 vs 2.53 ms, same-box A/B: no argument shuffling, scheduling across the call
// boundaries, 215 instead of 229 VGPRs), so the default build inlines every multiply.
typedef int32_t i32x13 __attribute__((ext_vector_type(13)));
__device__ __noinline__ i32x13 sqr30_ool(i32x13 a) {
    Fq30 x;
#pragma unroll
    for (int i = 0; i < F30_N; i++) x.v[i] = a[i];
    Fq30 z = sqr30_inline(x);
    i32x13 r;
#pragma unroll
    for (int i = 0; i < F30_N; i++) r[i] = z.v[i];
    return r;
}
KZG_HD Fq30 sqr30(const Fq30 &a) {
    i32x13 x;
#pragma unroll
    for (int i = 0; i < F30_N; i++) x[i] = a.v[i];
    i32x13 z = sqr30_ool(x);
    Fq30 r;
#pragma unroll
    for (int i = 0; i < F30_N; i++) r.v[i] = z[i];
    return r;
}
__device__ __noinline__ i32x13 mul30_ool(i32x13 a, i32x13 b) {
    Fq30 x, y;
#pragma unroll
    for (int i = 0; i < F30_N; i++) {
        x.v[i] = a[i];
        y.v[i] = b[i];
    }
    Fq30 z = mul30_inline<false>(x, y);
    i32x13 r;
#pragma unroll
    for (int i = 0; i < F30_N; i++) r[i] = z.v[i];
    return r;
}
KZG_HD Fq30 mul30(const Fq30 &a, const Fq30 &b) {
    i32x13 x, y;
#pragma unroll
    for (int i = 0; i < F30_N; i++) {
        x[i] = a.v[i];
        y[i] = b.v[i];
    }
    i32x13 z = mul30_ool(x, y);
    Fq30 r;
#pragma unroll
    for (int i = 0; i < F30_N; i++) r.v[i] = z[i];
    return r;
}
#elif defined(__HIP_DEVICE_COMPILE__) && !defined(KZG_C_MUL30)
// Default on the device: generated single-chain versions (tools/gen_mul30.py).  hipcc's code for the portable functions above
// sums every column in a second accumulator and merges it with the carry by a 64-bit add (36 v_lshl_add_u64 per multiply);
// with the products in inline-asm blocks the chain starts from the carry.  Same results bit for bit; measured +4 % batched
// MSM throughput (same-box A/B).  -DKZG_C_MUL30 selects the portable versions.
#include "mul30_gfx950.inc"
KZG_HD Fq30 mul30(const Fq30 &a, const Fq30 &b) { return mul30_asm(a, b); }
KZG_HD Fq30 sqr30(const Fq30 &a) { return sqr30_asm(a); }
#define KZG_HAVE_MULADD30_ASM 1
#else
KZG_HD Fq30 mul30(const Fq30 &a, const Fq30 &b) { return mul30_inline<false>(a, b); }
KZG_HD Fq30 sqr30(const Fq30 &a) { return sqr30_inline(a); }
#endif

// merged-subtraction forms: generated single-chain versions on the device (default build), the portable ones elsewhere.
// -DKZG_NO_MERGED_SUB selects product + sub30 (the previous formulation; A/B builds).
KZG_HD Fq30 sub30(const Fq30 &a, const Fq30 &b);
KZG_HD Fq30 add2x30(const Fq30 &a, const Fq30 &b);
KZG_HD Fq30 mul30_sub(const Fq30 &a, const Fq30 &b, const Fq30 &c) {
#if defined(KZG_NO_MERGED_SUB)
    return sub30(mul30(a, b), c);
#elif defined(KZG_HAVE_MULADD30_ASM)
    return mul30_sub_asm(a, b, c);
#else
    return mul30_sub_inline(a, b, c);
#endif
}
KZG_HD Fq30 sqr30_sub2(const Fq30 &a, const Fq30 &c, const Fq30 &e) {
#if defined(KZG_NO_MERGED_SUB)
    return sub30(sqr30(a), add2x30(c, e));
#elif defined(KZG_HAVE_MULADD30_ASM)
    return sqr30_sub2_asm(a, c, e);
#else
    return sqr30_sub2_inline<false>(a, c, e);
#endif
}

// unsigned-digit outputs (see mul30_inline): for products whose result meets only balanced partners afterwards.
// -DKZG_NO_UNSIGNED_DIGITS: the balanced versions (A/B builds).
KZG_HD Fq30 mul30u(const Fq30 &a, const Fq30 &b) {
#if defined(KZG_NO_UNSIGNED_DIGITS)
    return mul30(a, b);
#elif defined(KZG_HAVE_MULADD30_ASM)
    return mul30u_asm(a, b);
#else
    return mul30_inline<true>(a, b);
#endif
}
KZG_HD Fq30 sqr30_sub2u(const Fq30 &a, const Fq30 &c, const Fq30 &e) {
#if defined(KZG_NO_UNSIGNED_DIGITS) || defined(KZG_NO_MERGED_SUB)
    return sqr30_sub2(a, c, e);
#elif defined(KZG_HAVE_MULADD30_ASM)
    return sqr30_sub2u_asm(a, c, e);
#else
    return sqr30_sub2_inline<true>(a, c, e);
#endif
}

KZG_HD Fq30 muladd30(const Fq30 &a, const Fq30 &b, const Fq30 &c, const Fq30 &d) {
#if defined(KZG_HAVE_MULADD30_ASM)
    return muladd30_asm(a, b, c, d);
#else
    return muladd30_inline(a, b, c, d);
#endif
}

KZG_HD Fq30 zero30() {
    Fq30 z;
#pragma unroll
    for (int i = 0; i < F30_N; i++) z.v[i] = 0;
    return z;
}

// -a, limb-wise: |limb| <= 2^29 is kept (fit for a multiplication), the form is not the normalised one
KZG_HD Fq30 neg30(const Fq30 &a) {
    Fq30 r;
#pragma unroll
    for (int i = 0; i < F30_N; i++) r.v[i] = -a.v[i];
    return r;
}

// negate ? -a : a, branch-free
KZG_HD Fq30 cneg30(const Fq30 &a, bool negate) {
    const int32_t s = negate ? -1 : 0;
    Fq30 r;
#pragma unroll
    for (int i = 0; i < F30_N; i++) r.v[i] = (a.v[i] ^ s) - s;
    return r;
}

// a - b, normalised.  Operand limbs |.| <= 2^29.
KZG_HD Fq30 sub30(const Fq30 &a, const Fq30 &b) {
    Fq30 r;
#pragma unroll
    for (int i = 0; i < F30_N; i++) r.v[i] = a.v[i] - b.v[i];
    return normalize30(r);
}

// a + 2b, normalised.  Operands normalised (3 (2^29 - 1) + 2^29 + carry stays below 2^31).
KZG_HD Fq30 add2x30(const Fq30 &a, const Fq30 &b) {
    Fq30 r;
#pragma unroll
    for (int i = 0; i < F30_N; i++) r.v[i] = a.v[i] + 2 * b.v[i];
    return normalize30(r);
}

// x == 0 mod q for a normalised x with |x| < q (every Montgomery product of operands with |a||b| < 300 q^2)
KZG_HD bool is_zero30(const Fq30 &x) { return x.limbs_all_zero(); }

// saturated 12 x 32 limbs (an integer in [0, 2^384)) <-> 13 balanced 30-bit limbs of the same integer
KZG_HD Fq30 unpack30(const Fq &a) {
    Fq30 r;
#pragma unroll
    for (int i = 0; i < F30_N; i++) {
        int bit = F30_B * i, w = bit >> 5, sh = bit & 31;
        uint32_t lo = a.v[w] >> sh;
        uint32_t hi = (sh > 2 && w + 1 < 12) ? (a.v[w + 1] << (32 - sh)) : 0u;
        r.v[i] = (int32_t)((lo | hi) & F30_MASK);
    }
    return normalize30(r);
}
KZG_HD Fq pack30(Fq30 a) {  // value in [0, 2^384)
#pragma unroll
    for (int i = 0; i < F30_N - 1; i++) {  // balanced -> unsigned digits (floor carries)
        int32_t c = a.v[i] >> F30_B;
        a.v[i] = (int32_t)((uint32_t)a.v[i] & F30_MASK);
        a.v[i + 1] += c;
    }
    Fq r = Fq::zero();
#pragma unroll
    for (int i = 0; i < F30_N; i++) {
        int bit = F30_B * i, w = bit >> 5, sh = bit & 31;
        uint32_t u = (uint32_t)a.v[i];
        if (w < 12) r.v[w] |= u << sh;
        if (sh > 2 && w + 1 < 12) r.v[w + 1] |= u >> (32 - sh);
    }
    return r;
}

KZG_HD Fq30 one30() {
    Fq30 r;
#pragma unroll
    for (int i = 0; i < F30_N; i++) r.v[i] = Fq30Consts::one(i);
    return r;
}

// Montgomery-form conversions: x*2^384 (canonical, saturated)  <->  x*2^390 (balanced limbs)
KZG_HD Fq30 to30(const Fq &a) {  // |result| < 0.51 q
    Fq30 k;
#pragma unroll
    for (int i = 0; i < F30_N; i++) k.v[i] = Fq30Consts::k_to30(i);
    return mul30(unpack30(a), k);
}
KZG_HD Fq from30(const Fq30 &a) {  // a: any lazy value with |a| < 256 q and multiplication-grade limbs
    Fq30 k;
#pragma unroll
    for (int i = 0; i < F30_N; i++) k.v[i] = Fq30Consts::k_from30(i);
    Fq30 t = mul30(a, k);            // |t| < 0.71 q
#pragma unroll
    for (int i = 0; i < F30_N; i++) t.v[i] += Fq30Consts::mod(i);  // in (0.29 q, 1.71 q)
    Fq r = pack30(t);
    reduce_once(r);
    return r;
}

}  // namespace kzg
