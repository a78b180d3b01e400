// field29.h -- Fq in an UNSATURATED representation: 14 limbs of 29 bits, Montgomery radix R29 = 2^406.
//
// Why: with saturated 32-bit limbs every partial product of a multi-precision multiply needs a carry
// fold (v_mad_u64_u32 + v_addc_co_u32; 288 of each per Fq multiply).  With 29-bit limbs a whole column
// of a 14 x 14 product-scanning multiply (<= 28 products < 2^58) sums in ONE 64-bit accumulator with
// no carry handling at all: acc = a*b + acc is a bare v_mad_u64_u32.  392 mads + ~130 other
// instructions instead of 288 + ~500, and -- because R29 = 2^406 is 2^25 times larger than q -- no final
// conditional subtraction: for inputs below 2^12 q the Montgomery result is already below 3q.
//
// Values are kept lazily reduced ("< c*q" classes documented at each operation); limbs of anything fed
// to mul29 are normalised (< 2^29, one operand may be < 2^30).  Used only inside the MSM accumulation
// kernel; results are converted back to the canonical saturated Montgomery form when a partial is
// written out, so nothing outside sees this representation.
#pragma once
#include "field.h"

namespace kzg {

#include "fq29_consts.inc"

constexpr int F29_N = 14;
constexpr uint32_t F29_MASK = (1u << 29) - 1u;

struct Fq29 {
    uint32_t v[F29_N];
    KZG_HD bool limbs_all_zero() const {
        uint32_t t = 0;
#pragma unroll
        for (int i = 0; i < F29_N; i++) t |= v[i];
        return t == 0;
    }
};

// carry-propagate so that limbs 0..12 are < 2^29 (limb 13 keeps the rest); limbs must be non-negative
KZG_HD Fq29 normalize29(Fq29 a) {
#pragma unroll
    for (int i = 0; i < F29_N - 1; i++) {
        a.v[i + 1] += a.v[i] >> 29;
        a.v[i] &= F29_MASK;
    }
    return a;
}

// Montgomery product a*b/R29 mod q (lazy: result < q*(1 + a*b/(q*R29)) < 3q for a, b < 2^12 q).
// Limbs of a and b < 2^29 (one of the two operands may have limbs < 2^30).  Result limbs normalised.
KZG_HD Fq29 mul29_inline(const Fq29 &a, const Fq29 &b) {
    uint32_t m[F29_N];
    Fq29 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < F29_N; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * Fq29Consts::mod(k - i);
        m[k] = ((uint32_t)acc * Fq29Consts::INV) & F29_MASK;
        acc += (uint64_t)m[k] * Fq29Consts::mod(0);
        acc >>= 29;
    }
#pragma unroll
    for (int k = F29_N; k < 2 * F29_N - 1; k++) {
#pragma unroll
        for (int i = k - F29_N + 1; i < F29_N; i++) {
            acc += (uint64_t)a.v[i] * b.v[k - i];
            acc += (uint64_t)m[i] * Fq29Consts::mod(k - i);
        }
        r.v[k - F29_N] = (uint32_t)acc & F29_MASK;
        acc >>= 29;
    }
    r.v[F29_N - 1] = (uint32_t)acc;
    return r;
}

// (a*b + c*d)/R29 with ONE reduction: both double-width products accumulate in the same columns
// (<= 14 + 14 + 14 products below 2^58 per column < 2^64).  All four operands limb-normalised (< 2^29).
// Result < q (1 + (a*b + c*d)/(q R29)): lazy, limbs normalised.  Inlined at its (single) call site.
KZG_HD Fq29 muladd29_inline(const Fq29 &a, const Fq29 &b, const Fq29 &c, const Fq29 &d) {
    uint32_t m[F29_N];
    Fq29 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < F29_N; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) {
            acc += (uint64_t)a.v[i] * b.v[k - i];
            acc += (uint64_t)c.v[i] * d.v[k - i];
        }
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * Fq29Consts::mod(k - i);
        m[k] = ((uint32_t)acc * Fq29Consts::INV) & F29_MASK;
        acc += (uint64_t)m[k] * Fq29Consts::mod(0);
        acc >>= 29;
    }
#pragma unroll
    for (int k = F29_N; k < 2 * F29_N - 1; k++) {
#pragma unroll
        for (int i = k - F29_N + 1; i < F29_N; i++) {
            acc += (uint64_t)a.v[i] * b.v[k - i];
            acc += (uint64_t)c.v[i] * d.v[k - i];
            acc += (uint64_t)m[i] * Fq29Consts::mod(k - i);
        }
        r.v[k - F29_N] = (uint32_t)acc & F29_MASK;
        acc >>= 29;
    }
    r.v[F29_N - 1] = (uint32_t)acc;
    return r;
}

// Montgomery square: the 91 cross products are formed once against a pre-doubled copy of a (limbs < 2^30),
// 105 + 196 mads instead of 392.
KZG_HD Fq29 sqr29_inline(const Fq29 &a) {
    uint32_t m[F29_N], d[F29_N];
    Fq29 r;
#pragma unroll
    for (int i = 0; i < F29_N; i++) d[i] = a.v[i] << 1;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < F29_N; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) acc += (uint64_t)a.v[i] * d[k - i];
        if ((k & 1) == 0) acc += (uint64_t)a.v[k / 2] * a.v[k / 2];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * Fq29Consts::mod(k - i);
        m[k] = ((uint32_t)acc * Fq29Consts::INV) & F29_MASK;
        acc += (uint64_t)m[k] * Fq29Consts::mod(0);
        acc >>= 29;
    }
#pragma unroll
    for (int k = F29_N; k < 2 * F29_N - 1; k++) {
#pragma unroll
        for (int i = k - F29_N + 1; 2 * i < k; i++) acc += (uint64_t)a.v[i] * d[k - i];
        if ((k & 1) == 0) acc += (uint64_t)a.v[k / 2] * a.v[k / 2];
#pragma unroll
        for (int i = k - F29_N + 1; i < F29_N; i++) acc += (uint64_t)m[i] * Fq29Consts::mod(k - i);
        r.v[k - F29_N] = (uint32_t)acc & F29_MASK;
        acc >>= 29;
    }
    r.v[F29_N - 1] = (uint32_t)acc;
    return r;
}

#if defined(__HIP_DEVICE_COMPILE__)
typedef uint32_t u32x14 __attribute__((ext_vector_type(14)));
__device__ __noinline__ u32x14 sqr29_ool(u32x14 a) {
    Fq29 x;
#pragma unroll
    for (int i = 0; i < F29_N; i++) x.v[i] = a[i];
    Fq29 z = sqr29_inline(x);
    u32x14 r;
#pragma unroll
    for (int i = 0; i < F29_N; i++) r[i] = z.v[i];
    return r;
}
KZG_HD Fq29 sqr29(const Fq29 &a) {
    u32x14 x;
#pragma unroll
    for (int i = 0; i < F29_N; i++) x[i] = a.v[i];
    u32x14 z = sqr29_ool(x);
    Fq29 r;
#pragma unroll
    for (int i = 0; i < F29_N; i++) r.v[i] = z[i];
    return r;
}
__device__ __noinline__ u32x14 mul29_ool(u32x14 a, u32x14 b) {
    Fq29 x, y;
#pragma unroll
    for (int i = 0; i < F29_N; i++) {
        x.v[i] = a[i];
        y.v[i] = b[i];
    }
    Fq29 z = mul29_inline(x, y);
    u32x14 r;
#pragma unroll
    for (int i = 0; i < F29_N; i++) r[i] = z.v[i];
    return r;
}
KZG_HD Fq29 mul29(const Fq29 &a, const Fq29 &b) {
    u32x14 x, y;
#pragma unroll
    for (int i = 0; i < F29_N; i++) {
        x[i] = a.v[i];
        y[i] = b.v[i];
    }
    u32x14 z = mul29_ool(x, y);
    Fq29 r;
#pragma unroll
    for (int i = 0; i < F29_N; i++) r.v[i] = z[i];
    return r;
}
#else
KZG_HD Fq29 mul29(const Fq29 &a, const Fq29 &b) { return mul29_inline(a, b); }
KZG_HD Fq29 sqr29(const Fq29 &a) { return sqr29_inline(a); }
#endif

// a - b + C*q, C in {4, 8, 16, 32}: limb-wise (no borrows: every low limb of the constant is >= 2^30 - 2),
// then normalised.  Requires b normalised with value < C*q (top limb), a limbs < 2^30.
template <int C>
KZG_HD Fq29 sub29(const Fq29 &a, const Fq29 &b) {
    Fq29 r;
#pragma unroll
    for (int i = 0; i < F29_N; i++) {
        uint32_t c = C == 4 ? Fq29Consts::sub4(i) : C == 8 ? Fq29Consts::sub8(i) : C == 16 ? Fq29Consts::sub16(i) : Fq29Consts::sub32(i);
        r.v[i] = a.v[i] + c - b.v[i];
    }
    return normalize29(r);
}

// a + 2b, normalised
KZG_HD Fq29 add2x29(const Fq29 &a, const Fq29 &b) {
    Fq29 r;
#pragma unroll
    for (int i = 0; i < F29_N; i++) r.v[i] = a.v[i] + (b.v[i] << 1);
    return normalize29(r);
}

// x == 0 mod q for a Montgomery product x of operands below 2^12 q: x < q (1 + 2^24 q / R29) < 2q, so the only
// multiples of q it can be are 0 and q.  x normalised.
KZG_HD bool is_zero_mod_q_product(const Fq29 &x) {
    uint32_t d0 = 0, d1 = 0;
#pragma unroll
    for (int i = 0; i < F29_N; i++) {
        d0 |= x.v[i];
        d1 |= x.v[i] ^ Fq29Consts::q1(i);
    }
    return d0 == 0 || d1 == 0;
}

// saturated 12 x 32 limbs <-> 14 x 29 limbs of the same integer (< 2^384)
KZG_HD Fq29 unpack29(const Fq &a) {
    Fq29 r;
#pragma unroll
    for (int i = 0; i < F29_N; i++) {
        int bit = 29 * i, w = bit >> 5, sh = bit & 31;
        uint32_t lo = a.v[w] >> sh;
        uint32_t hi = (sh > 3 && w + 1 < 12) ? (a.v[w + 1] << (32 - sh)) : 0u;
        r.v[i] = (lo | hi) & F29_MASK;
    }
    return r;
}
KZG_HD Fq pack29(const Fq29 &a) {  // a normalised and < 2^384
    Fq r = Fq::zero();
#pragma unroll
    for (int i = 0; i < F29_N; i++) {
        int bit = 29 * i, w = bit >> 5, sh = bit & 31;
        if (w < 12) r.v[w] |= a.v[i] << sh;
        if (sh > 3 && w + 1 < 12) r.v[w + 1] |= a.v[i] >> (32 - sh);
    }
    return r;
}

// Montgomery-form conversions: x*2^384 (canonical, saturated)  <->  x*2^406 (29-bit limbs)
KZG_HD Fq29 to29(const Fq &a) {
    Fq29 k;
#pragma unroll
    for (int i = 0; i < F29_N; i++) k.v[i] = Fq29Consts::k_to29(i);
    return mul29(unpack29(a), k);  // < 2q... (input < q)
}
KZG_HD Fq from29(const Fq29 &a) {  // a: any lazy value < 2^12 q with normalised limbs
    Fq29 k;
#pragma unroll
    for (int i = 0; i < F29_N; i++) k.v[i] = Fq29Consts::k_from29(i);
    Fq r = pack29(mul29(a, k));    // < 3q < 2^384
    reduce_once(r);                // canonical after at most two subtractions of q
    reduce_once(r);
    return r;
}

KZG_HD Fq29 one29() {
    Fq29 r;
#pragma unroll
    for (int i = 0; i < F29_N; i++) r.v[i] = Fq29Consts::one(i);
    return r;
}

}  // namespace kzg
