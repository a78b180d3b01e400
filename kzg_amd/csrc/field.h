// field.h -- BLS12-381 Fq (381-bit) and Fr (255-bit) Montgomery arithmetic on 32-bit limbs.
//
// Written for the gfx950 VALU: the 32x32+64 multiply-add `v_mad_u64_u32` is the workhorse (288 per
// Fq multiply), everything is fully unrolled so limbs live in VGPRs, no local-memory arrays.
// The same source compiles for the host (g++) so tests/ can exercise it without a GPU and the host
// side of the library can derive twiddles/constants with identical arithmetic.
//
// Stands in for blstrs::Scalar / blst fp (external to the reference: Cargo.toml:27).
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define KZG_HD __host__ __device__ __forceinline__
#else
#define KZG_HD inline __attribute__((always_inline))
#endif

namespace kzg {

struct FqParams {
    static constexpr int N = 12;
    static constexpr uint32_t INV = 0xfffcfffdu;  // -q^-1 mod 2^32
    static KZG_HD uint32_t mod(int i) {
        constexpr uint32_t M[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                                    0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
        return M[i];
    }
    static KZG_HD uint32_t r1(int i) {  // 2^384 mod q
        constexpr uint32_t M[12] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u,
                                    0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
        return M[i];
    }
    static KZG_HD uint32_t r2(int i) {  // 2^768 mod q
        constexpr uint32_t M[12] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu,
                                    0x939d83c0u, 0x67eb88a9u, 0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
        return M[i];
    }
};

struct FrParams {
    static constexpr int N = 8;
    static constexpr uint32_t INV = 0xffffffffu;  // -r^-1 mod 2^32
    static KZG_HD uint32_t mod(int i) {
        constexpr uint32_t M[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                   0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
        return M[i];
    }
    static KZG_HD uint32_t r1(int i) {  // 2^256 mod r
        constexpr uint32_t M[8] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau,
                                   0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
        return M[i];
    }
    static KZG_HD uint32_t r2(int i) {  // 2^512 mod r
        constexpr uint32_t M[8] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu,
                                   0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
        return M[i];
    }
};

template <class P>
struct alignas(16) Fp {  // 16-B aligned so HBM/LDS accesses are dwordx4
    static constexpr int N = P::N;
    uint32_t v[N];

    static KZG_HD Fp zero() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = 0;
        return r;
    }
    static KZG_HD Fp one() {  // Montgomery one
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = P::r1(i);
        return r;
    }
    static KZG_HD Fp r2() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = P::r2(i);
        return r;
    }
    KZG_HD bool is_zero() const {
        uint32_t t = 0;
#pragma unroll
        for (int i = 0; i < N; i++) t |= v[i];
        return t == 0;
    }
    KZG_HD bool operator==(const Fp &o) const {
        uint32_t t = 0;
#pragma unroll
        for (int i = 0; i < N; i++) t |= v[i] ^ o.v[i];
        return t == 0;
    }
    KZG_HD bool operator!=(const Fp &o) const { return !(*this == o); }
};

// r = a - p if a >= p else a   (a < 2p)
template <class P>
KZG_HD void reduce_once(Fp<P> &a) {
    constexpr int N = P::N;
    uint32_t t[N];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t d = (uint64_t)a.v[i] - P::mod(i) - borrow;
        t[i] = (uint32_t)d;
        borrow = (uint32_t)(d >> 63);
    }
    if (!borrow) {
#pragma unroll
        for (int i = 0; i < N; i++) a.v[i] = t[i];
    }
}

template <class P>
KZG_HD Fp<P> add(const Fp<P> &a, const Fp<P> &b) {
    constexpr int N = P::N;
    Fp<P> r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t s = (uint64_t)a.v[i] + b.v[i] + c;
        r.v[i] = (uint32_t)s;
        c = (uint32_t)(s >> 32);
    }
    // both moduli leave the top bit of the top limb clear, so a + b < 2p < 2^(32N): no carry out
    reduce_once(r);
    return r;
}

template <class P>
KZG_HD Fp<P> sub(const Fp<P> &a, const Fp<P> &b) {
    constexpr int N = P::N;
    Fp<P> r;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t d = (uint64_t)a.v[i] - b.v[i] - borrow;
        r.v[i] = (uint32_t)d;
        borrow = (uint32_t)(d >> 63);
    }
    uint32_t mask = 0u - borrow;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t s = (uint64_t)r.v[i] + (P::mod(i) & mask) + c;
        r.v[i] = (uint32_t)s;
        c = (uint32_t)(s >> 32);
    }
    return r;
}

template <class P>
KZG_HD Fp<P> neg(const Fp<P> &a) {
    return sub(Fp<P>::zero(), a);
}

template <class P>
KZG_HD Fp<P> dbl(const Fp<P> &a) {
    return add(a, a);
}

// Montgomery product a*b*R^-1 mod p, CIOS, result fully reduced (< p).
template <class P>
KZG_HD Fp<P> mul_inline(const Fp<P> &a, const Fp<P> &b) {
    constexpr int N = P::N;
    uint32_t t[N + 2];
#pragma unroll
    for (int i = 0; i < N + 2; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < N; j++) {
            uint64_t x = (uint64_t)a.v[j] * b.v[i] + t[j] + c;
            t[j] = (uint32_t)x;
            c = x >> 32;
        }
        uint64_t x = (uint64_t)t[N] + c;
        t[N] = (uint32_t)x;
        t[N + 1] = (uint32_t)(x >> 32);
        uint32_t m = t[0] * P::INV;
        x = (uint64_t)m * P::mod(0) + t[0];
        c = x >> 32;
#pragma unroll
        for (int j = 1; j < N; j++) {
            x = (uint64_t)m * P::mod(j) + t[j] + c;
            t[j - 1] = (uint32_t)x;
            c = x >> 32;
        }
        x = (uint64_t)t[N] + c;
        t[N - 1] = (uint32_t)x;
        t[N] = t[N + 1] + (uint32_t)(x >> 32);
    }
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = t[i];
    reduce_once(r);  // t < 2p and 2p < 2^(32N) for both fields, so t[N] == 0 here
    return r;
}

#if defined(__HIP_DEVICE_COMPILE__)
// On the GPU the multiply is ONE out-of-line function per field (s_swappc call, operands and result in
// VGPR tuples via ext-vector types -- struct arguments would travel through scratch memory).  An
// inlined multiply is ~1250 instructions; a point addition has 10-14 of them, which would put every
// accumulation loop far outside the instruction cache.
template <int N>
struct LimbVec;
template <>
struct LimbVec<12> {
    typedef uint32_t type __attribute__((ext_vector_type(12)));
};
template <>
struct LimbVec<8> {
    typedef uint32_t type __attribute__((ext_vector_type(8)));
};

template <class P>
__device__ __noinline__ typename LimbVec<P::N>::type mul_ool(typename LimbVec<P::N>::type a,
                                                            typename LimbVec<P::N>::type b) {
    Fp<P> x, y;
#pragma unroll
    for (int i = 0; i < P::N; i++) {
        x.v[i] = a[i];
        y.v[i] = b[i];
    }
    Fp<P> z = mul_inline(x, y);
    typename LimbVec<P::N>::type r;
#pragma unroll
    for (int i = 0; i < P::N; i++) r[i] = z.v[i];
    return r;
}

template <class P>
KZG_HD Fp<P> mul(const Fp<P> &a, const Fp<P> &b) {
    typename LimbVec<P::N>::type x, y;
#pragma unroll
    for (int i = 0; i < P::N; i++) {
        x[i] = a.v[i];
        y[i] = b.v[i];
    }
    typename LimbVec<P::N>::type z = mul_ool<P>(x, y);
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < P::N; i++) r.v[i] = z[i];
    return r;
}
#else
template <class P>
KZG_HD Fp<P> mul(const Fp<P> &a, const Fp<P> &b) {
    return mul_inline(a, b);
}
#endif

template <class P>
KZG_HD Fp<P> sqr(const Fp<P> &a) {
    return mul(a, a);
}

template <class P>
KZG_HD Fp<P> to_mont(const Fp<P> &a) {
    return mul(a, Fp<P>::r2());
}

template <class P>
KZG_HD Fp<P> from_mont(const Fp<P> &a) {
    Fp<P> o = Fp<P>::zero();
    o.v[0] = 1;
    return mul(a, o);
}

// a^e for a 64-bit exponent (pow_vartime(&[e]))
template <class P>
KZG_HD Fp<P> pow_u64(const Fp<P> &a, uint64_t e) {
    Fp<P> acc = Fp<P>::one();
    Fp<P> base = a;
    while (e) {
        if (e & 1) acc = mul(acc, base);
        base = sqr(base);
        e >>= 1;
    }
    return acc;
}

// a^(p-2): Field::invert() by Fermat (a != 0).  Not unrolled: one loop over the exponent bits.
template <class P>
KZG_HD Fp<P> inv(const Fp<P> &a) {
    constexpr int N = P::N;
    uint32_t e[N];
    uint32_t borrow = 2;  // e = p - 2 (r's low limb is 1, so the borrow must ripple)
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t d = (uint64_t)P::mod(i) - borrow;
        e[i] = (uint32_t)d;
        borrow = (uint32_t)(d >> 63);
    }
    Fp<P> acc = Fp<P>::one();
    for (int i = N * 32 - 1; i >= 0; i--) {
        acc = sqr(acc);
        uint32_t limb = 0;
#pragma unroll
        for (int k = 0; k < N; k++) limb = (k == (i >> 5)) ? e[k] : limb;
        if ((limb >> (i & 31)) & 1) acc = mul(acc, a);
    }
    return acc;
}

template <class P>
KZG_HD Fp<P> from_u64(uint64_t x) {
    Fp<P> t = Fp<P>::zero();
    t.v[0] = (uint32_t)x;
    t.v[1] = (uint32_t)(x >> 32);
    return to_mont(t);
}

// true if the canonical (non-Montgomery) value in a is < p
template <class P>
KZG_HD bool is_canonical(const Fp<P> &a) {
    constexpr int N = P::N;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t d = (uint64_t)a.v[i] - P::mod(i) - borrow;
        borrow = (uint32_t)(d >> 63);
    }
    return borrow != 0;
}

typedef Fp<FqParams> Fq;
typedef Fp<FrParams> Fr;

// Scalar::root_of_unity() = 7^((r-1)/2^32) (ff::PrimeField, used at src/ft.rs:73), Montgomery form
KZG_HD Fr fr_root_of_unity() {
    Fr t;
    constexpr uint32_t M[8] = {0x439f0d2bu, 0x3829971fu, 0x8c2280b9u, 0xb6368350u,
                               0x22c813b4u, 0xd09b6819u, 0xdfe81f20u, 0x16a2a19eu};
#pragma unroll
    for (int i = 0; i < 8; i++) t.v[i] = M[i];
    return to_mont(t);
}
constexpr uint32_t FR_TWO_ADICITY = 32;       // Scalar::S
constexpr uint64_t FR_MULT_GENERATOR = 7;     // Scalar::multiplicative_generator()

}  // namespace kzg
