// msm_sort.h -- device helpers shared by the sort kernels of msm.hip and msm_wide.hip: scalar -> signed window digits, block scans.
#pragma once
#include "msm_internal.h"

namespace kzg {

// ---------------------------------------------------------------------------------------------
// scalar -> signed digits
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t sel8(const uint32_t s[8], int idx) {
    uint32_t r = s[0];
    r = idx == 1 ? s[1] : r;
    r = idx == 2 ? s[2] : r;
    r = idx == 3 ? s[3] : r;
    r = idx == 4 ? s[4] : r;
    r = idx == 5 ? s[5] : r;
    r = idx == 6 ? s[6] : r;
    r = idx == 7 ? s[7] : r;
    return r;
}

// The digit extraction needs the canonical value.  Montgomery input: from_mont() returns it.  Canonical input is taken mod r
// (a 256-bit value is < 2.3 r: at most two subtractions), so that a caller's non-canonical scalar gives the same group element
// at every window width -- the balanced c = 17 recoding (r - k) and the 15 x 17-bit window split both assume k < r.
__device__ __forceinline__ void load_scalar(const Fr *scalars, size_t i, int sfmt, uint32_t s[8]) {
    Fr v = scalars[i];
    if (sfmt == KZG_FR_MONT_LE_32) {
        v = from_mont(v);
    } else {
#pragma unroll
        for (int rep = 0; rep < 2; rep++) {
            uint32_t d[8];
            uint64_t bw = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                uint64_t t = (uint64_t)v.v[k] - FrParams::mod(k) - bw;
                d[k] = (uint32_t)t;
                bw = (t >> 63) & 1u;
            }
            if (!bw) {
#pragma unroll
                for (int k = 0; k < 8; k++) v.v[k] = d[k];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) s[k] = v.v[k];
}

// calls f(w, magnitude in [1, 2^(c-1)], negative) for every non-zero signed digit.
// balanced (the c = 17 single-pass mode, W * c = 255): a scalar with bit 254 set is replaced by r - k < 2^254 with every digit
// sign flipped (k = -(r - k) mod r), so the top window's raw digit stays <= 2^(c-1) and nothing carries out of window W - 1.
template <class F>
__device__ __forceinline__ void for_each_digit(const uint32_t s_in[8], int c, int W, bool balanced, F f) {
    uint32_t s[8];
#pragma unroll
    for (int k = 0; k < 8; k++) s[k] = s_in[k];
    uint32_t flip = 0;
    if (balanced && (s[7] & 0x40000000u)) {
        uint64_t bw = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            uint64_t d = (uint64_t)FrParams::mod(k) - s[k] - bw;
            s[k] = (uint32_t)d;
            bw = (d >> 63) & 1u;
        }
        flip = 1;
    }
    uint32_t carry = 0;
    const uint32_t mask = (1u << c) - 1u;
    const uint32_t half = 1u << (c - 1);
    for (int w = 0; w < W; w++) {
        int o = w * c;
        int limb = o >> 5, sh = o & 31;
        uint32_t lo = sel8(s, limb);
        uint32_t hi = (limb < 7) ? sel8(s, limb + 1) : 0u;
        uint64_t both = ((uint64_t)hi << 32) | lo;
        uint32_t raw = ((uint32_t)(both >> sh) & mask) + carry;
        uint32_t neg = raw > half ? 1u : 0u;
        uint32_t mag = neg ? ((1u << c) - raw) : raw;
        carry = neg;
        if (mag) f(w, mag, neg ^ flip);
    }
}

// The same for a window width fixed at compile time: every limb index and shift is a constant, so the digits come straight out
// of registers (the generic version selects limbs with a chain of compares).  Used for the production width c = 17, W = 15.
template <int C, int WN, class F>
__device__ __forceinline__ void for_each_digit_fixed(const uint32_t s_in[8], bool balanced, F f) {
    uint32_t s[9];
#pragma unroll
    for (int k = 0; k < 8; k++) s[k] = s_in[k];
    s[8] = 0;
    uint32_t flip = 0;
    if (balanced && (s[7] & 0x40000000u)) {
        uint64_t bw = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            uint64_t d = (uint64_t)FrParams::mod(k) - s[k] - bw;
            s[k] = (uint32_t)d;
            bw = (d >> 63) & 1u;
        }
        flip = 1;
    }
    uint32_t carry = 0;
    constexpr uint32_t mask = (1u << C) - 1u;
    constexpr uint32_t half = 1u << (C - 1);
#pragma unroll
    for (int w = 0; w < WN; w++) {
        constexpr int dummy = 0;
        (void)dummy;
        const int o = w * C, limb = o >> 5, sh = o & 31;
        const uint32_t lo = s[limb], hi = s[limb + 1 > 8 ? 8 : limb + 1];
        const uint64_t both = ((uint64_t)hi << 32) | lo;
        const uint32_t raw = ((uint32_t)(both >> sh) & mask) + carry;
        const uint32_t neg = raw > half ? 1u : 0u;
        const uint32_t mag = neg ? ((1u << C) - raw) : raw;
        carry = neg;
        if (mag) f(w, mag, neg ^ flip);
    }
}

__device__ __forceinline__ uint32_t block_scan_256(uint32_t v, uint32_t *lds, uint32_t *total_out) {
    // exclusive scan of one value per thread over a 256-thread block (4 waves): wave shuffles + one LDS pass
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        uint32_t t = lds[w];
        if (w < wave) woff += t;
        tot += t;
    }
    __syncthreads();
    *total_out = tot;
    return incl - v + woff;
}


// exclusive scan of one value per thread over a 1024-thread block; *total_out = the block sum
__device__ __forceinline__ uint32_t block_scan_1024(uint32_t v, uint32_t *wsum /* 16 words of LDS */, uint32_t *total_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t u = __shfl_up(incl, off, 64);
        if (lane >= off) incl += u;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) {
        uint32_t u = wsum[w];
        if (w < wave) woff += u;
        tot += u;
    }
    __syncthreads();
    *total_out = tot;
    return incl - v + woff;
}

// Level-1 record = (entry word, low 6 bucket bits).  Two encodings: 8 bytes (uint2) in general; 4 bytes when the table index
// fits 25 bits (table rows x points <= 2^25, i.e. up to 2^21 points with 15 rows): index | sign << 25 | low bits << 26 --
// half the record traffic of both levels.
struct Rec8 {
    typedef uint2 T;
    static __device__ __forceinline__ T pack(uint32_t index, uint32_t neg, uint32_t lo) { return make_uint2(index | (neg << 31), lo); }
    static __device__ __forceinline__ uint32_t entry(const T &r) { return r.x; }
    static __device__ __forceinline__ uint32_t lo(const T &r) { return r.y; }
    static __device__ __forceinline__ T invalid() { return make_uint2(0u, 0xffffffffu); }
    static __device__ __forceinline__ bool valid(const T &r) { return r.y != 0xffffffffu; }
};
struct Rec4 {
    typedef uint32_t T;
    static __device__ __forceinline__ T pack(uint32_t index, uint32_t neg, uint32_t lo) { return index | (neg << 25) | (lo << 26); }
    static __device__ __forceinline__ uint32_t entry(const T &r) { return (r & 0x1ffffffu) | (((r >> 25) & 1u) << 31); }
    static __device__ __forceinline__ uint32_t lo(const T &r) { return r >> 26; }
    static __device__ __forceinline__ T invalid() { return 0xffffffffu; }  // index 2^25 - 1 with sign and lo = 63: never packed (index < 2^25 - 1)
    static __device__ __forceinline__ bool valid(const T &r) { return r != 0xffffffffu; }
};
constexpr uint64_t REC4_MAX_INDEX = (1ull << 25) - 1;  // exclusive bound on table rows x padded points for Rec4
// c = 20 (msm_wide.hip): (entry word, low 9 bucket bits | bin << 16): the bin rides along so that level 1 can stage finished records
struct Rec20 {
    typedef uint2 T;
    static __device__ __forceinline__ uint32_t entry(const T &r) { return r.x; }
    static __device__ __forceinline__ uint32_t lo(const T &r) { return r.y & 0xffffu; }
    static __device__ __forceinline__ T invalid() { return make_uint2(0u, 0xffffffffu); }
    static __device__ __forceinline__ bool valid(const T &r) { return r.y != 0xffffffffu; }
};

}  // namespace kzg
