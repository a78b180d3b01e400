// naf.h -- width-18 non-adjacent-form recoding of a scalar for the positional tables (kzg_srs::naf, msm.hip k_naf_recode).
// Host + device: tests/host_math.cpp checks it against a big-integer restatement.
//
// k = sum_t d_t 2^(p_t) with every d_t odd, |d_t| < 2^17, and p_(t+1) - p_t >= 18: on average 254 / 19 + 0.5 = 13.9 digits for a
// balanced 254-bit scalar instead of the 15 of fixed 17-bit windows.  A digit record:
//   bits 0..16  bucket index (|d| - 1) / 2      bit 17  sign (already combined with `flip`)
//   bits 18..25 bit position p (= table row)    bit 31  valid
#pragma once
#include <stdint.h>

#ifndef KZG_HD
#if defined(__HIPCC__)
#define KZG_HD __host__ __device__ __forceinline__
#else
#define KZG_HD inline
#endif
#endif

namespace kzg {

constexpr int NAF_W = 18, NAF_MAX_DIGITS = 15;
constexpr uint32_t NAF_REC_VALID = 0x80000000u;

KZG_HD int naf_ctz64(uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffsll((unsigned long long)v) - 1;
#else
    return __builtin_ctzll(v);
#endif
}

// L: the scalar's 8 little-endian 32-bit limbs followed by 4 zero limbs (value below 2^254).  out[0 .. return value) = records.
// Walks from bit 0 with a carry: without a carry the next digit starts at the next 1 bit; with one (the previous digit was
// negative, i.e. 2^18 was borrowed from above) at the next 0 bit, which the carry turns into a 1.  The digit is the 18 bits from
// there, read as negative when its top bit is set.
KZG_HD int naf18_digits(const uint32_t *L, uint32_t flip, uint32_t *out) {
    int pos = 0, cnt = 0;
    uint32_t carry = 0;
    while (pos < 256 && cnt < NAF_MAX_DIGITS) {
        const int w = pos >> 5, sh = pos & 31;
        const uint64_t lo = ((uint64_t)L[w + 1] << 32) | L[w];
        const uint64_t v = sh ? (lo >> sh) | ((uint64_t)L[w + 2] << (64 - sh)) : lo;  // bits pos .. pos + 63
        const uint64_t look = carry ? ~v : v;
        if (look == 0) {  // 64 zeros (ones under a carry): no digit starts here
            pos += 64;
            continue;
        }
        const int p = pos + naf_ctz64(look);
        if (p >= 256) break;  // a carry running into the zero padding: impossible for a value below 2^254
        const int w2 = p >> 5, sh2 = p & 31;
        const uint64_t lo2 = ((uint64_t)L[w2 + 1] << 32) | L[w2];
        uint32_t x = ((uint32_t)(lo2 >> sh2) & ((1u << NAF_W) - 1u)) | 1u;  // the carry, if any, made bit p a one
        uint32_t neg = 0;
        if (x >= (1u << (NAF_W - 1))) {
            x = (1u << NAF_W) - x;
            neg = 1;
        }
        carry = neg;
        out[cnt++] = ((x - 1u) >> 1) | ((neg ^ flip) << 17) | ((uint32_t)p << 18) | NAF_REC_VALID;
        pos = p + NAF_W;
    }
    return cnt;
}

}  // namespace kzg
