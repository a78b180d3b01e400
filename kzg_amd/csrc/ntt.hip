// ntt.hip -- radix-2 NTT / iNTT over Fr, natural order in and out, in place.
//
// Semantics of EvaluationDomain::fft / ifft (src/ft.rs:111-140) and best_fft / serial_fft
// (src/ft.rs:274-333): out[i] = sum_j a[j] w^(ij), w = root_of_unity^(2^(32-exp)) per compute_omega
// (src/ft.rs:55-76); the inverse uses w^-1 and scales by d^-1.  The reference's bit-reversal +
// log n in-place passes are a CPU schedule; on the MI355X the transform is a four-step
// decomposition n = n1 * n2 whose sub-transforms (<= 2^12 points, 128 KiB) run entirely in the CU's
// 160 KiB LDS:
//   pass 1: for every column j2, an n1-point NTT over stride-n2 elements, times w^(j2*k1)
//           (two-level twiddle table), written to scratch in the same layout;
//   pass 2: for every row k1, an n2-point NTT over contiguous elements, written transposed
//           (index k1 + n1*k2) back into the caller's buffer -- natural order, no bit-reversal pass.
// Each block takes VEC adjacent columns / rows so every global access is a >= 64..128 B segment.
// Data may be canonical or Montgomery: the butterflies are linear and the twiddles are Montgomery
// constants, so mont_mul(a, w) preserves whichever form a is in.
#include "common.h"
#include "fr29.h"
#include <algorithm>

namespace kzg {

// A twiddle table: entry i is the constant c_i as the pair (w[i], wp[i]) = (c_i, floor(c_i 2^261 / r)) in 9 x 29-bit limbs -- what
// the Shoup product of fr29.h multiplies by (mulshoup29).  Two arrays of 36-byte elements.
struct Tw29 {
    Fr29 *w = nullptr, *wp = nullptr;
};

struct NttPlan {
    uint32_t log_n = 0, k1 = 0, k2 = 0;
    int inverse = 0;
    Tw29 tw1;    // w_{n1}^i, i < n1/2 (or w_n^i for the single-tile case)
    Tw29 tw2;    // w_{n2}^i, i < n2/2
    Tw29 tw_lo;  // w_n^i, i < 2^lo_bits
    Tw29 tw_hi;  // w_n^(i << lo_bits)
    uint32_t lo_bits = 0;
    // log_n <= 21: w_n^(j2*k1) * scale at [k1*n2 + j2] -- ONE inter-pass multiply, scale folded in.  This one table stays in the
    // Montgomery-29 form (w 2^261 mod r, 36 B per element, multiplied by mul29r): pass 1's epilogue is where the kernel waits for
    // HBM, and the (w, wp) pair would double what it reads there (same-box: 63 -> 68 us for pass 1 at 2^20 with the pair)
    Fr29 *tw_full = nullptr;
    Fr29 scale, scale_p;  // d^-1 for the inverse, one otherwise (pair)
    // three-pass plans (ntt_run3): k1 = 8 outer columns, then a (k2 x k3)-point inner transform per outer index; tw1 / tw2 / tw3 the
    // stage twiddles of the three passes, tw_lo / tw_hi the two-level table of w_n (inter-pass product of pass A), tw_full the inner
    // inter-pass table w'^(j3 k2) * scale of 2^(k2 + k3) entries (w' = w_n^(2^k1))
    uint32_t k3 = 0;
    Tw29 tw3;
};

Fr host_omega(uint32_t exp) {
    // Scalar::root_of_unity().pow_vartime(&[1 << (Scalar::S - exp)])   (src/ft.rs:73)
    return pow_u64(fr_root_of_unity(), 1ull << (FR_TWO_ADICITY - exp));
}

__global__ __launch_bounds__(256) void k_pow_table(Fr base, Fr scale, size_t count, Fr *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    out[i] = mul(scale, pow_u64(base, (uint64_t)i));
}

int pow_table(kzg_ctx *ctx, hipStream_t stream, const Fr &base_mont, const Fr &scale_mont, size_t count, Fr *d_out) {
    if (!count) return KZG_OK;
    KZG_LAUNCH(ctx, stream, "k_pow_table", k_pow_table, (unsigned)((count + 255) / 256), 256, 0, base_mont, scale_mont,
               count, d_out);
    return KZG_OK;
}

// twiddle tables for the kernels: base^i as the pair (w, floor(w 2^261 / r)) in 29-bit limbs
__global__ __launch_bounds__(256) void k_pow_table29(Fr base, size_t count, Fr29 *out_w, Fr29 *out_wp) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Fr29 w, wp;
    fr29_shoup_from_twiddle(fr29_twiddle_from_mont(pow_u64(base, (uint64_t)i)), w, wp);
    out_w[i] = w;
    out_wp[i] = wp;
}

// full inter-pass twiddle table: entry [k1*n2 + j2] = w_n^(j2*k1) * scale * 2^261 mod r (one 36-byte read replaces one of the two
// inter-pass multiplications)
__global__ __launch_bounds__(256) void k_twiddle_full(Tw29 tw_lo, Tw29 tw_hi, uint32_t lo_bits, uint32_t k2, Fr scale_mont, size_t n,
                                                      Fr29 *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t ex = (uint64_t)(i & (((size_t)1 << k2) - 1)) * (uint64_t)(i >> k2);
    // (table construction: through the Montgomery-29 form of the product, then to the pair)
    const size_t ih = ex >> lo_bits, il = ex & ((1u << lo_bits) - 1);
    Fr29 t = mulshoup29(fr29_twiddle_from_mont(scale_mont), tw_hi.w[ih], tw_hi.wp[ih]);   // scale 2^261 w_hi (mod r), below 2r
    t = mulshoup29(t, tw_lo.w[il], tw_lo.wp[il]);
    out[i] = fr29_unpack(fr29_pack_canonical(t));  // canonical limbs: usable as the small operand of mul29r
}

extern __shared__ __attribute__((aligned(16))) Fr29 lds_fr29[];

__device__ __forceinline__ uint32_t bitrev(uint32_t x, uint32_t bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

// LDS bank swizzle.  Every access set of these kernels is "idx = const ^ (lane bit i -> idx bit B[i])": radix-4
// stages leave two index bits fixed and spread the lanes over bits up to 6, the tile loads place consecutive
// sources at bit-reversed positions (lane bits -> top index bits), the stores interleave the vectors.  With plain
// lds[idx] those sets alias on 4..32 lanes per bank (rocprofv3: 80 % of LDS cycles were conflict replays).
// Elements are 9 dwords (odd), so 32 lanes are conflict-free iff their indices differ mod 32; the linear map
// phys = idx ^ XOR_{b>=5, idx bit b set} MASK[b] with the per-shape masks below (tools/find_lds_swizzle.py
// checks every pattern's GF(2) rank) makes that hold for every phase.  Index [t][vec_log], 5 bits per mask (vec_log 3..5: the wide
// tiles of the three-pass transforms, 16 or 32 short vectors side by side).
static const uint64_t LDS_SWIZZLE[13][6] = {
    {0x0ull, 0x0ull, 0x0ull, 0x0ull, 0x0ull, 0x0ull},  // t = 0
    {0x0ull, 0x0ull, 0x0ull, 0x0ull, 0x0ull, 0x0ull},  // t = 1
    {0x0ull, 0x0ull, 0x0ull, 0x0ull, 0x0ull, 0x0ull},  // t = 2
    {0x0ull, 0x0ull, 0x0ull, 0x0ull, 0x0ull, 0x0ull},  // t = 3
    {0x0ull, 0x0ull, 0x0ull, 0x0ull, 0x0ull, 0x0ull},  // t = 4
    {0x0ull, 0x1full, 0x36dull, 0x4acbull, 0x7728bull, 0x1c7076dull},  // t = 5
    {0x7ull, 0x33full, 0x3355ull, 0xfb36dull, 0x6fb36dull, 0x306fb36dull},  // t = 6
    {0x36dull, 0x4acbull, 0xfb36dull, 0x6fb36dull, 0x2d9f8dbbull, 0x4966d065dull},  // t = 7
    {0xb3full, 0xf0b3full, 0x1c7076dull, 0x14cb46bbull, 0x5fc518b6dull, 0x0ull},  // t = 8
    {0x94d9bull, 0x1c7076dull, 0x2d9f8dbbull, 0x4966d065dull, 0x0ull, 0x0ull},  // t = 9
    {0x1c7076dull, 0x306fb36dull, 0x6997f0f6dull, 0x0ull, 0x0ull, 0x0ull},  // t = 10
    {0x306fb36dull, 0x4966d065dull, 0x0ull, 0x0ull, 0x0ull, 0x0ull},  // t = 11
    {0x6997f0f6dull, 0x0ull, 0x0ull, 0x0ull, 0x0ull, 0x0ull},  // t = 12
};

__device__ __forceinline__ uint32_t swz(uint32_t idx, uint64_t masks) {
    uint32_t x = idx;
#pragma unroll
    for (int b = 0; b < 9; ++b) {
        uint32_t bit = (uint32_t)((int32_t)(idx << (26 - b)) >> 31);  // all-ones iff idx bit 5+b is set
        x ^= bit & ((uint32_t)(masks >> (5 * b)) & 31u);
    }
    return x;
}

// swz(1 << k): a single set bit contributes its own mask (bits >= 5) or nothing -- two scalar instructions for a wave-uniform k
__device__ __forceinline__ uint32_t swz_bit(uint32_t k, uint64_t masks) {
    return (1u << k) ^ (k >= 5 ? (uint32_t)(masks >> (5 * (k - 5))) & 31u : 0u);
}

// DIT stages over `vec` independent 2^t-point vectors held in LDS at lds[swz(v << t | pos)] (36-byte elements),
// input already in bit-reversed position order.  Stage pairs
// (s, s+1) are fused: a thread takes the 4 elements {p, p+m, p+2m, p+3m}, m = 2^s, through both stages in
// registers (radix-4), halving the LDS round trips and barriers; an odd t starts with one radix-2 stage.
// Lazy arithmetic (fr29.h): twiddle products are Shoup products (below 2r, normalised); the butterflies' sums and differences
// (u + t, u - t + 4r) stay UNnormalised through the pair -- a sum that is multiplied next goes into the product as it is -- and the
// four results are normalised once, when they are stored.  Values grow by at most 8r per pair along the element chain that is
// never multiplied (12.4 r in the first pair, whose unmultiplied sum needs 8r): below 53 r after the six pairs of the largest
// tile, against the 70 r = 2^261 the product accepts (tests/test_host_math.py drives exactly this chain).
__device__ __forceinline__ void lds_ntt_stages29(Fr29 *lds, uint32_t t, uint32_t vec, const Tw29 tw, uint64_t sw) {
    uint32_t s = 0;
    if (t & 1) {  // radix-2, twiddle 1; both inputs are raw 256-bit loads
        const uint32_t total = vec << (t - 1);
        for (uint32_t b = threadIdx.x; b < total; b += blockDim.x) {
            uint32_t p0 = swz(b << 1, sw), p1 = p0 ^ 1;
            Fr29 u = lds[p0], w = lds[p1], sm, df;
            fr29_butterfly_lazy(u, w, sm, df);
            lds[p0] = fr29_normalize(sm);
            lds[p1] = fr29_normalize(df);
        }
        __syncthreads();
        s = 1;
    }
    // One radix-4 butterfly per thread and pair (the launches use tile / 4 threads).  Butterfly b of pair s sits at the tile
    // positions idx_s(b) ^ {0, m, 2m, 3m}, idx_s(b) = ((b >> s) << (s + 2)) | (b & (m - 1)): a bit permutation of b.  From one pair
    // to the next only bits s, s+1 of b move (from positions s+2, s+3 down to s, s+1), and the swizzle is GF(2)-linear, so the
    // swizzled position is CARRIED: p0 ^= (bit s ? D1 : 0) ^ (bit s+1 ? D2 : 0) with two wave-uniform constants -- five VALU
    // instructions per pair instead of the ~45 of a fresh swz() (profiles/r05_isa_inventory_ntt.txt).  (Measured: 10 % fewer VALU
    // instructions per transform and the SAME kernel time -- these kernels are not issue-bound; profiles/r05_ab_ntt.txt.)
    const uint32_t b = threadIdx.x;
    const bool active = b < (vec << (t - 2));
    uint32_t p0 = swz(((b >> s) << (s + 2)) | (b & ((1u << s) - 1u)), sw);
    for (; s + 1 < t; s += 2) {
        const uint32_t m = 1u << s;
        const uint32_t d1 = swz_bit(s, sw), d2 = swz_bit(s + 1, sw);  // the swizzle is linear: swz(p ^ m) = swz(p) ^ swz(m)
        if (active) {
            const uint32_t j = b & (m - 1);
            const uint32_t p1 = p0 ^ d1, p2 = p0 ^ d2, p3 = p1 ^ d2;
            Fr29 x0 = lds[p0], x1 = lds[p1], x2 = lds[p2], x3 = lds[p3];
            Fr29 s0, y1, s2, y3, z0, z1, z2, z3;
            if (s != 0) {
                // stage s: twiddle w_{2m}^j on the odd halves; stage s+1: w_{4m}^j and w_{4m}^(j+m)
                const uint32_t ia = j << (t - 1 - s), ib = j << (t - 2 - s), ic = (j + m) << (t - 2 - s);
                const Fr29 a = tw.w[ia], ap = tw.wp[ia];
                mulshoup29x2(x1, a, ap, x3, a, ap);   // the two products that share a twiddle as ONE instruction stream, chains interleaved:
                                                      // two independent dependency chains in flight per wave (-2 % at every size)
                fr29_butterfly_lazy(x0, x1, s0, y1);
                fr29_butterfly_lazy(x2, x3, s2, y3);
                s2 = mulshoup29(s2, tw.w[ib], tw.wp[ib]);   // (these two as an interleaved pair as well: 94 live registers for the pair alone,
                y3 = mulshoup29(y3, tw.w[ic], tw.wp[ic]);   // the kernels spill at their 128 and lose 14 %: profiles/r05_ab_ntt.txt)
                fr29_butterfly_lazy(s0, s2, z0, z2);
                fr29_butterfly_lazy(y1, y3, z1, z3);
            } else {
                // first pair of a tile (raw inputs, j = 0): stage-0 twiddles are 1, stage 1 has 1 and w_4
                const uint32_t ic = m << (t - 2);
                fr29_butterfly_lazy(x0, x1, s0, y1);
                fr29_butterfly_lazy(x2, x3, s2, y3);
                s2 = fr29_normalize(s2);                        // below 4.5 r, unmultiplied: the 8r butterfly
                y3 = mulshoup29(y3, tw.w[ic], tw.wp[ic]);
                fr29_butterfly_lazy8(s0, s2, z0, z2);
                fr29_butterfly_lazy(y1, y3, z1, z3);
            }
            lds[p0] = fr29_normalize(z0);
            lds[p1] = fr29_normalize(z1);
            lds[p2] = fr29_normalize(z2);
            lds[p3] = fr29_normalize(z3);
        }
        // the next pair's position: bits s, s+1 of b move from positions s+2, s+3 down to s, s+1
        const uint32_t D1 = swz_bit(s + 2, sw) ^ d1, D2 = swz_bit(s + 3, sw) ^ d2;
        p0 ^= (((uint32_t)((int32_t)(b << (31 - s)) >> 31)) & D1) ^ (((uint32_t)((int32_t)(b << (30 - s)) >> 31)) & D2);
        __syncthreads();
    }
}

// The four tile positions a thread touches in a load / store phase: element e_i = threadIdx.x + i * blockDim.x (the launches use
// tile / 4 threads, a power of two).  Every position function of these kernels is a bit permutation of e, i.e. GF(2)-linear, and so
// is the swizzle: phys(e_i) = phys(threadIdx.x) ^ phys(i * blockDim.x), the second term wave-uniform (scalar ALU).  One swz() per
// thread and phase instead of four.
template <class IdxFn>
__device__ __forceinline__ void tile_positions4(IdxFn idx, uint64_t sw, uint32_t (&p)[4]) {
    p[0] = swz(idx(threadIdx.x), sw);
    // idx(blockDim.x) and idx(2 blockDim.x) are single bits (blockDim.x is a power of two and idx permutes bits)
    const uint32_t q1 = swz_bit(__builtin_ctz(idx(blockDim.x)), sw), q2 = swz_bit(__builtin_ctz(idx(2 * blockDim.x)), sw);
    p[1] = p[0] ^ q1;
    p[2] = p[0] ^ q2;
    p[3] = p[1] ^ q2;
}

// Whole transform in one tile (log_n <= 12).
__global__ __launch_bounds__(1024) void k_ntt_single(Fr *data, uint32_t t, const Tw29 tw, Fr29 scale, Fr29 scale_p, uint64_t sw) {
    const uint32_t n = 1u << t;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) lds_fr29[swz(bitrev(i, t), sw)] = fr29_unpack(data[i]);
    __syncthreads();
    if (t == 1) {
        if (threadIdx.x == 0) {
            Fr29 u = lds_fr29[0], w = lds_fr29[1], sm, df;
            fr29_butterfly_lazy(u, w, sm, df);
            lds_fr29[0] = fr29_normalize(sm);
            lds_fr29[1] = fr29_normalize(df);
        }
        __syncthreads();
    } else if (t >= 2) {
        lds_ntt_stages29(lds_fr29, t, 1, tw, sw);
    }
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) data[i] = fr29_pack_canonical(mulshoup29(lds_fr29[swz(i, sw)], scale, scale_p));
}

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one, MI355X_MICROARCH.md), each with its own L2.  The tile
// a block takes is therefore chosen so that the blocks of ONE XCD take ADJACENT tiles: in pass 1 their 128-byte column segments then
// make up runs of a row in one L2 (4 KB per row and XCD at 2^20 instead of 128 B in each of eight L2s): pass 1 -6 % at 2^20, -13 % at
// 2^21, -8 % at 2^24 (profiles/r05_ntt_xcd_probe.txt).  In pass 2 the same order concentrates an XCD's transposed 64-byte stores on
// one 2 KB column range of every 32 KB row (+5 % at 2^20-2^22: channel camping), so pass 2 keeps the plain order except at 2^24
// (keeping only the PAIRS of tiles whose 64-byte halves make up a line on one XCD was measured too: no better than the plain order).
__device__ __forceinline__ uint32_t xcd_tile(uint32_t b, uint32_t g, int on) { return (!on || (g & 7u)) ? b : (b & 7u) * (g >> 3) + (b >> 3); }

// pass 1: columns j2 = tile*vec .. +vec-1; element (j1, j2) at in[j1*n2 + j2]
__global__ __launch_bounds__(1024) void k_ntt_pass1(const Fr *in, Fr *out, uint32_t k1, uint32_t k2, uint32_t vec_log,
                                                    const Tw29 tw1, const Tw29 tw_lo, const Tw29 tw_hi, uint32_t lo_bits,
                                                    const Fr29 *tw_full, uint64_t sw, int xcd) {
    const uint32_t vec = 1u << vec_log;
    const uint32_t j2_0 = xcd_tile(blockIdx.x, gridDim.x, xcd) << vec_log;
    // every thread moves exactly four elements (the launch uses total / 4 threads): all four loads are issued before the first
    // is unpacked, so a wave waits for HBM once, not four times
    {
        Fr raw[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t e = threadIdx.x + i * blockDim.x;
            raw[i] = in[((size_t)(e >> vec_log) << k2) + j2_0 + (e & (vec - 1))];  // consecutive threads -> consecutive columns
        }
        uint32_t pos[4];
        tile_positions4([&](uint32_t e) { return ((e & (vec - 1)) << k1) | bitrev(e >> vec_log, k1); }, sw, pos);
#pragma unroll
        for (int i = 0; i < 4; i++) lds_fr29[pos[i]] = fr29_unpack(raw[i]);
    }
    __syncthreads();
    lds_ntt_stages29(lds_fr29, k1, vec, tw1, sw);
    const uint32_t lo_mask = (1u << lo_bits) - 1;
    Fr29 twf[4];
    if (tw_full) {  // the four table entries of this thread, requested together
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t e = threadIdx.x + i * blockDim.x;
            twf[i] = tw_full[((size_t)(e >> vec_log) << k2) + j2_0 + (e & (vec - 1))];
        }
    }
    uint32_t opos[4];
    tile_positions4([&](uint32_t e) { return ((e & (vec - 1)) << k1) | (e >> vec_log); }, sw, opos);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t e = threadIdx.x + i * blockDim.x;
        uint32_t v = e & (vec - 1), kk1 = e >> vec_log;
        uint32_t j2 = j2_0 + v;
        Fr29 x = lds_fr29[opos[i]];
        if (tw_full) {
            x = mul29r(x, twf[i]);  // Montgomery-29 product: below 1.4 r for x < 64 r (as two interleaved pairs: no gain, measured)
        } else {
            const uint64_t ex = (uint64_t)j2 * kk1;  // < n
            const size_t ih = ex >> lo_bits, il = ex & lo_mask;
            x = mulshoup29(x, tw_hi.w[ih], tw_hi.wp[ih]);
            x = mulshoup29(x, tw_lo.w[il], tw_lo.wp[il]);
        }
        // below 2r < 2^256: the scratch keeps this representative (pass 2 unpacks any 256-bit integer); no canonicalisation
        out[((size_t)kk1 << k2) + j2] = fr29_pack_raw(x);
    }
}

// pass 2: rows k1 = blockIdx.x*vec .. +vec-1; row k1 contiguous at in[k1*n2 ..]; out[k1 + n1*k2]
// SHORT: the transform's input has at most n2 non-zero leading elements (a zero-padded short polynomial: Z = prod (X - x_i) on the
// evaluation coset of create_witness_batched).  Only row 0 of the n1 x n2 matrix is non-zero, so every column transform of pass 1
// returns its one input in all n1 positions and pass 1 collapses to the inter-pass twiddle: row k1 of pass 2's input is
// x[j2] * w_n^(j2 k1) * scale for j2 < nnz and zero beyond.  The tile load computes that from the nnz inputs (`in` = a private copy
// of them) and pass 1 is not launched at all.
template <bool SHORT>
__global__ __launch_bounds__(1024) void k_ntt_pass2(const Fr *in, Fr *out, uint32_t k1, uint32_t k2, uint32_t vec_log,
                                                    const Tw29 tw2, Fr29 scale, Fr29 scale_p, int scale_folded, uint64_t sw,
                                                    uint32_t nnz, const Fr29 *tw_full, const Tw29 tw_lo, const Tw29 tw_hi, uint32_t lo_bits, int xcd) {
    const uint32_t n2 = 1u << k2, vec = 1u << vec_log;
    const uint32_t r0 = xcd_tile(blockIdx.x, gridDim.x, xcd) << vec_log;
    if (!SHORT) {
        Fr raw[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t e = threadIdx.x + i * blockDim.x;
            raw[i] = in[((size_t)(r0 + (e >> k2)) << k2) + (e & (n2 - 1))];  // consecutive threads -> consecutive row elements
        }
        uint32_t pos[4];
        tile_positions4([&](uint32_t e) { return ((e >> k2) << k2) | bitrev(e & (n2 - 1), k2); }, sw, pos);
#pragma unroll
        for (int i = 0; i < 4; i++) lds_fr29[pos[i]] = fr29_unpack(raw[i]);
    } else {
        const uint32_t lo_mask = (1u << lo_bits) - 1;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t e = threadIdx.x + i * blockDim.x;
            const uint32_t j2 = e & (n2 - 1), row = r0 + (e >> k2);
            Fr29 x;
#pragma unroll
            for (int l = 0; l < R29_N; l++) x.v[l] = 0;
            if (j2 < nnz) {
                x = fr29_unpack(in[j2]);
                if (tw_full) {
                    x = mul29r(x, tw_full[((size_t)row << k2) + j2]);
                } else {
                    const uint64_t ex = (uint64_t)j2 * row;
                    const size_t ih = ex >> lo_bits, il = ex & lo_mask;
                    x = mulshoup29(x, tw_hi.w[ih], tw_hi.wp[ih]);
                    x = mulshoup29(x, tw_lo.w[il], tw_lo.wp[il]);
                }
            }
            lds_fr29[swz(((e >> k2) << k2) | bitrev(j2, k2), sw)] = x;
        }
    }
    __syncthreads();
    lds_ntt_stages29(lds_fr29, k2, vec, tw2, sw);
    uint32_t opos[4];
    tile_positions4([&](uint32_t e) { return ((e & (vec - 1)) << k2) | (e >> vec_log); }, sw, opos);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t e = threadIdx.x + i * blockDim.x;
        uint32_t v = e & (vec - 1), kk2 = e >> vec_log;  // consecutive threads -> consecutive k1
        Fr29 x = lds_fr29[opos[i]];
        // the scale already sits in the inter-pass table: only canonicalise (x < 64 r) -- no multiplication
        x = scale_folded ? fr29_reduce_below_2r(x) : mulshoup29(x, scale, scale_p);
        out[((size_t)kk2 << k1) + r0 + v] = fr29_pack_canonical(x);
    }
}

// ---- round 6: the same transform with the tile load fused into the first stage pair and the store into the last ----------------
// k_ntt_pass1 / k_ntt_pass2 above move a tile in three lock-stepped phases: load -> LDS, barrier, five stage pairs (LDS -> registers ->
// LDS, a barrier each), barrier, LDS -> store.  A thread of the load phase already holds exactly the four inputs of one first-pair
// butterfly (rows r, r + n/4, r + n/2, r + 3n/4 of its column), and a thread of the store phase writes exactly the four outputs of one
// last-pair butterfly (k, k + n/4, k + n/2, k + 3n/4): so the first pair runs on the loaded registers and the last pair's results go
// straight to global memory.  Two LDS round trips and two barriers fewer per pass (4 instead of 6 at 2^10 points), and the memory
// phases stop being block-wide: a wave starts its first butterfly when ITS loads have landed and stores while others still compute.
// BPT = 2: two butterflies per thread at half the threads (two waves per SIMD, 256 VGPRs): the middle pairs' twiddles are shared by
// the two butterflies (half the twiddle loads) and every product has an independent partner for the interleaved two-chain stream.
struct NttTile {
    uint32_t t;        // log2 of the transform length of this pass
    uint32_t kother;   // column pass (PASS 1): log2 of the column stride = of the columns per batch; row pass (PASS 2): log2 of the
                       // output stride of the transform index (k1 bits, plus the middle bits of a three-pass transform)
    uint32_t vec_log;  // 2^vec_log columns (pass 1) / rows (pass 2) per tile
    // Three-pass transforms (n = n1 n2 n3, log n >= 22; ntt_run3).  Column pass over a BATCH of matrices: tile -> (batch = tile >>
    // tpb_log, column group = the low bits), batch b starts at b << (t + kother).  Row pass over gathered rows: tile -> (k2 = tile >>
    // g1_log, k1 group = the low bits), row (k1, k2) is read at ((k1 << mid_log) + k2) << t and written at
    // (k3 << kother) + (k2 << (kother - mid_log)) + k1.  Two-pass transforms: tpb_log = g1_log = 31, mid_log = 0.
    uint32_t tpb_log = 31, g1_log = 31, mid_log = 0;
    uint32_t lo_bits, nnz;
    int xcd, scale_folded;
    uint64_t sw;
    Tw29 tw, tw_lo, tw_hi;
    const Fr29 *tw_full;
    Fr29 scale, scale_p;
};

template <int PASS, bool SHORT>
__device__ __forceinline__ Fr29 ntt_tile_input(const Fr *in, const NttTile &a, uint32_t tile0, uint32_t vv, uint32_t pos) {
    // pos: index along the transform (row j1 in pass 1, element j2 of the row in pass 2); vv: column / row inside the tile
    if (PASS == 1) return fr29_unpack(in[((size_t)pos << a.kother) + tile0 + vv]);
    if (!SHORT) return fr29_unpack(in[((size_t)(tile0 + vv) << a.t) + pos]);
    Fr29 x;
#pragma unroll
    for (int l = 0; l < R29_N; l++) x.v[l] = 0;
    if (pos < a.nnz) {  // zero-padded short input: pass 1 collapsed to the inter-pass twiddle (see k_ntt_pass2<SHORT>)
        const uint32_t row = tile0 + vv;
        x = fr29_unpack(in[pos]);
        if (a.tw_full) {
            x = mul29r(x, a.tw_full[((size_t)row << a.t) + pos]);
        } else {
            const uint64_t ex = (uint64_t)pos * row;
            const size_t ih = ex >> a.lo_bits, il = ex & ((1u << a.lo_bits) - 1);
            x = mulshoup29(x, a.tw_hi.w[ih], a.tw_hi.wp[ih]);
            x = mulshoup29(x, a.tw_lo.w[il], a.tw_lo.wp[il]);
        }
    }
    return x;
}

template <int PASS, int BPT, bool SHORT>
__global__ __launch_bounds__(1024 / BPT) void k_ntt_tile(const Fr *in, Fr *out, const NttTile a) {
    Fr29 *lds = lds_fr29;
    const uint32_t t = a.t, vl = a.vec_log, vec = 1u << vl, Q = 1u << (t - 2);
    const uint32_t nthreads = blockDim.x;  // = (vec << t) / (4 BPT)
    const uint32_t tile = xcd_tile(blockIdx.x, gridDim.x, a.xcd);
    // column pass: first column of the tile inside its batch, and the batch's base; row pass: first row (k1) of the tile and its k2
    const uint32_t tile0 = (PASS == 1 ? (tile & ((1u << a.tpb_log) - 1u)) : (tile & ((1u << a.g1_log) - 1u))) << vl;
    const uint32_t hi_id = PASS == 1 ? (tile >> a.tpb_log) : (tile >> a.g1_log);   // batch (pass 1) / k2 (pass 2); 0 for two-pass transforms
    const size_t in_base = PASS == 1 ? ((size_t)hi_id << (t + a.kother)) : ((size_t)hi_id << t);
    const size_t out_base = PASS == 1 ? in_base : ((size_t)hi_id << (a.kother - a.mid_log));
    const uint32_t row_shift = t + a.mid_log;   // row pass: rows of a tile are 2^row_shift elements apart
    const uint64_t sw = a.sw;
    const Tw29 tw = a.tw;
    // ---- first pair (or the radix-2 stage of an odd length), on the loaded registers --------------------------------------------
    {
        Fr29 e[BPT][4];
        uint32_t p[BPT];
#pragma unroll
        for (int u = 0; u < BPT; u++) {
            const uint32_t g = threadIdx.x + u * nthreads;
            // pass 1: consecutive threads -> consecutive columns, then rows; pass 2: consecutive threads -> consecutive row elements
            const uint32_t vv = PASS == 1 ? (g & (vec - 1)) : (g >> (t - 2)), q = PASS == 1 ? (g >> vl) : (g & (Q - 1));
            if (PASS == 1 || !SHORT) {  // all loads of a thread in flight before the first unpack
                Fr raw[4];
#pragma unroll
                for (int i = 0; i < 4; i++)
                    raw[i] = PASS == 1 ? in[in_base + ((size_t)(q + i * Q) << a.kother) + tile0 + vv]
                                       : in[in_base + ((size_t)(tile0 + vv) << row_shift) + q + i * Q];
#pragma unroll
                for (int i = 0; i < 4; i++) e[u][i] = fr29_unpack(raw[i]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) e[u][i] = ntt_tile_input<PASS, SHORT>(in, a, tile0, vv, q + i * Q);
            }
            p[u] = swz((vv << t) | (bitrev(q, t - 2) << 2), sw);  // positions p ^ {0,1,2,3} hold rows q, q + 2Q, q + Q, q + 3Q
        }
        if (t & 1) {
#pragma unroll
            for (int u = 0; u < BPT; u++) {
                Fr29 s0, d0, s1, d1;
                fr29_butterfly_lazy(e[u][0], e[u][2], s0, d0);
                fr29_butterfly_lazy(e[u][1], e[u][3], s1, d1);
                lds[p[u]] = fr29_normalize(s0);
                lds[p[u] ^ 1] = fr29_normalize(d0);
                lds[p[u] ^ 2] = fr29_normalize(s1);
                lds[p[u] ^ 3] = fr29_normalize(d1);
            }
        } else {
            const uint32_t ic = 1u << (t - 2);
            const Fr29 c = tw.w[ic], cp = tw.wp[ic];
            Fr29 s0[BPT], y1[BPT], s2[BPT], y3[BPT];
#pragma unroll
            for (int u = 0; u < BPT; u++) {
                fr29_butterfly_lazy(e[u][0], e[u][2], s0[u], y1[u]);
                fr29_butterfly_lazy(e[u][1], e[u][3], s2[u], y3[u]);
                s2[u] = fr29_normalize(s2[u]);
            }
            if constexpr (BPT == 2) mulshoup29x2(y3[0], c, cp, y3[1], c, cp);
            else y3[0] = mulshoup29(y3[0], c, cp);
#pragma unroll
            for (int u = 0; u < BPT; u++) {
                Fr29 z0, z1, z2, z3;
                fr29_butterfly_lazy8(s0[u], s2[u], z0, z2);
                fr29_butterfly_lazy(y1[u], y3[u], z1, z3);
                lds[p[u]] = fr29_normalize(z0);
                lds[p[u] ^ 1] = fr29_normalize(z1);
                lds[p[u] ^ 2] = fr29_normalize(z2);
                lds[p[u] ^ 3] = fr29_normalize(z3);
            }
        }
    }
    __syncthreads();
    // ---- middle pairs: LDS -> registers -> LDS (lds_ntt_stages29's pair; two butterflies of a thread differ in bit h of b, above s) ----
    uint32_t s = (t & 1) ? 1 : 2;
    const uint32_t b = threadIdx.x;
    const uint32_t h = __builtin_ctz(nthreads);
    uint32_t p0 = swz(((b >> s) << (s + 2)) | (b & ((1u << s) - 1u)), sw);
    for (; s + 2 < t; s += 2) {
        const uint32_t m = 1u << s;
        const uint32_t d1 = swz_bit(s, sw), d2 = swz_bit(s + 1, sw), dh = swz_bit(h + 2, sw);
        const uint32_t j = b & (m - 1);
        const uint32_t ia = j << (t - 1 - s), ib = j << (t - 2 - s), ic = (j + m) << (t - 2 - s);
        const Fr29 wa = tw.w[ia], wap = tw.wp[ia];
        Fr29 x0[BPT], x1[BPT], x2[BPT], x3[BPT];
#pragma unroll
        for (int u = 0; u < BPT; u++) {
            const uint32_t pu = p0 ^ (u ? dh : 0u);
            x0[u] = lds[pu];
            x1[u] = lds[pu ^ d1];
            x2[u] = lds[pu ^ d2];
            x3[u] = lds[pu ^ d1 ^ d2];
        }
        const Fr29 wb = tw.w[ib], wbp = tw.wp[ib], wc = tw.w[ic], wcp = tw.wp[ic];
        if constexpr (BPT == 2) {
            mulshoup29x2(x1[0], wa, wap, x1[1], wa, wap);
            mulshoup29x2(x3[0], wa, wap, x3[1], wa, wap);
        } else {
            mulshoup29x2(x1[0], wa, wap, x3[0], wa, wap);
        }
        Fr29 s0[BPT], y1[BPT], s2[BPT], y3[BPT];
#pragma unroll
        for (int u = 0; u < BPT; u++) {
            fr29_butterfly_lazy(x0[u], x1[u], s0[u], y1[u]);
            fr29_butterfly_lazy(x2[u], x3[u], s2[u], y3[u]);
        }
        if constexpr (BPT == 2) {
            mulshoup29x2(s2[0], wb, wbp, s2[1], wb, wbp);
            mulshoup29x2(y3[0], wc, wcp, y3[1], wc, wcp);
        } else {
            s2[0] = mulshoup29(s2[0], wb, wbp);
            y3[0] = mulshoup29(y3[0], wc, wcp);
        }
#pragma unroll
        for (int u = 0; u < BPT; u++) {
            const uint32_t pu = p0 ^ (u ? dh : 0u);
            Fr29 z0, z1, z2, z3;
            fr29_butterfly_lazy(s0[u], s2[u], z0, z2);
            fr29_butterfly_lazy(y1[u], y3[u], z1, z3);
            lds[pu] = fr29_normalize(z0);
            lds[pu ^ d1] = fr29_normalize(z1);
            lds[pu ^ d2] = fr29_normalize(z2);
            lds[pu ^ d1 ^ d2] = fr29_normalize(z3);
        }
        const uint32_t D1 = swz_bit(s + 2, sw) ^ d1, D2 = swz_bit(s + 3, sw) ^ d2;
        p0 ^= (((uint32_t)((int32_t)(b << (31 - s)) >> 31)) & D1) ^ (((uint32_t)((int32_t)(b << (30 - s)) >> 31)) & D2);
        __syncthreads();
    }
    // ---- last pair (s = t - 2): results multiplied / reduced and stored from the registers ------------------------------------------
    {
        const uint32_t d1 = swz_bit(t - 2, sw), d2 = swz_bit(t - 1, sw);
        const uint32_t lo_mask = (1u << a.lo_bits) - 1;
#pragma unroll
        for (int u = 0; u < BPT; u++) {
            const uint32_t g = threadIdx.x + u * nthreads;
            const uint32_t vv = g & (vec - 1), j = g >> vl;  // consecutive threads -> consecutive columns (pass 1) / rows (pass 2)
            const uint32_t pu = swz((vv << t) | j, sw);
            // the four inter-pass table entries of this butterfly: requested before the LDS reads where the registers allow it (BPT = 2,
            // 256 VGPRs), one ahead of its use behind the butterflies at four waves per SIMD (128 VGPRs: requested together they spill 8-21 registers)
            Fr29 twf[4];
            if (BPT == 2 && PASS == 1 && a.tw_full) {
#pragma unroll
                for (int i = 0; i < 4; i++) twf[i] = a.tw_full[((size_t)(j + i * Q) << a.kother) + tile0 + vv];
            }
            Fr29 x0 = lds[pu], x1 = lds[pu ^ d1], x2 = lds[pu ^ d2], x3 = lds[pu ^ d1 ^ d2];
            Fr29 s0, y1, s2, y3, z[4];
            {
                const uint32_t ia = j << 1, ib = j, ic = j + Q;
                const Fr29 wa = tw.w[ia], wap = tw.wp[ia];
                mulshoup29x2(x1, wa, wap, x3, wa, wap);
                fr29_butterfly_lazy(x0, x1, s0, y1);
                fr29_butterfly_lazy(x2, x3, s2, y3);
                if constexpr (BPT == 2) {
                    mulshoup29x2(s2, tw.w[ib], tw.wp[ib], y3, tw.w[ic], tw.wp[ic]);
                } else {
                    s2 = mulshoup29(s2, tw.w[ib], tw.wp[ib]);
                    y3 = mulshoup29(y3, tw.w[ic], tw.wp[ic]);
                }
                fr29_butterfly_lazy(s0, s2, z[0], z[2]);
                fr29_butterfly_lazy(y1, y3, z[1], z[3]);
            }
            // BPT = 1: the table entry of element i + 1 is requested before the product of element i (one entry in flight beside the one in
            // use: 126 VGPRs, no spill; pass 1 -1..2 % at 2^20 / 2^21, profiles/r06_ab_ntt.txt)
            Fr29 twn;
            if (BPT == 1 && PASS == 1 && a.tw_full) twn = a.tw_full[((size_t)j << a.kother) + tile0 + vv];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t kk = j + i * Q;
                Fr29 x = fr29_normalize(z[i]);
                if (PASS == 1) {
                    const uint32_t j2 = tile0 + vv;
                    if (a.tw_full) {
                        const Fr29 twc = twn;
                        if (BPT == 1 && i < 3) twn = a.tw_full[((size_t)(kk + Q) << a.kother) + j2];
                        x = mul29r(x, BPT == 2 ? twf[i] : twc);
                    } else {
                        const uint64_t ex = (uint64_t)j2 * kk;
                        const size_t ih = ex >> a.lo_bits, il = ex & lo_mask;
                        x = mulshoup29(x, a.tw_hi.w[ih], a.tw_hi.wp[ih]);
                        x = mulshoup29(x, a.tw_lo.w[il], a.tw_lo.wp[il]);
                    }
                    out[out_base + ((size_t)kk << a.kother) + j2] = fr29_pack_raw(x);
                } else {
                    x = a.scale_folded ? fr29_reduce_below_2r(x) : mulshoup29(x, a.scale, a.scale_p);
                    out[out_base + ((size_t)kk << a.kother) + tile0 + vv] = fr29_pack_canonical(x);
                }
            }
        }
    }
}

static int pow_table29(kzg_ctx *ctx, hipStream_t st, const Fr &base, size_t count, Tw29 *out) {
    if (!count) count = 1;
    KZG_HIP_CHECK(ctx, hipMalloc((void **)&out->w, count * sizeof(Fr29)));
    KZG_HIP_CHECK(ctx, hipMalloc((void **)&out->wp, count * sizeof(Fr29)));
    KZG_LAUNCH(ctx, st, "k_pow_table29", k_pow_table29, (unsigned)((count + 255) / 256), 256, 0, base, count, out->w, out->wp);
    return KZG_OK;
}

static void plan_free(NttPlan *p) {
    for (Tw29 *t : {&p->tw1, &p->tw2, &p->tw3, &p->tw_lo, &p->tw_hi}) {
        if (t->w) hipFree(t->w);
        if (t->wp) hipFree(t->wp);
    }
    if (p->tw_full) hipFree(p->tw_full);
    delete p;
}

static int ntt_plan_build(kzg_ctx *ctx, hipStream_t st, NttPlan *p) {
    const uint32_t log_n = p->log_n;
    Fr w = host_omega(log_n);
    if (p->inverse) w = inv(w);
    size_t n = (size_t)1 << log_n;
    Fr scale = p->inverse ? inv(from_u64<FrParams>((uint64_t)n)) : Fr::one();
    fr29_shoup_from_twiddle(fr29_twiddle_from_mont(scale), p->scale, p->scale_p);
    if (log_n <= 12) {
        p->k1 = log_n;
        p->k2 = 0;
        return pow_table29(ctx, st, w, log_n ? (n >> 1) : 1, &p->tw1);
    }
    p->k1 = (log_n + 1) / 2;
    p->k2 = log_n - p->k1;
    size_t n1 = (size_t)1 << p->k1, n2 = (size_t)1 << p->k2;
    p->lo_bits = p->k1;
    size_t nlo = (size_t)1 << p->lo_bits, nhi = (size_t)1 << (log_n - p->lo_bits);
    KZG_TRY(pow_table29(ctx, st, pow_u64(w, (uint64_t)n2), n1 >> 1, &p->tw1));  // w_{n1}
    KZG_TRY(pow_table29(ctx, st, pow_u64(w, (uint64_t)n1), n2 >> 1, &p->tw2));  // w_{n2}
    KZG_TRY(pow_table29(ctx, st, w, nlo, &p->tw_lo));
    KZG_TRY(pow_table29(ctx, st, pow_u64(w, (uint64_t)nlo), nhi, &p->tw_hi));
    if (log_n <= 21) {  // 36 B per element, 38 MB at 2^20 per direction; measured -7.5 % at 2^20, nothing at 2^22
        KZG_HIP_CHECK(ctx, hipMalloc((void **)&p->tw_full, n * sizeof(Fr29)));
        KZG_LAUNCH(ctx, st, "k_twiddle_full", k_twiddle_full, (unsigned)((n + 255) / 256), 256, 0, p->tw_lo, p->tw_hi, p->lo_bits, p->k2,
                   scale, n, p->tw_full);
    }
    return KZG_OK;
}

static int ntt_plan3_build(kzg_ctx *ctx, hipStream_t st, NttPlan *p);
static int ntt_plan(kzg_ctx *ctx, hipStream_t st, uint32_t log_n, int inverse, NttPlan **out, bool three = false) {
    uint32_t key = log_n * 2 + (inverse ? 1 : 0) + (three ? 1000 : 0);
    // leased lanes (concurrent fft / create_witness_batched / verify_poly callers) share the plans: one builder at a time, and a
    // plan is published only after the stream that filled its tables has been synchronised
    std::lock_guard<std::mutex> clk(ctx->cache_mu);
    auto it = ctx->ntt_plans.find(key);
    if (it != ctx->ntt_plans.end()) {
        *out = it->second;
        return KZG_OK;
    }
    if (!ctx->attr_ntt_set) {
        KZG_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)k_ntt_single, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096));
        KZG_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)k_ntt_pass1, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096));
        KZG_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)k_ntt_pass2<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096));
        KZG_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)k_ntt_pass2<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096));
        for (const void *k : {(const void *)k_ntt_tile<1, 1, false>, (const void *)k_ntt_tile<1, 2, false>, (const void *)k_ntt_tile<2, 1, false>,
                              (const void *)k_ntt_tile<2, 2, false>, (const void *)k_ntt_tile<2, 1, true>, (const void *)k_ntt_tile<2, 2, true>})
            KZG_HIP_CHECK(ctx, hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096));
        ctx->attr_ntt_set = true;
    }
    NttPlan *p = new NttPlan();
    p->log_n = log_n;
    p->inverse = inverse;
    int rc = three ? ntt_plan3_build(ctx, st, p) : ntt_plan_build(ctx, st, p);
    if (rc == KZG_OK && hipStreamSynchronize(st) != hipSuccess) rc = fail(ctx, KZG_ERR_HIP, "NTT twiddle tables");
    if (rc != KZG_OK) {  // nothing half-built stays behind
        hipStreamSynchronize(st);
        plan_free(p);
        return rc;
    }
    ctx->ntt_plans[key] = p;
    *out = p;
    return KZG_OK;
}

// ---- three-pass transforms (round 6): n = 2^8 * n2 * n3 ------------------------------------------------------------------------
// At 2^22 .. 2^24 the two-pass split leaves pass 1 with tiles of 2, 1 and 1 columns of 2^11 / 2^12 points: 64- and 32-byte segments
// 64 KB / 128 KB apart in both its loads and its stores (pass 1 1.14 ms against pass 2's 0.89 at 2^24), and above 2^21 the inter-pass
// twiddle is a two-level product (two multiplications).  Three passes of at most 2^8 points have 16 or 32 vectors per tile (512- / 1024-
// byte segments everywhere), the same number of multiplications per element (three passes of 4 - 1 stage products, two + one inter-pass
// products: 12, as the two-pass transform with its two-level product), and pay one more trip through HBM and one more unpack / pack:
//   A  for every column c = (j2, j3) of the 2^8 x 2^(k2 + k3) matrix: 2^8-point transform over stride-2^(k2 + k3) elements, times
//      w_n^(c k1) (two-level table), data -> scratch in the same layout              (k_ntt_tile<1>, 16 columns per tile)
//   B  for every k1 and every column j3 of its 2^k2 x 2^k3 matrix: 2^k2-point transform over stride-2^k3 elements, times
//      w'^(j3 k2) * scale (one full table of 2^(k2 + k3) entries, w' = w_n^(2^8)), in place in scratch   (k_ntt_tile<1>, batched)
//   C  for every (k1, k2): 2^k3-point transform of the contiguous row, written to X[k1 + 2^8 k2 + 2^(8 + k2) k3]: a tile takes 16 / 32
//      consecutive k1 of one k2, so its stores are 512- / 1024-byte segments           (k_ntt_tile<2>, gathered rows)
static int ntt_plan3_build(kzg_ctx *ctx, hipStream_t st, NttPlan *p) {
    const uint32_t log_n = p->log_n;
    Fr w = host_omega(log_n);
    if (p->inverse) w = inv(w);
    const size_t n = (size_t)1 << log_n;
    const Fr scale = p->inverse ? inv(from_u64<FrParams>((uint64_t)n)) : Fr::one();
    fr29_shoup_from_twiddle(fr29_twiddle_from_mont(scale), p->scale, p->scale_p);
    p->k1 = 8;
    const uint32_t inner = log_n - 8;
    p->k2 = (inner + 1) / 2;
    p->k3 = inner - p->k2;
    const size_t n1 = 256, n2 = (size_t)1 << p->k2, n3 = (size_t)1 << p->k3, ni = (size_t)1 << inner;
    p->lo_bits = 12;
    KZG_TRY(pow_table29(ctx, st, pow_u64(w, (uint64_t)ni), n1 >> 1, &p->tw1));        // w_{n1} = w^(n / n1)
    KZG_TRY(pow_table29(ctx, st, pow_u64(w, (uint64_t)(n1 * n3)), n2 >> 1, &p->tw2));  // w_{n2} = w'^(n3)
    KZG_TRY(pow_table29(ctx, st, pow_u64(w, (uint64_t)(n1 * n2)), n3 >> 1, &p->tw3));  // w_{n3} = w'^(n2)
    KZG_TRY(pow_table29(ctx, st, w, (size_t)1 << p->lo_bits, &p->tw_lo));
    KZG_TRY(pow_table29(ctx, st, pow_u64(w, 1ull << p->lo_bits), n >> p->lo_bits, &p->tw_hi));
    // the inner inter-pass table from a two-level table of w' of its own (freed below)
    Tw29 ilo, ihi;
    const Fr wi = pow_u64(w, (uint64_t)n1);
    const uint32_t ilo_bits = p->k2;
    int rc = pow_table29(ctx, st, wi, (size_t)1 << ilo_bits, &ilo);
    if (rc == KZG_OK) rc = pow_table29(ctx, st, pow_u64(wi, 1ull << ilo_bits), ni >> ilo_bits, &ihi);
    if (rc == KZG_OK && hipMalloc((void **)&p->tw_full, ni * sizeof(Fr29)) != hipSuccess) rc = fail(ctx, KZG_ERR_ALLOC, "hipMalloc(NTT inter-pass table)");
    if (rc == KZG_OK) {
        KZG_LAUNCH(ctx, st, "k_twiddle_full", k_twiddle_full, (unsigned)((ni + 255) / 256), 256, 0, ilo, ihi, ilo_bits, p->k3, scale, ni, p->tw_full);
        if (hipStreamSynchronize(st) != hipSuccess) rc = fail(ctx, KZG_ERR_HIP, "NTT twiddle tables");
    }
    for (Tw29 *t : {&ilo, &ihi}) {
        if (t->w) hipFree(t->w);
        if (t->wp) hipFree(t->wp);
    }
    return rc;
}

static int ntt_run3(kzg_ctx *ctx, int lane, Fr *d_data, uint32_t log_n, int inverse) {
    hipStream_t st = ctx->lanes[lane].stream;
    NttPlan *p = nullptr;
    KZG_TRY(ntt_plan(ctx, st, log_n, inverse, &p, true));
    const size_t n = (size_t)1 << log_n;
    Fr *scratch = (Fr *)lane_alloc(ctx, lane, n * sizeof(Fr));
    if (!scratch) return fail(ctx, KZG_ERR_ALLOC, "NTT workspace not reserved");
    const uint32_t inner = p->k2 + p->k3;
    NttTile a, b, c;
    a.t = 8; a.kother = inner; a.vec_log = 4; a.lo_bits = p->lo_bits; a.nnz = 0; a.xcd = ctx->opt_ntt_xcd & 1; a.scale_folded = 0;
    a.sw = LDS_SWIZZLE[8][4]; a.tw = p->tw1; a.tw_lo = p->tw_lo; a.tw_hi = p->tw_hi; a.tw_full = nullptr; a.scale = p->scale; a.scale_p = p->scale_p;
    b = a;
    b.t = p->k2; b.kother = p->k3; b.vec_log = 12 - p->k2; b.tpb_log = p->k3 - b.vec_log; b.xcd = 0; b.sw = LDS_SWIZZLE[p->k2][b.vec_log];
    b.tw = p->tw2; b.tw_full = p->tw_full;
    c = a;
    c.t = p->k3; c.vec_log = 12 - p->k3; c.g1_log = 8 - c.vec_log; c.mid_log = p->k2; c.kother = 8 + p->k2; c.xcd = 0;
    c.sw = LDS_SWIZZLE[p->k3][c.vec_log]; c.tw = p->tw3; c.tw_full = nullptr; c.scale_folded = 1;
    const size_t lds = (size_t)4096 * sizeof(Fr29);
    const unsigned grid = (unsigned)(n >> 12);
    KZG_LAUNCH(ctx, st, "k_ntt_pass1", (k_ntt_tile<1, 1, false>), grid, 1024, lds, d_data, scratch, a);
    KZG_LAUNCH(ctx, st, "k_ntt_pass1b", (k_ntt_tile<1, 1, false>), grid, 1024, lds, scratch, scratch, b);
    KZG_LAUNCH(ctx, st, "k_ntt_pass2", (k_ntt_tile<2, 1, false>), grid, 1024, lds, scratch, d_data, c);
    return KZG_OK;
}

void ntt_plans_free(kzg_ctx *ctx) {
    for (auto &kv : ctx->ntt_plans) plan_free(kv.second);
    ctx->ntt_plans.clear();
}

// ---- sizes above 2^24: one more four-step level around the two-pass transform -------------------------------------------
// n = A * B with B = 2^24 and A = 2^(log_n - 24) <= 16:  X[k1 + A k2] = sum_j2 w_B^(j2 k2) w_n^(j2 k1) (sum_j1 x[j1 B + j2] w_A^(j1 k1)).
//   k_ntt_outer: thread j2 takes its A elements (stride B), an A-point transform in registers (w_A = w_n^B), times w_n^(j2 k1)
//                (and 1/A for the inverse), back in place at [k1 B + j2];
//   A two-pass transforms of the contiguous rows [k1 B, (k1 + 1) B)  (ntt_run, which scales by 1/B for the inverse);
//   k_ntt_outer_transpose: [k1 B + k2] -> [k1 + A k2]  (out of place, copied back).
// Plain saturated Fr arithmetic: this level does A + log A multiplications per element against the ~10 of the inner transform.
template <int LOGA>
__global__ __launch_bounds__(256) void k_ntt_outer(Fr *data, size_t B, const Fr *pw_lo, const Fr *pw_hi, uint32_t lo_bits, Fr wA, Fr scale) {
    constexpr int A = 1 << LOGA;
    const size_t j2 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j2 >= B) return;
    Fr x[A];
#pragma unroll
    for (int j1 = 0; j1 < A; j1++) x[j1] = data[(size_t)j1 * B + j2];
    // bit-reversal + DIT stages, twiddles w_A^(j << (LOGA - 1 - s)) by repeated multiplication
    Fr y[A];
#pragma unroll
    for (int i = 0; i < A; i++) {
        int r = 0;
#pragma unroll
        for (int b = 0; b < LOGA; b++) r |= ((i >> b) & 1) << (LOGA - 1 - b);
        y[r] = x[i];
    }
    Fr wp[A / 2 > 0 ? A / 2 : 1];  // w_A^i
    wp[0] = Fr::one();
#pragma unroll
    for (int i = 1; i < A / 2; i++) wp[i] = mul(wp[i - 1], wA);
#pragma unroll
    for (int st = 0; st < LOGA; st++) {
        const int m = 1 << st;
#pragma unroll
        for (int k = 0; k < A; k += 2 * m)
#pragma unroll
            for (int j = 0; j < m; j++) {
                Fr t = j == 0 ? y[k + j + m] : mul(y[k + j + m], wp[j << (LOGA - 1 - st)]);
                Fr u = y[k + j];
                y[k + j] = add(u, t);
                y[k + j + m] = sub(u, t);
            }
    }
    // inter-level twiddle w_n^(j2 k1), scale folded in
    const Fr w1 = mul(pw_hi[j2 >> lo_bits], pw_lo[j2 & (((size_t)1 << lo_bits) - 1)]);
    Fr w = scale;
#pragma unroll
    for (int k1 = 0; k1 < A; k1++) {
        data[(size_t)k1 * B + j2] = mul(y[k1], w);
        w = mul(w, w1);
    }
}

template <int LOGA>
__global__ __launch_bounds__(256) void k_ntt_outer_transpose(const Fr *in, Fr *out, size_t B) {
    constexpr int A = 1 << LOGA;
    const size_t k2 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k2 >= B) return;
#pragma unroll
    for (int k1 = 0; k1 < A; k1++) out[(size_t)A * k2 + k1] = in[(size_t)k1 * B + k2];
}

static int ntt_run_large(kzg_ctx *ctx, int lane, Fr *d_data, uint32_t log_n, int inverse) {
    const uint32_t la = log_n - 24;
    const size_t B = (size_t)1 << 24, n = (size_t)1 << log_n, A = (size_t)1 << la;
    hipStream_t st = ctx->lanes[lane].stream;
    Fr w = host_omega(log_n);
    if (inverse) w = inv(w);
    const uint32_t lo_bits = 12;
    Fr *scratch = (Fr *)lane_alloc(ctx, lane, n * sizeof(Fr));
    Fr *pw_lo = (Fr *)lane_alloc(ctx, lane, ((size_t)1 << lo_bits) * sizeof(Fr));
    Fr *pw_hi = (Fr *)lane_alloc(ctx, lane, (B >> lo_bits) * sizeof(Fr));
    if (!scratch || !pw_lo || !pw_hi) return fail(ctx, KZG_ERR_ALLOC, "NTT workspace not reserved");
    KZG_TRY(pow_table(ctx, st, w, Fr::one(), (size_t)1 << lo_bits, pw_lo));
    KZG_TRY(pow_table(ctx, st, pow_u64(w, 1ull << lo_bits), Fr::one(), B >> lo_bits, pw_hi));
    const Fr wA = pow_u64(w, (uint64_t)B);
    const Fr scale = inverse ? inv(from_u64<FrParams>((uint64_t)A)) : Fr::one();
    const unsigned grid = (unsigned)(B / 256);
    switch (la) {
        case 1: KZG_LAUNCH(ctx, st, "k_ntt_outer", k_ntt_outer<1>, grid, 256, 0, d_data, B, pw_lo, pw_hi, lo_bits, wA, scale); break;
        case 2: KZG_LAUNCH(ctx, st, "k_ntt_outer", k_ntt_outer<2>, grid, 256, 0, d_data, B, pw_lo, pw_hi, lo_bits, wA, scale); break;
        case 3: KZG_LAUNCH(ctx, st, "k_ntt_outer", k_ntt_outer<3>, grid, 256, 0, d_data, B, pw_lo, pw_hi, lo_bits, wA, scale); break;
        default: KZG_LAUNCH(ctx, st, "k_ntt_outer", k_ntt_outer<4>, grid, 256, 0, d_data, B, pw_lo, pw_hi, lo_bits, wA, scale); break;
    }
    for (size_t k1 = 0; k1 < A; k1++) {
        const size_t mark = ctx->lanes[lane].arena_used;  // the inner transform's scratch is released after each row
        KZG_TRY(ntt_run(ctx, lane, d_data + k1 * B, 24, inverse));
        ctx->lanes[lane].arena_used = mark;
    }
    switch (la) {
        case 1: KZG_LAUNCH(ctx, st, "k_ntt_outer_transpose", k_ntt_outer_transpose<1>, grid, 256, 0, d_data, scratch, B); break;
        case 2: KZG_LAUNCH(ctx, st, "k_ntt_outer_transpose", k_ntt_outer_transpose<2>, grid, 256, 0, d_data, scratch, B); break;
        case 3: KZG_LAUNCH(ctx, st, "k_ntt_outer_transpose", k_ntt_outer_transpose<3>, grid, 256, 0, d_data, scratch, B); break;
        default: KZG_LAUNCH(ctx, st, "k_ntt_outer_transpose", k_ntt_outer_transpose<4>, grid, 256, 0, d_data, scratch, B); break;
    }
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(d_data, scratch, n * sizeof(Fr), hipMemcpyDeviceToDevice, st));
    return KZG_OK;
}

// does ntt_run take the short-input path for a transform of 2^log_n points whose input has nnz leading non-zero elements?  (Then
// d_data[nnz ..) is never read and need not be zero-filled.)
bool ntt_short_input_ok(uint32_t log_n, size_t nnz) { return log_n > 12 && log_n <= 24 && nnz <= ((size_t)1 << (log_n / 2)); }

// nnz: the caller vouches that d_data[nnz ..) is zero (it need not even be written): a zero-padded short polynomial.  With
// nnz <= n2 the column pass is skipped (k_ntt_pass2<true>).
int ntt_run(kzg_ctx *ctx, int lane, Fr *d_data, uint32_t log_n, int inverse, size_t nnz) {
    if (log_n >= FR_TWO_ADICITY) return fail(ctx, KZG_ERR_DEGREE_TOO_LARGE, "polynomial degree too large");
    if (log_n > 28) return fail(ctx, KZG_ERR_SHAPE, "NTT sizes above 2^28 are not supported (2^24-point two-pass transforms under one 16-point outer level)");
    if (log_n > 24) return ntt_run_large(ctx, lane, d_data, log_n, inverse);
    // three passes from 2^ntt_three_from on (default 23: profiles/r06_ab_ntt.txt), unless the input is short enough for the two-pass transform's short-input
    // path (one pass) or the round-5 kernels are selected
    if (ctx->opt_ntt_kernel && ctx->opt_ntt_three_from && log_n >= (uint32_t)ctx->opt_ntt_three_from && log_n >= 20 &&
        !(nnz <= ((size_t)1 << (log_n - (log_n + 1) / 2))))
        return ntt_run3(ctx, lane, d_data, log_n, inverse);
    hipStream_t st = ctx->lanes[lane].stream;
    NttPlan *p = nullptr;
    KZG_TRY(ntt_plan(ctx, st, log_n, inverse, &p));
    if (log_n <= 12) {
        size_t n = (size_t)1 << log_n;
        unsigned threads = n >= 4096 ? 1024 : (n >= 256 ? (unsigned)(n / 4) : 64);
        KZG_LAUNCH(ctx, st, "k_ntt_single", k_ntt_single, 1, threads, n * sizeof(Fr29), d_data, log_n, p->tw1, p->scale, p->scale_p,
                   LDS_SWIZZLE[log_n][0]);
        return KZG_OK;
    }
    size_t n = (size_t)1 << log_n;
    Fr *scratch = (Fr *)lane_alloc(ctx, lane, n * sizeof(Fr));
    if (!scratch) return fail(ctx, KZG_ERR_ALLOC, "NTT workspace not reserved");
    // vec adjacent columns/rows per block: as many as fit the LDS (4096 elements x 36 B = 144 KiB), at most 4 -- but never so many
    // that fewer than 256 blocks are left for the 256 CUs (2^19: 128 blocks of 4096 -> 256 of 2048, pass 1 54.4 -> 38.8 us; 2^18:
    // 35.0 -> 28.1 us; profiles/r05_ntt_tile_probe.txt)
    const uint32_t vmax = (uint32_t)ctx->opt_ntt_vec_log;
    uint32_t vec1 = 12 - p->k1 < vmax ? 12 - p->k1 : vmax;
    while (vec1 && (1u << (p->k2 - vec1)) < 256u) vec1--;
    // rows are contiguous, so narrower pass-2 tiles cost no coalescing and two blocks share a CU: one loads / stores while the
    // other computes (same-box at 2^20: 58.2 -> 54.2 us; 2^22: 2048-element tiles 224.7 -> 205.2 us); pass 1 needs its 128-byte
    // column segments
    const uint32_t vmax2 = vmax < (uint32_t)ctx->opt_ntt_vec2_log ? vmax : (uint32_t)ctx->opt_ntt_vec2_log;
    uint32_t vec2 = 12 - p->k2 < vmax2 ? 12 - p->k2 : vmax2;
    if (ctx->opt_ntt_vec2_log <= 1 && p->k2 <= 11 && p->k2 + vec2 > 11) vec2 = 11 - p->k2;
    while (vec2 && (1u << (p->k1 - vec2)) < 256u) vec2--;
    size_t lds1 = ((size_t)1 << (p->k1 + vec1)) * sizeof(Fr29), lds2 = ((size_t)1 << (p->k2 + vec2)) * sizeof(Fr29);
    unsigned g1 = 1u << (p->k2 - vec1), g2 = 1u << (p->k1 - vec2);
    // one radix-4 butterfly per thread and stage; smaller tiles leave room for a second block per CU
    unsigned th1 = std::min(1024u, 1u << (p->k1 + vec1 - 2)), th2 = std::min(1024u, 1u << (p->k2 + vec2 - 2));
    // XCD-aware tile order (xcd_tile): pass 1 always; pass 2 only at 2^24, the one size where it measured a gain (option ntt_xcd: 0 off,
    // 1 this rule, 2 pass 2 only, 3 both; profiles/r05_ntt_xcd_probe.txt)
    const int xcd1 = ctx->opt_ntt_xcd & 1, xcd2 = ctx->opt_ntt_xcd == 1 ? (log_n >= 24) : (ctx->opt_ntt_xcd >> 1 & 1);
    if (ctx->opt_ntt_kernel) {  // round 6: load / store fused into the first / last stage pair (k_ntt_tile)
        NttTile a1, a2;
        a1.t = p->k1; a1.kother = p->k2; a1.vec_log = vec1; a1.lo_bits = p->lo_bits; a1.nnz = 0; a1.xcd = xcd1; a1.scale_folded = 0;
        a1.sw = LDS_SWIZZLE[p->k1][vec1]; a1.tw = p->tw1; a1.tw_lo = p->tw_lo; a1.tw_hi = p->tw_hi; a1.tw_full = p->tw_full;
        a1.scale = p->scale; a1.scale_p = p->scale_p;
        a2 = a1;
        a2.t = p->k2; a2.kother = p->k1; a2.vec_log = vec2; a2.xcd = xcd2; a2.scale_folded = p->tw_full ? 1 : 0;
        a2.sw = LDS_SWIZZLE[p->k2][vec2]; a2.tw = p->tw2;
        const unsigned nb1 = 1u << (p->k1 + vec1 - 2), nb2 = 1u << (p->k2 + vec2 - 2);  // radix-4 butterflies per tile
        const bool two1 = ctx->opt_ntt_kernel == 2 && nb1 >= 512, two2 = ctx->opt_ntt_kernel == 2 && nb2 >= 512;
        const bool is_short = nnz <= ((size_t)1 << p->k2);
        if (is_short) {
            if (nnz) KZG_HIP_CHECK(ctx, hipMemcpyAsync(scratch, d_data, nnz * sizeof(Fr), hipMemcpyDeviceToDevice, st));  // pass 2 writes d_data
            a2.nnz = (uint32_t)nnz;
            if (two2) KZG_LAUNCH(ctx, st, "k_ntt_pass2_short", (k_ntt_tile<2, 2, true>), g2, nb2 / 2, lds2, scratch, d_data, a2);
            else KZG_LAUNCH(ctx, st, "k_ntt_pass2_short", (k_ntt_tile<2, 1, true>), g2, nb2, lds2, scratch, d_data, a2);
            return KZG_OK;
        }
        a2.tw_full = nullptr;
        if (two1) KZG_LAUNCH(ctx, st, "k_ntt_pass1", (k_ntt_tile<1, 2, false>), g1, nb1 / 2, lds1, d_data, scratch, a1);
        else KZG_LAUNCH(ctx, st, "k_ntt_pass1", (k_ntt_tile<1, 1, false>), g1, nb1, lds1, d_data, scratch, a1);
        if (two2) KZG_LAUNCH(ctx, st, "k_ntt_pass2", (k_ntt_tile<2, 2, false>), g2, nb2 / 2, lds2, scratch, d_data, a2);
        else KZG_LAUNCH(ctx, st, "k_ntt_pass2", (k_ntt_tile<2, 1, false>), g2, nb2, lds2, scratch, d_data, a2);
        return KZG_OK;
    }
    if (nnz <= ((size_t)1 << p->k2)) {
        if (nnz) KZG_HIP_CHECK(ctx, hipMemcpyAsync(scratch, d_data, nnz * sizeof(Fr), hipMemcpyDeviceToDevice, st));  // pass 2 writes d_data
        KZG_LAUNCH(ctx, st, "k_ntt_pass2_short", k_ntt_pass2<true>, g2, th2, lds2, scratch, d_data, p->k1, p->k2, vec2, p->tw2, p->scale,
                   p->scale_p, p->tw_full ? 1 : 0, LDS_SWIZZLE[p->k2][vec2], (uint32_t)nnz, p->tw_full, p->tw_lo, p->tw_hi, p->lo_bits, xcd2);
        return KZG_OK;
    }
    KZG_LAUNCH(ctx, st, "k_ntt_pass1", k_ntt_pass1, g1, th1, lds1, d_data, scratch, p->k1, p->k2, vec1, p->tw1, p->tw_lo,
               p->tw_hi, p->lo_bits, p->tw_full, LDS_SWIZZLE[p->k1][vec1], xcd1);
    KZG_LAUNCH(ctx, st, "k_ntt_pass2", k_ntt_pass2<false>, g2, th2, lds2, scratch, d_data, p->k1, p->k2, vec2, p->tw2, p->scale, p->scale_p,
               p->tw_full ? 1 : 0, LDS_SWIZZLE[p->k2][vec2], 0u, (const Fr29 *)nullptr, Tw29(), Tw29(), 0u, xcd2);
    return KZG_OK;
}

}  // namespace kzg
