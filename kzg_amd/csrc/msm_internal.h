// msm_internal.h -- declarations shared by msm.hip (the c <= 16 pipeline and all orchestration) and msm_wide.hip (the
// kernels only the wide-window path uses; kept in their own translation unit so that they do not perturb the code
// generation of the hot kernels: a same-box A/B showed k_accum_affine 2.5 % and k_bucket_reduce 13 % slower with
// everything in one TU).
#pragma once
#include "common.h"

namespace kzg {

constexpr uint32_t ACC_SLOTS = 256 * 4 * KZG_ACCUM_WAVES * 64;  // resident threads of k_accum_affine: 256 CUs x 4 SIMDs x waves/SIMD x 64
// Latency-bound tail kernels run next to other MSMs' accumulation kernels (batched mode).  A block of k_accum_affine is 4
// waves, one per SIMD of a CU, so a lone 64-thread tail block on one SIMD strands the other three SIMDs of that
// half-CU for as long as it lives; 256-thread tail blocks take exactly one accumulation-block slot instead.
constexpr int TAIL_THREADS = 256;
#ifndef KZG_FOLD_FANIN
#define KZG_FOLD_FANIN 8
#endif
#ifndef KZG_FAST_LEVELS
#define KZG_FAST_LEVELS 2
#endif
// Two grid-wide fold rounds of fan-in 8 settle every bucket that was split over <= 64 threads: one round for uniform scalars at
// 2^20 (4-6 partials per bucket; the second launch pair then returns at once), two at the small sizes where a bucket is many
// equal-split chunks long (2^16: c = 12, ~64 partials per bucket).  The single-block k_fold_rest finishes the others.
// Same-box A/B against two rounds of fan-in 4: 2^20 equal, u64-valued +4 %, 2^16 +19 % (single commit 2.1 -> 1.4 ms).
constexpr int LK = KZG_FOLD_FANIN;   // fan-in of the fold rounds
constexpr int SUM_L = 4;  // fan-in of the plain tree sum
#ifndef KZG_REDUCE_CH
#define KZG_REDUCE_CH 8
#endif
constexpr int REDUCE_CH = KZG_REDUCE_CH;  // buckets per k_bucket_reduce thread
constexpr int MAX_LEVELS = 24;
// wide windows (16 < c <= 20): bucket id = hi (c - 16 bits) : lo (15 bits).  Pass 1 sorts by lo with the LDS counting sort,
// pass 2 is a stable partition by hi; between the passes hi travels in bits 27..30 of the entry word, which limits the
// wide mode to W * npad < 2^27 table rows (n <= 2^22 at W = 13).
constexpr int WIDE_LO_BITS = 15;
constexpr int WIDE_HI_SHIFT = 27;
constexpr uint32_t WIDE_HI_MASK = 0xfu << WIDE_HI_SHIFT;

struct MsmState {
    uint32_t M;            // sorted entries
    uint32_t ntasks;       // tasks of the level being run
    uint32_t done;         // every bucket holds <= 1 partial
    uint32_t final_level;  // index of the start[] array describing the final partial list
    uint32_t final_buf;    // which ping-pong buffer holds it
    uint32_t max_cnt;
    uint32_t E;            // sorted entries per round-1 thread (equal split)
    uint32_t pad[1];
};


#if defined(__HIPCC__)
// Single-block exclusive scan over `B` per-bucket values produced by f(b); writes out[0..B] and returns the
// total.  Each wave owns a contiguous segment and walks it in rounds of 64 consecutive buckets, so every
// global access is a coalesced 256-B row.  Two code paths:
//   * the segment fits SCAN_MAX_ROUNDS rounds (1024 threads at B = 2^15): all rounds are loaded into registers first
//     (their latencies overlap), scanned with wave shuffles and a running carry, one LDS pass combines the waves;
//   * longer segments (the 256-thread launches of the batched mode, where a block must fit beside resident accumulation
//     blocks instead of waiting for a whole free CU): two passes over the segment in batches of SCAN_BATCH rounds --
//     wave totals first, then the scan proper with the wave offset known; f is evaluated twice.
constexpr int SCAN_MAX_ROUNDS = 32;  // 1024 threads x 32 rounds = 2^15 buckets (c <= 16)
constexpr int SCAN_BATCH = 8;
template <class F>
__device__ __forceinline__ uint32_t block_exclusive_scan(int B, F f, uint32_t *out, uint32_t *lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int seg = (B + nwaves - 1) / nwaves;          // buckets per wave
    const int rounds = (seg + 63) >> 6;
    const int base = wave * seg;
    const int end = base + seg < B ? base + seg : B;
    if (rounds <= SCAN_MAX_ROUNDS) {
        uint32_t v[SCAN_MAX_ROUNDS];
#pragma unroll
        for (int r = 0; r < SCAN_MAX_ROUNDS; r++) {
            int b = base + r * 64 + lane;
            v[r] = (r < rounds && b < end) ? f(b) : 0u;
        }
        uint32_t carry = 0;
#pragma unroll
        for (int r = 0; r < SCAN_MAX_ROUNDS; r++) {
            uint32_t x = v[r], incl = x;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                uint32_t t = __shfl_up(incl, off, 64);
                if (lane >= off) incl += t;
            }
            v[r] = incl - x + carry;
            carry += __shfl(incl, 63, 64);
        }
        if (lane == 0) lds[wave] = carry;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t run = 0;
            for (int w = 0; w < nwaves; w++) {
                uint32_t t = lds[w];
                lds[w] = run;
                run += t;
            }
            lds[nwaves] = run;
        }
        __syncthreads();
        const uint32_t woff = lds[wave], total = lds[nwaves];
#pragma unroll
        for (int r = 0; r < SCAN_MAX_ROUNDS; r++) {
            int b = base + r * 64 + lane;
            if (r < rounds && b < end) out[b] = v[r] + woff;
        }
        if (threadIdx.x == 0) out[B] = total;
        __syncthreads();
        return total;
    }
    // long segments: pass 1, wave totals
    uint32_t mine = 0;
    for (int r0 = 0; r0 < rounds; r0 += SCAN_BATCH) {
        uint32_t x[SCAN_BATCH];
#pragma unroll
        for (int j = 0; j < SCAN_BATCH; j++) {
            int b = base + (r0 + j) * 64 + lane;
            x[j] = (r0 + j < rounds && b < end) ? f(b) : 0u;
        }
#pragma unroll
        for (int j = 0; j < SCAN_BATCH; j++) mine += x[j];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off, 64);
    if (lane == 0) lds[wave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int w = 0; w < nwaves; w++) {
            uint32_t t = lds[w];
            lds[w] = run;
            run += t;
        }
        lds[nwaves] = run;
    }
    __syncthreads();
    uint32_t carry = lds[wave];
    const uint32_t total = lds[nwaves];
    // pass 2: the scan proper
    for (int r0 = 0; r0 < rounds; r0 += SCAN_BATCH) {
        uint32_t x[SCAN_BATCH];
#pragma unroll
        for (int j = 0; j < SCAN_BATCH; j++) {
            int b = base + (r0 + j) * 64 + lane;
            x[j] = (r0 + j < rounds && b < end) ? f(b) : 0u;
        }
#pragma unroll
        for (int j = 0; j < SCAN_BATCH; j++) {
            uint32_t incl = x[j];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                uint32_t t = __shfl_up(incl, off, 64);
                if (lane >= off) incl += t;
            }
            int b = base + (r0 + j) * 64 + lane;
            if (r0 + j < rounds && b < end) out[b] = incl - x[j] + carry;
            carry += __shfl(incl, 63, 64);
        }
    }
    if (threadIdx.x == 0) out[B] = total;
    __syncthreads();
    return total;
}

__device__ __forceinline__ void find_task(const uint32_t *task_start, int B, uint32_t t, uint32_t &b, uint32_t &j) {
    uint32_t lo = 0, hi = (uint32_t)B;  // task_start[lo] <= t < task_start[hi]
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (task_start[mid] <= t) lo = mid; else hi = mid;
    }
    b = lo;
    j = t - task_start[lo];
}

#endif

// msm.hip
int sum_level_run(kzg_ctx *ctx, hipStream_t st, const MsmPoint *in, uint32_t count, int L, MsmPoint *out);

// msm_wide.hip
constexpr int HI_BLOCKS = 256;   // blocks of the stable partition
constexpr int HI_THREADS = 1024;
constexpr int SEG = 2048;        // buckets per scan block (256 threads x 8)
int wide_sort_pass2(kzg_ctx *ctx, hipStream_t st, const uint32_t *entries1, const MsmState *state, int nhi, uint32_t *blockcnt,
                    uint32_t *binbase, const uint32_t *lo_start, int B_lo, uint32_t *entries2, uint32_t *bucket_start);
int wide_s1_layout(kzg_ctx *ctx, hipStream_t st, const uint32_t *bucket_start, int Btot, MsmState *state, uint32_t *segsums,
                   uint32_t *segmaxs, uint32_t *segtotal, uint32_t *s1_out);
// one fold level: task counts ceil(partials / L) per bucket, `done` detection; apply = also write out_start
int wide_level_scan(kzg_ctx *ctx, hipStream_t st, const uint32_t *in_start, uint32_t *out_start, int Btot, int L, MsmState *state,
                    uint32_t level, uint32_t in_buf, uint32_t *segsums, uint32_t *segmaxs, uint32_t *segtotal, bool apply);
// sum (b+1) X_b over Btot = R x C buckets -> *result
// the fold levels beyond the first FAST_LEVELS, in ONE single-block kernel that returns at once when every bucket already
// holds one partial (the normal case): replaces ~14 no-op launches per MSM, whose queueing delays under a full GPU cost
// 3-4 % of the batched throughput (measured).  Only adversarial inputs (few distinct digits) ever do work here.
constexpr int FAST_LEVELS = KZG_FAST_LEVELS;
int fold_rest_run(kzg_ctx *ctx, hipStream_t st, MsmPoint *buf0, MsmPoint *buf1, uint32_t *starts, int B, int L, int level0,
                  int max_level, MsmState *state);
int wide_bucket_reduce(kzg_ctx *ctx, int lane, const MsmPoint *buf0, const MsmPoint *buf1, const uint32_t *starts, int Btot, int C,
                       const MsmState *state, MsmPoint *rows, MsmPoint *cols, MsmPoint *red0, MsmPoint *red1, MsmPoint *chunks,
                       MsmPoint *sum_scratch, MsmPoint *scratch3, MsmPoint *result);

}  // namespace kzg
