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
constexpr int SUM_L = 4;  // fan-in of the plain tree sum
// wide windows (16 < c <= 20): bucket id = hi (c - 16 bits) : lo (15 bits).  Pass 1 sorts by lo with the LDS counting sort,
// pass 2 is a stable partition by hi; between the passes hi travels in bits 27..30 of the entry word, which limits the
// wide mode to W * npad < 2^27 table rows (n <= 2^22 at W = 13).
constexpr int WIDE_LO_BITS = 15;
constexpr int WIDE_HI_SHIFT = 27;
constexpr uint32_t WIDE_HI_MASK = 0xfu << WIDE_HI_SHIFT;

struct MsmState {
    uint32_t M;            // sorted entries
    uint32_t ntasks;       // threads of round 1 that get work: ceil(M / E)
    uint32_t reserved[4];
    uint32_t E;            // sorted entries per round-1 thread (equal split)
    uint32_t ovf_tasks;    // slices of buckets with too many partials for k_fold_dense (msm_tail.hip); zeroed by k_scan_buckets
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
// msm_tail.hip: everything after the bucket accumulation (fold to one point per bucket, sum (b+1) B_b)
struct TailLayout {
    size_t off_dense, off_rows, off_cols, off_Q, off_tasks, off_arrive, off_result, bytes;
};
TailLayout tail_layout(int B, size_t T1_max);
int msm_tail_run(kzg_ctx *ctx, hipStream_t st, const MsmPoint *part, MsmPoint *scratch, const uint32_t *s1, int B,
                 size_t expected_partials, MsmState *state, char *tail_base, const TailLayout &L, MsmPoint **d_result);

}  // namespace kzg
