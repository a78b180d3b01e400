// msm.hip -- Pippenger multi-scalar multiplication over BLS12-381 G1 for gfx950.
//
// Replaces G1Projective::multi_exp at the reference's call sites (src/coeff_form.rs:61,78,102;
// src/eval_form.rs:118,136).  The result is the group element sum_i s_i * P_i; how it is computed
// is free, so the structure below is chosen for the MI355X, not for the CPU the reference runs on:
//
//   * The SRS is resident in HBM as W rows (W = ceil(256/c); 15 at c = 17, the width used from 2^17 points on), row w holding
//     the affine points 2^(c*w) * P_i in the signed 30-bit representation the inner loop computes in (112 B per point,
//     1.64 GiB for 2^20 points -- cheap in 288 GB); row 0 also in the canonical 96-byte form.
//     Every signed c-bit digit of every scalar therefore lands in ONE shared set of 2^(c-1)
//     buckets: there is no per-window bucket reduction and no window-combine doubling chain.
//   * Digits -> buckets by a one-pass counting sort whose histogram / cursor array lives in LDS (2^15 u32 counters = 128 KiB
//     of the CU's 160 KiB; at c = 17 the blocks walk their scalars twice, half the 2^16 buckets per walk): k_hist,
//     k_scan_*, k_scatter.
//   * Bucket accumulation (k_accum_affine) is an equal split of the sorted entry list over exactly
//     the resident thread slots (XYZZ mixed adds, points gathered from the resident table, one partial
//     per bucket a thread touches).  No atomics, no unbounded per-thread chain, so adversarial inputs (all-equal
//     scalars) stay bounded.
//   * msm_tail.hip: the partial sums are folded into one point per bucket by lane groups, sum_b (b+1) * B_b by row /
//     column sums + bit-sliced weighted sums (depth ~35 additions), and one Fq inversion for the affine result
//     (k_emit_points).
//
// Everything is enqueued on one stream with device-side counts; the host never syncs inside an MSM.
#include "msm_internal.h"
#include "msm_sort.h"
#include "emit.h"
#include "naf.h"

namespace kzg {

// dispatch: the fixed-width version for c = 17 (W = 15, balanced), the generic one otherwise
template <class F>
__device__ __forceinline__ void digits_of(const uint32_t s[8], int c, int W, bool balanced, F f) {
    if (c == 17 && W == 15) for_each_digit_fixed<17, 15>(s, balanced, f);
    else for_each_digit(s, c, W, balanced, f);
}

// ---------------------------------------------------------------------------------------------
// counting sort by bucket, LDS histogram / cursors
// ---------------------------------------------------------------------------------------------
extern __shared__ __attribute__((aligned(16))) uint32_t lds_u32[];

// mode 0: B u32 counters (c <= 16, or the low 15 bucket bits of the wide path); mode 2 (c = 17 single pass): B = 2^16 buckets, the
// u32 counters of half of them fit the LDS, so the block walks its scalars twice (as k_scatter does); balanced scalars
// [w_lo, w_hi): the windows of this pass (all of them unless the SRS keeps fewer table rows than windows, option window_rows)
__global__ __launch_bounds__(1024) void k_hist(const Fr *scalars, size_t n, int sfmt, int c, int W, int B,
                                               size_t per_block, uint32_t *blk_hist, int mode, int w_lo, int w_hi) {
    const bool pk = mode == 2;
    const int BH = pk ? B / 2 : B;
    size_t i0 = (size_t)blockIdx.x * per_block;
    size_t i1 = i0 + per_block < n ? i0 + per_block : n;
    uint32_t *dst = blk_hist + (size_t)blockIdx.x * B;
    for (int half = 0; half < (pk ? 2 : 1); half++) {
        const uint32_t base = (uint32_t)half * (uint32_t)BH;
        if (half) __syncthreads();
        for (int b = threadIdx.x; b < BH; b += blockDim.x) lds_u32[b] = 0;
        __syncthreads();
        for (size_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
            uint32_t s[8];
            load_scalar(scalars, i, sfmt, s);
            digits_of(s, c, W, pk, [&](int w, uint32_t mag, uint32_t) {
                if (w < w_lo || w >= w_hi) return;
                const uint32_t idx = (mag - 1) & (uint32_t)(B - 1);
                if (pk && (idx >> 15) != (uint32_t)half) return;
                atomicAdd(&lds_u32[idx - base], 1u);
            });
        }
        __syncthreads();
        for (int b = threadIdx.x; b < BH; b += blockDim.x) dst[base + b] = lds_u32[b];
    }
}

// ---- bucket starts and the round-1 layout, in two multi-block kernels (no single-block pass over all buckets) ----
// k_scan_a: block j owns the 256 buckets [256 j, 256 j + 256).  Per bucket: exclusive scan over the sort blocks' counts (in
// place), total[b] = bucket size; then the block-local exclusive scan local[b] of the totals and the block aggregate agg[j].
constexpr int SCAN_SEG = 256;
__global__ __launch_bounds__(SCAN_SEG) void k_scan_a(uint32_t *blk_hist, int G, int B, uint32_t *total, uint32_t *local, uint32_t *agg,
                                                     uint32_t *ready) {
    __shared__ uint32_t lds[4];
    if (threadIdx.x == 0) ready[blockIdx.x] = 0;  // k_scan_b's chained flag counts
    const int b = blockIdx.x * SCAN_SEG + threadIdx.x;
    uint32_t run = 0;
    if (b < B) {
        for (int g = 0; g < G; g++) {
            uint32_t t = blk_hist[(size_t)g * B + b];
            blk_hist[(size_t)g * B + b] = run;
            run += t;
        }
        total[b] = run;
    }
    uint32_t tot;
    uint32_t ex = block_scan_256(run, lds, &tot);
    if (b < B) local[b] = ex;
    if (threadIdx.x == 0) agg[blockIdx.x] = tot;
}

// k_scan_b: every block scans the (<= 256) block aggregates itself -> M, the equal-split chunk E, its own bucket starts.
// Round 1 is an EQUAL SPLIT: thread s folds the sorted entries [s*E, (s+1)*E), E = ceil(M / slots), and emits one partial per
// (thread, bucket) run.  Run starts are the multiples of E and the non-empty bucket starts, so the partial list is ordered by
// bucket and bucket b's partials are [S1[b], S1[b+1]) with
//     S1[b] = ceil(start[b] / E) + #{non-empty b' < b : start[b'] mod E != 0}.
// The count over the preceding blocks is a chained scan: block j publishes its own count in ready[j] (count << 1 | 1, one
// relaxed atomic word) and thread t < j of block j waits for ready[t].  Blocks are dispatched in order, so a block only ever
// waits for blocks that are already running or done (the forward-progress assumption of every decoupled look-back scan).
__device__ __forceinline__ uint32_t mod_u32(uint32_t x, uint32_t e, double rcp_e) {
    uint32_t q = (uint32_t)((double)x * rcp_e);
    uint32_t r = x - q * e;          // q is floor(x / e) or one off in either direction
    if ((int32_t)r < 0) r += e;
    if (r >= e) r -= e;
    return r;
}

__global__ __launch_bounds__(SCAN_SEG) void k_scan_b(const uint32_t *total, const uint32_t *local, const uint32_t *agg, int B, int NB,
                                                     uint32_t *bucket_start, uint32_t *s1, MsmState *st, uint32_t slots, uint32_t *ready) {
    __shared__ uint32_t lds[4];
    __shared__ uint32_t my_prefix;
    uint32_t M;
    {
        uint32_t v = (int)threadIdx.x < NB ? agg[threadIdx.x] : 0u;
        uint32_t ex = block_scan_256(v, lds, &M);
        if (threadIdx.x == blockIdx.x) my_prefix = ex;
    }
    __syncthreads();
    uint32_t E = (M + slots - 1) / slots;
    if (E < KZG_ACCUM_MIN_CHUNK) E = KZG_ACCUM_MIN_CHUNK;
    const double rcp_e = 1.0 / (double)E;
    const int b = blockIdx.x * SCAN_SEG + threadIdx.x;
    uint32_t start = 0, f = 0;
    if (b < B) {
        start = my_prefix + local[b];
        f = (total[b] != 0 && mod_u32(start, E, rcp_e) != 0) ? 1u : 0u;
    }
    uint32_t ftot;
    const uint32_t fex = block_scan_256(f, lds, &ftot);
    if (threadIdx.x == 0) __hip_atomic_store(&ready[blockIdx.x], (ftot << 1) | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t prev = 0;
    if (threadIdx.x < blockIdx.x) {
        uint32_t w;
        while (((w = __hip_atomic_load(&ready[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 1u) == 0) __builtin_amdgcn_s_sleep(1);
        prev = w >> 1;
    }
    uint32_t before;
    block_scan_256(prev, lds, &before);
    if (b < B) {
        bucket_start[b] = start;
        s1[b] = before + fex + (start + E - 1) / E;
    }
    if (blockIdx.x == (unsigned)NB - 1 && threadIdx.x == 0) {
        bucket_start[B] = M;
        s1[B] = before + ftot + (M + E - 1) / E;
        st->M = M;
        st->E = E;
        st->ntasks = (M + E - 1) / E;
        st->ovf_tasks = 0;
    }
}

static int scan_run(kzg_ctx *ctx, hipStream_t st, uint32_t *blk_hist, int G, int B, uint32_t *total, uint32_t *local, uint32_t *agg,
                    uint32_t *bucket_start, uint32_t *s1, MsmState *state, uint32_t slots) {
    const int NB = (B + SCAN_SEG - 1) / SCAN_SEG;  // <= 256 (B <= 2^16)
    uint32_t *ready = agg + SCAN_SEG;
    KZG_LAUNCH(ctx, st, "k_scan_a", k_scan_a, NB, SCAN_SEG, 0, blk_hist, G, B, total, local, agg, ready);
    KZG_LAUNCH(ctx, st, "k_scan_b", k_scan_b, NB, SCAN_SEG, 0, total, local, agg, B, NB, bucket_start, s1, state, slots, ready);
    return KZG_OK;
}

// mode 0: c <= 16 (u32 cursors in LDS); mode 1: wide path (sorts by the low 15 bucket bits, the high bits ride in bits 27..30 of
// the entry word until the second pass strips them); mode 2: c = 17 single pass, 2^16 buckets: the 32-bit cursors of only half the
// buckets fit the LDS, so the block walks its scalars twice, scattering the digits of one bucket half per walk (the digit
// extraction is cheap next to the scattered stores; cursors kept in L2 instead cost +83 % on this kernel)
__global__ __launch_bounds__(1024) void k_scatter(const Fr *scalars, size_t n, int sfmt, int c, int W, int B,
                                                  size_t per_block, const uint32_t *blk_off,
                                                  const uint32_t *bucket_start, uint32_t row_stride,
                                                  uint32_t idx_base, uint32_t *entries, int mode, int w_lo, int w_hi) {
    const uint32_t *off = blk_off + (size_t)blockIdx.x * B;
    const bool pk = mode == 2;
    const int BH = pk ? B / 2 : B;  // buckets whose cursors are resident per walk
    size_t i0 = (size_t)blockIdx.x * per_block;
    size_t i1 = i0 + per_block < n ? i0 + per_block : n;
    for (int half = 0; half < (pk ? 2 : 1); half++) {
        const uint32_t base = (uint32_t)half * (uint32_t)BH;
        if (half) __syncthreads();
        for (int b = threadIdx.x; b < BH; b += blockDim.x) lds_u32[b] = bucket_start[base + b] + off[base + b];
        __syncthreads();
        for (size_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
            uint32_t s[8];
            load_scalar(scalars, i, sfmt, s);
            digits_of(s, c, W, pk, [&](int w, uint32_t mag, uint32_t neg) {
                if (w < w_lo || w >= w_hi) return;
                const uint32_t idx = (mag - 1) & (uint32_t)(B - 1);
                if (pk && (idx >> 15) != (uint32_t)half) return;
                const uint32_t pos = atomicAdd(&lds_u32[idx - base], 1u);
                const uint32_t hi = mode == 1 ? (((mag - 1) >> WIDE_LO_BITS) << WIDE_HI_SHIFT) : 0u;
                entries[pos] = ((uint32_t)(w - w_lo) * row_stride + idx_base + (uint32_t)i) | hi | (neg << 31);
            });
        }
    }
}

// ---------------------------------------------------------------------------------------------
// two-level counting sort for the production width (c = 17, 2^16 buckets = 1024 bins of 64)
// ---------------------------------------------------------------------------------------------
// The single-pass scatter above writes every 4-byte entry to its own 32-byte sector: 256 producer blocks x 2^16 buckets leave
// about one entry per (block, bucket), so the stores are 8x write-amplified (PMC: 504 MB written for 63 MB of entries) and the
// per-(block, bucket) counters are another 2 x 64 MB.  Two levels keep every store stream sequential:
//   level 1  k_bin_hist / k_bin_scan / k_bin_scatter: by the high 10 bucket bits.  Each sort block appends (entry, low 6 bits)
//            records to 1024 runs of ~30-60 records; a run's stores hit consecutive addresses and merge in the L2.
//   level 2  k_bin_sort: one block per bin counts its 64 buckets in LDS, then places the bin's records: all stores of a block
//            land in the bin's own <= few-hundred-KB range.
// The bucket sizes come out of level 2 (total[]), bin_base[] is the exclusive scan of the bin sizes = the bucket starts of every
// 64th bucket, so k_scan_b needs no cross-block pass for the starts.
constexpr int BIN_SHIFT = 6, BIN_BUCKETS = 1 << BIN_SHIFT;  // NBINS = 1024 bins (msm_internal.h)

__global__ __launch_bounds__(1024) void k_bin_hist(const Fr *scalars, size_t n, int sfmt, size_t per_block, uint32_t *blk_bins,
                                                   int w_lo, int w_hi) {
    KZG_SIDE_PRIO_STMT;
    __shared__ uint32_t h[NBINS];
    for (int b = threadIdx.x; b < NBINS; b += blockDim.x) h[b] = 0;
    __syncthreads();
    size_t i0 = (size_t)blockIdx.x * per_block;
    size_t i1 = i0 + per_block < n ? i0 + per_block : n;
    for (size_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        uint32_t s[8];
        load_scalar(scalars, i, sfmt, s);
        for_each_digit_fixed<17, 15>(s, true, [&](int w, uint32_t mag, uint32_t) {
            if (w < w_lo || w >= w_hi) return;
            atomicAdd(&h[(mag - 1) >> BIN_SHIFT], 1u);
        });
    }
    __syncthreads();
    for (int b = threadIdx.x; b < NBINS; b += blockDim.x) blk_bins[(size_t)blockIdx.x * NBINS + b] = h[b];
}

// per bin: exclusive scan over the sort blocks' counts (in place) and the bin size.  Block q owns bins [64 q, 64 q + 64); its 16
// waves each take a slice of the sort blocks (<= 32 counters per thread, loaded in one batch), the slices are chained through LDS.
constexpr int BIN_SCAN_SLICES = 16, BIN_SCAN_MAXG = 32;
constexpr int SORT2_MAX_BLOCKS = BIN_SCAN_SLICES * BIN_SCAN_MAXG;  // level-1 sort blocks k_bin_scan can chain (512)
__global__ __launch_bounds__(1024) void k_bin_scan(uint32_t *blk_bins, int G, uint32_t *bin_total, uint32_t *ready) {
    KZG_SIDE_PRIO_STMT;
    __shared__ uint32_t part[BIN_SCAN_SLICES][64];
    if (blockIdx.x == 0 && threadIdx.x < SCAN_SEG) ready[threadIdx.x] = 0;  // k_scan_b's chained flag counts
    const int binl = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int bin = blockIdx.x * 64 + binl;
    const int gs = (G + BIN_SCAN_SLICES - 1) / BIN_SCAN_SLICES;  // <= BIN_SCAN_MAXG (G <= 512)
    const int g0 = slice * gs;
    uint32_t v[BIN_SCAN_MAXG];
#pragma unroll
    for (int k = 0; k < BIN_SCAN_MAXG; k++) v[k] = (k < gs && g0 + k < G) ? blk_bins[(size_t)(g0 + k) * NBINS + bin] : 0u;
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < BIN_SCAN_MAXG; k++) {
        const uint32_t t = v[k];
        v[k] = sum;
        sum += t;
    }
    part[slice][binl] = sum;
    __syncthreads();
    uint32_t pre = 0, tot = 0;
#pragma unroll
    for (int s2 = 0; s2 < BIN_SCAN_SLICES; s2++) {
        const uint32_t t = part[s2][binl];
        if (s2 < slice) pre += t;
        tot += t;
    }
#pragma unroll
    for (int k = 0; k < BIN_SCAN_MAXG; k++)
        if (k < gs && g0 + k < G) blk_bins[(size_t)(g0 + k) * NBINS + bin] = pre + v[k];
    if (slice == 0) bin_total[bin] = tot;
}


// Level 1.  A block takes its scalars in chunks of 1024 (one per thread).  Per chunk the <= 15 K records are first sorted by bin
// inside the LDS (count -> scan -> place, one packed word per record), then written out in that order: a wave's 64 stores
// cover ~4 runs of ~15 consecutive records instead of 64 unrelated addresses, so the L2 sees ~7x fewer write requests.
// packed word: bin (10) | low 6 bucket bits (6) | sign (1) | window - w_lo (4) | thread = scalar index in the chunk (10)
constexpr int BIN_SCATTER_LDS = (3 * NBINS + 16 + 15 * 1024) * 4;
template <class REC>
__global__ __launch_bounds__(1024) void k_bin_scatter(const Fr *scalars, size_t n, int sfmt, size_t per_block, const uint32_t *blk_off,
                                                      const uint32_t *bin_total, uint32_t *bin_base, uint32_t row_stride,
                                                      uint32_t idx_base, typename REC::T *rec, int w_lo, int w_hi) {
    KZG_SIDE_PRIO_STMT;
    uint32_t *cur = lds_u32, *cnt = cur + NBINS, *off = cnt + NBINS, *wsum = off + NBINS, *stage = wsum + 16;
    const uint32_t tid = threadIdx.x;
    {   // bin_base = exclusive scan of the bin sizes (every block computes it; block 0 publishes it for the later kernels)
        uint32_t tot;
        const uint32_t ex = block_scan_1024(bin_total[tid], wsum, &tot);
        cur[tid] = ex + blk_off[(size_t)blockIdx.x * NBINS + tid];
        if (blockIdx.x == 0) {
            bin_base[tid] = ex;
            if (tid == 0) bin_base[NBINS] = tot;
        }
    }
    const size_t i0 = (size_t)blockIdx.x * per_block;
    const size_t i1 = i0 + per_block < n ? i0 + per_block : n;
    for (size_t c0 = i0; c0 < i1; c0 += 1024) {
        cnt[tid] = 0;
        __syncthreads();
        uint32_t pk[15], rk[15];
#pragma unroll
        for (int w = 0; w < 15; w++) pk[w] = 0xffffffffu;
        if (c0 + tid < i1) {
            uint32_t s[8];
            load_scalar(scalars, c0 + tid, sfmt, s);
            for_each_digit_fixed<17, 15>(s, true, [&](int w, uint32_t mag, uint32_t neg) {
                if (w < w_lo || w >= w_hi) return;
                const uint32_t idx = mag - 1, bin = idx >> BIN_SHIFT;
                rk[w] = atomicAdd(&cnt[bin], 1u);
                pk[w] = bin | ((idx & (BIN_BUCKETS - 1)) << 10) | (neg << 16) | ((uint32_t)(w - w_lo) << 17) | (tid << 21);
            });
        }
        __syncthreads();
        uint32_t tot;
        off[tid] = block_scan_1024(cnt[tid], wsum, &tot);
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 15; w++)
            if (pk[w] != 0xffffffffu) stage[off[pk[w] & (NBINS - 1)] + rk[w]] = pk[w];
        __syncthreads();
        const uint32_t ebase = idx_base + (uint32_t)c0;
        for (uint32_t p = tid; p < tot; p += 1024) {
            const uint32_t v = stage[p], bin = v & (NBINS - 1);
            const uint32_t index = ((v >> 17) & 15u) * row_stride + ebase + (v >> 21);
            rec[cur[bin] + p - off[bin]] = REC::pack(index, (v >> 16) & 1u, (v >> 10) & (BIN_BUCKETS - 1));
        }
        __syncthreads();
        cur[tid] += cnt[tid];
    }
}

// ---- positional tables: width-18 NAF digits (kzg_srs::naf) ----------------------------------------------------------------
// A digit record: bits 0..16 bucket index (|d| - 1) / 2, bit 17 sign, bits 18..25 bit position (= table row), bit 31 valid.
// k_naf_recode turns every scalar into <= 15 records (stored digit-ordinal-major, recs[k * n + i]: coalesced) and counts the
// level-1 bins of its sort block on the way (the k_bin_hist of this path); k_bin_scatter_naf is k_bin_scatter reading records.
// Recoding walks the scalar from bit 0 with a carry: without a carry the next digit starts at the next 1 bit, with a carry
// (the previous digit was negative: 2^18 was borrowed) at the next 0 bit, which the carry turns into a 1; the digit is the 18
// bits from there, taken as a negative number when its top bit is set.  Digits are odd and at least 18 positions apart.
__global__ __launch_bounds__(1024) void k_naf_recode(const Fr *scalars, size_t n, int sfmt, size_t per_block, uint32_t *recs, uint32_t *blk_bins) {
    __shared__ uint32_t h[NBINS];
    __shared__ uint32_t limbs[1024 * 12];  // 8 limbs of the balanced scalar + zero padding, per thread (dynamic bit addressing)
    for (int b = threadIdx.x; b < NBINS; b += blockDim.x) h[b] = 0;
    __syncthreads();
    uint32_t *L = limbs + threadIdx.x * 12;
    size_t i0 = (size_t)blockIdx.x * per_block;
    size_t i1 = i0 + per_block < n ? i0 + per_block : n;
    for (size_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        uint32_t s[8];
        load_scalar(scalars, i, sfmt, s);
        uint32_t flip = 0;
        if (s[7] & 0x40000000u) {  // k >= 2^254: use r - k < 2^254 with every digit sign flipped
            uint64_t bw = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                uint64_t d = (uint64_t)FrParams::mod(k) - s[k] - bw;
                s[k] = (uint32_t)d;
                bw = (d >> 63) & 1u;
            }
            flip = 1;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) L[k] = s[k];
        L[8] = L[9] = L[10] = L[11] = 0;
        uint32_t d[NAF_MAX_DIGITS];
        const int cnt = naf18_digits(L, flip, d);
#pragma unroll
        for (int k = 0; k < NAF_MAX_DIGITS; k++) {
            const uint32_t r = k < cnt ? d[k] : 0u;
            recs[(size_t)k * n + i] = r;
            if (r) atomicAdd(&h[(r & 0x1ffffu) >> BIN_SHIFT], 1u);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < NBINS; b += blockDim.x) blk_bins[(size_t)blockIdx.x * NBINS + b] = h[b];
}

// k_bin_scatter for digit records: the packed LDS word carries the digit's ORDINAL (4 bits) where the window index was; the bit
// position (table row) is read back from the record when the entry is written out (the block read it a moment ago: L2).
template <class REC>
__global__ __launch_bounds__(1024) void k_bin_scatter_naf(const uint32_t *recs, size_t n, size_t per_block, const uint32_t *blk_off,
                                                          const uint32_t *bin_total, uint32_t *bin_base, uint32_t row_stride,
                                                          uint32_t idx_base, typename REC::T *rec) {
    uint32_t *cur = lds_u32, *cnt = cur + NBINS, *off = cnt + NBINS, *wsum = off + NBINS, *stage = wsum + 16;
    const uint32_t tid = threadIdx.x;
    {
        uint32_t tot;
        const uint32_t ex = block_scan_1024(bin_total[tid], wsum, &tot);
        cur[tid] = ex + blk_off[(size_t)blockIdx.x * NBINS + tid];
        if (blockIdx.x == 0) {
            bin_base[tid] = ex;
            if (tid == 0) bin_base[NBINS] = tot;
        }
    }
    const size_t i0 = (size_t)blockIdx.x * per_block;
    const size_t i1 = i0 + per_block < n ? i0 + per_block : n;
    for (size_t c0 = i0; c0 < i1; c0 += 1024) {
        cnt[tid] = 0;
        __syncthreads();
        uint32_t pk[NAF_MAX_DIGITS], rk[NAF_MAX_DIGITS];
#pragma unroll
        for (int k = 0; k < NAF_MAX_DIGITS; k++) {
            pk[k] = 0xffffffffu;
            if (c0 + tid < i1) {
                const uint32_t r = recs[(size_t)k * n + c0 + tid];
                if (r & NAF_REC_VALID) {
                    const uint32_t idx = r & 0x1ffffu, bin = idx >> BIN_SHIFT;
                    rk[k] = atomicAdd(&cnt[bin], 1u);
                    pk[k] = bin | ((idx & (BIN_BUCKETS - 1)) << 10) | (((r >> 17) & 1u) << 16) | ((uint32_t)k << 17) | (tid << 21);
                }
            }
        }
        __syncthreads();
        uint32_t tot;
        off[tid] = block_scan_1024(cnt[tid], wsum, &tot);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NAF_MAX_DIGITS; k++)
            if (pk[k] != 0xffffffffu) stage[off[pk[k] & (NBINS - 1)] + rk[k]] = pk[k];
        __syncthreads();
        const uint32_t ebase = idx_base + (uint32_t)c0;
        for (uint32_t p = tid; p < tot; p += 1024) {
            const uint32_t v = stage[p], bin = v & (NBINS - 1), t = v >> 21, k = (v >> 17) & 15u;
            const uint32_t row = (recs[(size_t)k * n + c0 + t] >> 18) & 0xffu;
            const uint32_t index = row * row_stride + ebase + t;
            rec[cur[bin] + p - off[bin]] = REC::pack(index, (v >> 16) & 1u, (v >> 10) & (BIN_BUCKETS - 1));
        }
        __syncthreads();
        cur[tid] += cnt[tid];
    }
}

// Level 2 (block = bin: bucket sizes -> total[], then the records -> entries[] in bucket order, chunk-sorted in LDS first so that
// each bucket's share of a chunk is one run of stores) lives in msm_wide.hip with the handling of oversized bins: sort2_level2.

// k_scan_b for the two-level sort: block j's first bucket starts at bin_base[4 j], M = bin_base[NBINS]
__global__ __launch_bounds__(SCAN_SEG) void k_scan_b_bins(const uint32_t *total, const uint32_t *bin_base, int B, int NB,
                                                          uint32_t *bucket_start, uint32_t *s1, MsmState *st, uint32_t slots,
                                                          uint32_t *ready) {
    KZG_SIDE_PRIO_STMT;
    __shared__ uint32_t lds[4];
    const uint32_t M = bin_base[NBINS];
    const uint32_t my_prefix = bin_base[blockIdx.x * (SCAN_SEG / BIN_BUCKETS)];
    uint32_t E = (M + slots - 1) / slots;
    if (E < KZG_ACCUM_MIN_CHUNK) E = KZG_ACCUM_MIN_CHUNK;
    const double rcp_e = 1.0 / (double)E;
    const int b = blockIdx.x * SCAN_SEG + threadIdx.x;
    const uint32_t cnt = b < B ? total[b] : 0u;
    uint32_t btot;
    const uint32_t start = my_prefix + block_scan_256(cnt, lds, &btot);
    const uint32_t f = (cnt != 0 && mod_u32(start, E, rcp_e) != 0) ? 1u : 0u;
    uint32_t ftot;
    const uint32_t fex = block_scan_256(f, lds, &ftot);
    if (threadIdx.x == 0) __hip_atomic_store(&ready[blockIdx.x], (ftot << 1) | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t prev = 0;
    if (threadIdx.x < blockIdx.x) {
        uint32_t w;
        while (((w = __hip_atomic_load(&ready[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 1u) == 0) __builtin_amdgcn_s_sleep(1);
        prev = w >> 1;
    }
    uint32_t before;
    block_scan_256(prev, lds, &before);
    if (b < B) {
        bucket_start[b] = start;
        s1[b] = before + fex + (start + E - 1) / E;
    }
    if (blockIdx.x == (unsigned)NB - 1 && threadIdx.x == 0) {
        bucket_start[B] = M;
        s1[B] = before + ftot + (M + E - 1) / E;
        st->M = M;
        st->E = E;
        st->ntasks = (M + E - 1) / E;
        st->ovf_tasks = 0;
    }
}

// ---------------------------------------------------------------------------------------------
// accumulation rounds
// ---------------------------------------------------------------------------------------------
// table30 entry: G1Affine30 = 2 x 13 limbs + pad = KZG_ROW_BYTES (112 B = 7 x 16 B by default); the 7 x 16 B that hold the
// coordinates are what gets loaded
__device__ __forceinline__ G1Affine30 load_entry_point30(const uint4 *table30, uint32_t ent) {
#if defined(KZG_TIMING_GATHER)
    // timing experiments only (wrong results): where the cost of the gathers comes from.  1: every gather inside the same 128 KiB (no
    // HBM, no TLB misses); 2: 512 lines, each in a 2 MiB page of its own (no HBM, TLB misses); 3: inside 64 MiB (Infinity Cache);
    // 4: inside 512 MiB (HBM, a quarter of the pages)
    const uint32_t e_ = ent & 0x7fffffffu;
    const size_t row_ = KZG_TIMING_GATHER == 1 ? (e_ & 0x3ffu) : KZG_TIMING_GATHER == 2 ? ((size_t)(e_ & 0x1ffu) << 14)
                        : KZG_TIMING_GATHER == 3 ? (e_ & 0x7ffffu) : (e_ & 0x3fffffu);
    const uint4 *src = table30 + row_ * (KZG_ROW_BYTES / 16);
#else
    const uint4 *src = table30 + (size_t)(ent & 0x7fffffffu) * (KZG_ROW_BYTES / 16);
#endif
    G1Affine30 p;
    uint4 *dst = reinterpret_cast<uint4 *>(&p);
#if defined(KZG_GATHER_NT)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int k = 0; k < 7; k++) {   // A/B: every table line is read once per MSM
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src) + k);
        dst[k] = make_uint4(v.x, v.y, v.z, v.w);
    }
#else
#pragma unroll
    for (int k = 0; k < 7; k++) dst[k] = src[k];
#endif
    return p;
}

// round 1 (dominant kernel): thread s folds its E consecutive sorted entries with XYZZ mixed adds in the
// signed 30-bit field representation (curve30.h), gathering each precomputed point from the
// resident 30-bit table (next point prefetched under the add), and writes one partial per bucket it
// touches.  Every thread has the same amount of work,
// so the kernel ends without a straggler round.
__global__ __launch_bounds__(256, KZG_ACCUM_WAVES) void k_accum_affine(const uint32_t *entries, const uint32_t *bucket_start,
                                                      const uint32_t *s1, int B, const uint4 *table30,
                                                      MsmPoint *out, const MsmState *st) {
    const uint32_t E = st->E, M = st->M;
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t lo64 = (uint64_t)s * E;
    if (lo64 >= M) return;
    const uint32_t lo = (uint32_t)lo64;
    const uint32_t hi = (M - lo < E) ? M : lo + E;
    // bucket containing entry lo: start[b] <= lo < start[b+1]
    uint32_t bl = 0, bh = (uint32_t)B;  // start[bl] <= lo < start[bh]
    while (bh - bl > 1) {
        uint32_t mid = (bl + bh) >> 1;
        if (bucket_start[mid] <= lo) bl = mid; else bh = mid;
    }
    uint32_t b = bl;
    uint32_t bend = bucket_start[b + 1];
    // first output slot: run starts before lo = s multiples of E + non-aligned non-empty bucket starts < lo
    uint32_t f_next = s1[b + 1] - (bucket_start[b + 1] + E - 1) / E;  // F[b+1]
    uint32_t pos = s + f_next;
    uint32_t ent = entries[lo];
    G1Affine30 cur = load_entry_point30(table30, ent);
    G1Xyzz30 acc = g1_from_affine30(cur, ent >> 31);
    // the entry words run one step ahead of the gathers (ent_next = entry k+1 while entry k is being added), so a gather's address
    // never waits for the load of its own index
    uint32_t ent_next = 0;
    if (lo + 1 < hi) {
        ent = entries[lo + 1];
        cur = load_entry_point30(table30, ent);
    }
#ifndef KZG_NO_ENTRY_LOOKAHEAD
    if (lo + 2 < hi) ent_next = entries[lo + 2];
#endif
    for (uint32_t k = lo + 1; k < hi; k++) {
        // `cur` / `ent` hold entry k.  mode 0: mixed add; 1: restart the accumulator from cur; 2: skip (identity point)
        const bool neg_k = (ent >> 31) != 0;
        const uint32_t ent_k = ent;
        int mode = 0;
        Madd30Mid mid;
        if (k == bend) {  // bucket boundary: flush and restart
            out[pos++] = g1_normalize30(acc);
            do {
                b++;
                bend = bucket_start[b + 1];
            } while (bend <= k);
            mode = 1;
        } else if (cur.is_inf_table()) {
            mode = 2;
        } else if (acc.inf) {
            mode = 1;
        }
        if (mode == 0) mid = g1_madd30_phase1(acc, cur, neg_k);
        if (mode == 1) acc = g1_from_affine30(cur, neg_k);
        // entry k's point is dead now: start the gather of entry k+1 into the same registers; its latency
        // hides under the eight remaining multiplies of this addition
        if (k + 1 < hi) {
#ifndef KZG_NO_ENTRY_LOOKAHEAD
            ent = ent_next;
            if (k + 2 < hi) ent_next = entries[k + 2];
#else
            ent = entries[k + 1];
#endif
            cur = load_entry_point30(table30, ent);
        }
        if (mode == 0) acc = g1_madd30_phase2(acc, mid, neg_k, [&]() { return load_entry_point30(table30, ent_k); });
    }
    out[pos] = g1_normalize30(acc);
}

__global__ __launch_bounds__(256) void k_sum_level(const MsmPoint *in, uint32_t count, int L, MsmPoint *out) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t nout = (count + L - 1) / L;
    if (t >= nout) return;
    uint32_t s = t * L, e = s + L < count ? s + L : count;
    MsmPoint acc = in[s];
    for (uint32_t k = s + 1; k < e; k++) acc = g1_add30(acc, in[k]);
    out[t] = acc;
}

// output formatting (Curve::to_affine + serialisation): emit.h
__device__ __forceinline__ size_t format_bytes_dev(int fmt) {
    return fmt == KZG_G1_JACOBIAN_MONT_144 ? 144 : fmt == KZG_G1_ZCASH_COMPRESSED_48 ? 48 : 96;
}

__global__ __launch_bounds__(64) void k_emit_points(const MsmPoint *pts, size_t count, size_t stride_pts, uint8_t *out, int fmt) {
    KZG_SIDE_PRIO_STMT;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    emit_one(pts[i * stride_pts], out + i * format_bytes_dev(fmt), fmt);
}

// out[g] = sum_{i < count} pts[g * gstride + i * istride]   (count is small: one partial per GPU)
__global__ __launch_bounds__(64) void k_sum_groups(const MsmPoint *pts, uint32_t count, uint32_t groups, size_t gstride,
                                                   size_t istride, MsmPoint *out) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= groups) return;
    MsmPoint acc = MsmPoint::infinity();
    for (uint32_t i = 0; i < count; i++) acc = g1_add30(acc, pts[(size_t)g * gstride + (size_t)i * istride]);
    out[g] = acc;
}

int sum_groups_emit(kzg_ctx *ctx, int lane, const MsmPoint *d_pts, size_t count, size_t groups, size_t gstride, size_t istride,
                    MsmPoint *d_tmp, void *d_out, int ofmt) {
    hipStream_t st = ctx->lanes[lane].stream;
    KZG_LAUNCH(ctx, st, "k_sum_groups", k_sum_groups, (unsigned)((groups + 63) / 64), 64, 0, d_pts, (uint32_t)count,
               (uint32_t)groups, gstride, istride, d_tmp);
    if (!d_out) return KZG_OK;  // the caller converts d_tmp itself (a few host-bound results: emit.h on the calling thread)
    KZG_LAUNCH(ctx, st, "k_emit_points", k_emit_points, (unsigned)((groups + 63) / 64), 64, 0, d_tmp, groups, (size_t)1,
               (uint8_t *)d_out, ofmt);
    return KZG_OK;
}

__global__ __launch_bounds__(256) void k_points_to30(const G1Xyzz *in, MsmPoint *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = g1_xyzz_to30(in[i]);
}
__global__ __launch_bounds__(256) void k_points_from30(const MsmPoint *in, G1Xyzz *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = g1_xyzz_from30(in[i]);
}
__global__ void k_point_set_infinity(MsmPoint *p) { *p = MsmPoint::infinity(); }

int points_to30(kzg_ctx *ctx, hipStream_t st, const G1Xyzz *d_in, MsmPoint *d_out, size_t n) {
    if (n) KZG_LAUNCH(ctx, st, "k_points_to30", k_points_to30, (unsigned)((n + 255) / 256), 256, 0, d_in, d_out, n);
    return KZG_OK;
}
int points_from30(kzg_ctx *ctx, hipStream_t st, const MsmPoint *d_in, G1Xyzz *d_out, size_t n) {
    if (n) KZG_LAUNCH(ctx, st, "k_points_from30", k_points_from30, (unsigned)((n + 255) / 256), 256, 0, d_in, d_out, n);
    return KZG_OK;
}
int point_set_infinity(kzg_ctx *ctx, hipStream_t st, MsmPoint *d_pt) {
    KZG_LAUNCH(ctx, st, "k_point_set_infinity", k_point_set_infinity, 1, 1, 0, d_pt);
    return KZG_OK;
}

size_t point_format_bytes(int fmt) {
    switch (fmt) {
        case KZG_G1_AFFINE_MONT_96: return 96;
        case KZG_G1_JACOBIAN_MONT_144: return 144;
        case KZG_G1_ZCASH_UNCOMPRESSED_96: return 96;
        case KZG_G1_ZCASH_COMPRESSED_48: return 48;
        default: return 0;
    }
}

int emit_point(kzg_ctx *ctx, int lane, const MsmPoint *d_point, void *d_out, int ofmt) {
    if (!point_format_bytes(ofmt)) return fail(ctx, KZG_ERR_SHAPE, "unknown G1 output format");
    hipStream_t st = ctx->lanes[lane].stream;
    KZG_LAUNCH(ctx, st, "k_emit_points", k_emit_points, 1, 64, 0, d_point, (size_t)1, (size_t)1, (uint8_t *)d_out,
               ofmt);
    return KZG_OK;
}

// ---------------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------------
static int sort_blocks(size_t n) {
    size_t g = (n + 2047) / 2048;
    if (g < 1) g = 1;
    if (g > 256) g = 256;
    return (int)g;
}

// level 1 of the two-level sort keeps 1024 cursors in LDS, so two blocks fit a CU
static int sort2_blocks(size_t n) {
    size_t g = (n + 2047) / 2048;
    if (g < 1) g = 1;
    if (g > SORT2_MAX_BLOCKS) g = SORT2_MAX_BLOCKS;
    return (int)g;
}

size_t sum_points_scratch_count(size_t count) { return (count + SUM_L - 1) / SUM_L + 64; }

struct MsmLayout {
    int B, G, G2;
    size_t M_max, T1_max;
    size_t off_bins, off_bin_base, off_recs, off_seg, off_hv;
    size_t off_blk_hist, off_total, off_local, off_agg, off_bucket_start, off_s1, off_state, off_entries, off_bufA, off_bufB, off_tail,
        off_pass, bytes;
    TailLayout tail;
};

static MsmLayout msm_layout(const kzg_srs *srs, size_t n) {
    MsmLayout L;
    L.B = 1 << (srs->c - 1);
    L.G = sort_blocks(n);
    L.M_max = n * srs_entries_per_scalar(srs);  // entries of one pass
    L.T1_max = (size_t)ACC_SLOTS + L.B + 1;  // round-1 partials: one per thread slot + one per bucket boundary
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t r = o;
        o = align_up(o + bytes, 256);
        return r;
    };
    // c = 17: the region holds either the per-(block, bucket) counters of the single-pass sort or the level-1 records of the
    // two-level sort (8 B per entry), whichever the context runs
    L.G2 = sort2_blocks(n);
    size_t sort_bytes = (size_t)L.G * L.B * 4;
    if (srs->narrow17 && L.M_max * 8 > sort_bytes) sort_bytes = L.M_max * 8;
    if (srs->sort20) sort_bytes = L.M_max * 8;  // c = 20 has no single-pass mode here: the records only
    L.off_blk_hist = take(sort_bytes);
    L.off_bins = take((size_t)L.G2 * NBINS * 4);
    L.off_bin_base = take((2 * NBINS + 1) * 4);  // bin starts, then bin sizes
    L.off_recs = take(srs->naf ? (size_t)NAF_MAX_DIGITS * n * 4 : 0);  // positional tables: the digit records of the scalars
    L.off_seg = take(srs->sort20 ? 3 * 256 * 4 : 0);  // segment sums / maxima / total of the round-1 layout scan (wide_s1_layout)
    L.off_hv = take(srs->sort20 ? sort2_hv_bytes(SORT20_BUCKETS) : (srs->narrow17 || srs->naf) ? sort2_hv_bytes(BIN_BUCKETS) : 0);  // bins sorted in slices (msm_wide.hip)
    L.off_total = take((size_t)L.B * 4);
    L.off_local = take((size_t)L.B * 4);
    L.off_agg = take(2 * SCAN_SEG * 4 + 64);  // block aggregates + the chained flag counts
    L.off_bucket_start = take((size_t)(L.B + 1) * 4);
    L.off_s1 = take((size_t)(L.B + 1) * 4);
    L.off_state = take(sizeof(MsmState));
    L.off_entries = take(L.M_max * 4 + 16);
    L.off_bufA = take(L.T1_max * sizeof(MsmPoint));
    L.off_bufB = take(L.T1_max * sizeof(MsmPoint));  // slice sums of overflowing buckets, at the bucket's own offsets
    L.tail = tail_layout(L.B, L.T1_max);
    L.off_tail = take(L.tail.bytes);
    L.off_pass = take(((size_t)(srs->W + srs->rows - 1) / srs->rows + 1) * sizeof(MsmPoint));
    L.bytes = o;
    return L;
}

// ---- wide mode layout / orchestration ----
struct WideLayout {
    int B_lo, nhi, Btot, G;
    size_t M_max, T1_max;
    size_t off_blk_hist, off_total, off_local, off_agg, off_lo_start, off_s1_lo, off_state, off_entries1, off_entries2, off_bucket_start, off_s1,
        off_blockcnt, off_binbase, off_segsums, off_segmaxs, off_segtotal, off_bufA, off_bufB, off_tail, bytes;
    TailLayout tail;
};

static WideLayout wide_layout(const kzg_srs *srs, size_t n) {
    WideLayout L;
    L.B_lo = 1 << WIDE_LO_BITS;
    L.nhi = 1 << (srs->c - 1 - WIDE_LO_BITS);
    L.Btot = L.nhi * L.B_lo;
    L.G = sort_blocks(n);
    L.M_max = n * (size_t)srs->W;
    L.T1_max = (size_t)ACC_SLOTS + L.Btot + 1;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t r = o;
        o = align_up(o + bytes, 256);
        return r;
    };
    L.off_blk_hist = take((size_t)L.G * L.B_lo * 4);
    L.off_total = take((size_t)L.B_lo * 4);
    L.off_local = take((size_t)L.B_lo * 4);
    L.off_agg = take(2 * SCAN_SEG * 4 + 64);  // block aggregates + the chained flag counts
    L.off_lo_start = take((size_t)(L.B_lo + 1) * 4);
    L.off_s1_lo = take((size_t)(L.B_lo + 1) * 4);
    L.off_state = take(sizeof(MsmState));
    L.off_entries1 = take(L.M_max * 4 + 16);
    L.off_entries2 = take(L.M_max * 4 + 16);
    L.off_bucket_start = take((size_t)(L.Btot + 1) * 4);
    L.off_s1 = take((size_t)(L.Btot + 1) * 4);
    L.off_blockcnt = take(16 * HI_BLOCKS * 4);
    L.off_binbase = take(16 * HI_BLOCKS * 4);
    L.off_segsums = take(256 * 4);
    L.off_segmaxs = take(256 * 4);
    L.off_segtotal = take(256);
    L.off_bufA = take(L.T1_max * sizeof(MsmPoint));
    L.off_bufB = take(L.T1_max * sizeof(MsmPoint));
    L.tail = tail_layout(L.Btot, L.T1_max);
    L.off_tail = take(L.tail.bytes);
    L.bytes = o;
    return L;
}

// round-1 partials the equal split is expected to emit (one per thread that gets work + one per bucket boundary inside a chunk)
static size_t expected_partials(size_t M_max, uint32_t slots, int B) {
    size_t thr = M_max / KZG_ACCUM_MIN_CHUNK + 1 < (size_t)slots ? M_max / KZG_ACCUM_MIN_CHUNK + 1 : (size_t)slots;
    return thr + (size_t)B;
}

static int msm_run_wide(kzg_ctx *ctx, int lane, const kzg_srs *srs, size_t offset, const void *d_scalars, size_t n, int sfmt,
                        MsmPoint **d_result) {
    if ((uint64_t)srs->W * srs->npad >= (1ull << WIDE_HI_SHIFT))
        return fail(ctx, KZG_ERR_SHAPE, "SRS too large for the wide-window entry encoding (window_bits 18..20: W * n < 2^27)");
    hipStream_t st = ctx->lanes[lane].stream;
    const MsmMode mm = ctx->lanes[lane].mode;
    WideLayout L = wide_layout(srs, n ? n : 1);
    char *base = (char *)lane_alloc(ctx, lane, L.bytes);
    if (!base) return fail(ctx, KZG_ERR_ALLOC, "MSM workspace not reserved");
    if (n == 0) {
        *d_result = (MsmPoint *)(base + L.off_tail + L.tail.off_result);
        return point_set_infinity(ctx, st, *d_result);
    }
    if (!ctx->attr_msm_set) {
        KZG_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)k_hist, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096));
        KZG_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)k_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096));
        KZG_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)k_bin_scatter<Rec4>, hipFuncAttributeMaxDynamicSharedMemorySize, BIN_SCATTER_LDS));
        KZG_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)k_bin_scatter<Rec8>, hipFuncAttributeMaxDynamicSharedMemorySize, BIN_SCATTER_LDS));
        ctx->attr_msm_set = true;
    }
    const int B_lo = L.B_lo, Btot = L.Btot, nhi = L.nhi, G = L.G, c = srs->c, W = srs->W;
    uint32_t *blk_hist = (uint32_t *)(base + L.off_blk_hist), *total = (uint32_t *)(base + L.off_total);
    uint32_t *lo_start = (uint32_t *)(base + L.off_lo_start), *s1_lo = (uint32_t *)(base + L.off_s1_lo);
    MsmState *state = (MsmState *)(base + L.off_state);
    uint32_t *entries1 = (uint32_t *)(base + L.off_entries1), *entries2 = (uint32_t *)(base + L.off_entries2);
    uint32_t *bucket_start = (uint32_t *)(base + L.off_bucket_start), *s1 = (uint32_t *)(base + L.off_s1);
    uint32_t *blockcnt = (uint32_t *)(base + L.off_blockcnt), *binbase = (uint32_t *)(base + L.off_binbase);
    uint32_t *segsums = (uint32_t *)(base + L.off_segsums), *segmaxs = (uint32_t *)(base + L.off_segmaxs);
    uint32_t *segtotal = (uint32_t *)(base + L.off_segtotal);
    MsmPoint *bufA = (MsmPoint *)(base + L.off_bufA), *bufB = (MsmPoint *)(base + L.off_bufB);
    const Fr *sc = (const Fr *)d_scalars;
    size_t per_block = (n + G - 1) / G;
    size_t lds_bytes = (size_t)B_lo * 4;

    // pass 1: LDS counting sort by the low 15 bucket bits (hi rides in the entry word)
    KZG_LAUNCH(ctx, st, "k_hist", k_hist, G, mm.sort_threads, lds_bytes, sc, n, sfmt, c, W, B_lo, per_block, blk_hist, 0, 0, W);
    const uint32_t slots = (uint32_t)mm.accum_blocks * 256u;
    KZG_TRY(scan_run(ctx, st, blk_hist, G, B_lo, total, (uint32_t *)(base + L.off_local), (uint32_t *)(base + L.off_agg), lo_start, s1_lo,
                     state, slots));  // M, E, ntasks
    KZG_LAUNCH(ctx, st, "k_scatter", k_scatter, G, mm.sort_threads, lds_bytes, sc, n, sfmt, c, W, B_lo, per_block, blk_hist, lo_start,
               (uint32_t)srs->npad, (uint32_t)offset, entries1, 1, 0, W);
    // pass 2: stable partition by hi; also yields the starts of all nhi * 2^15 buckets
    KZG_TRY(wide_sort_pass2(ctx, st, entries1, state, nhi, blockcnt, binbase, lo_start, B_lo, entries2, bucket_start));
    // equal-split layout of round 1 over the full bucket set
    KZG_TRY(wide_s1_layout(ctx, st, bucket_start, Btot, state, segsums, segmaxs, segtotal, s1));
    size_t thr1 = L.M_max / KZG_ACCUM_MIN_CHUNK + 1 < (size_t)slots ? L.M_max / KZG_ACCUM_MIN_CHUNK + 1 : (size_t)slots;
    unsigned grid1 = (unsigned)((thr1 + 255) / 256);
    KZG_LAUNCH(ctx, st, "k_accum_affine", k_accum_affine, grid1, 256, 0, entries2, bucket_start, s1, Btot,
               (const uint4 *)srs->table30, bufA, state);
    return msm_tail_run(ctx, st, mm, bufA, bufB, s1, Btot, expected_partials(L.M_max, slots, Btot), state, base + L.off_tail, L.tail,
                        d_result);
}

size_t msm_workspace_bytes(const kzg_srs *srs, size_t n) {
    if (srs->sort20) {  // either path, by option sort_single_pass at call time (the older one only below its size limit)
        const size_t a = msm_layout(srs, n ? n : 1).bytes;
        const size_t b = (uint64_t)srs->W * srs->npad < (1ull << WIDE_HI_SHIFT) ? wide_layout(srs, n ? n : 1).bytes : 0;
        return a > b ? a : b;
    }
    return (srs->c > 16 && !srs->narrow17) ? wide_layout(srs, n ? n : 1).bytes : msm_layout(srs, n ? n : 1).bytes;
}

int sum_level_run(kzg_ctx *ctx, hipStream_t st, const MsmPoint *in, uint32_t count, int L, MsmPoint *out) {
    size_t nout = (count + L - 1) / L;
    KZG_LAUNCH(ctx, st, "k_sum_level", k_sum_level, (unsigned)((nout + TAIL_THREADS - 1) / TAIL_THREADS), TAIL_THREADS, 0, in, count, L, out);
    return KZG_OK;
}

int sum_points_run(kzg_ctx *ctx, int lane, MsmPoint *d_points, size_t count, MsmPoint *d_scratch, MsmPoint **d_result) {
    hipStream_t st = ctx->lanes[lane].stream;
    MsmPoint *in = d_points;
    MsmPoint *bufs[2] = {d_scratch, d_scratch + sum_points_scratch_count(count)};
    int which = 0;
    while (count > 1) {
        size_t nout = (count + SUM_L - 1) / SUM_L;
        KZG_LAUNCH(ctx, st, "k_sum_level", k_sum_level, (unsigned)((nout + TAIL_THREADS - 1) / TAIL_THREADS), TAIL_THREADS, 0, in, (uint32_t)count,
                   SUM_L, bufs[which]);
        in = bufs[which];
        which ^= 1;
        count = nout;
    }
    *d_result = in;
    return KZG_OK;
}

// result = sum_p 2^(shift p) S_p  (Horner from the top pass down: shift doublings + one addition per pass)
__global__ __launch_bounds__(64) void k_combine_passes(const MsmPoint *S, int passes, int shift, MsmPoint *result) {
    if (threadIdx.x != 0) return;
    MsmPoint acc = S[passes - 1];
    for (int p = passes - 2; p >= 0; p--) {
        for (int k = 0; k < shift; k++) acc = g1_dbl30(acc);
        acc = g1_add30(acc, S[p]);
    }
    *result = acc;
}

// One MSM on the lane's stream: counting sort, bucket accumulation (on `accum_stream` when the batched pipeline runs every
// accumulation kernel on dedicated streams: then `sorted_ev` / `accum_ev` order the two), tail (msm_tail.hip).
static int msm_run_narrow(kzg_ctx *ctx, int lane, const kzg_srs *srs, size_t offset, const void *d_scalars, size_t n, int sfmt,
                          MsmPoint **d_result, hipStream_t accum_stream, hipEvent_t sorted_ev, hipEvent_t accum_ev, MsmPending *defer) {
    hipStream_t st = ctx->lanes[lane].stream;
    const MsmMode mm = ctx->lanes[lane].mode;
    MsmLayout L = msm_layout(srs, n ? n : 1);
    char *base = (char *)lane_alloc(ctx, lane, L.bytes);
    if (!base) return fail(ctx, KZG_ERR_ALLOC, "MSM workspace not reserved");
    if (n == 0) {
        *d_result = (MsmPoint *)(base + L.off_tail + L.tail.off_result);
        return point_set_infinity(ctx, st, *d_result);
    }
    if (!ctx->attr_msm_set) {  // per context (= per device): the LDS opt-in is a per-device function attribute
        KZG_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)k_hist, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096));
        KZG_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)k_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096));
        KZG_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)k_bin_scatter<Rec4>, hipFuncAttributeMaxDynamicSharedMemorySize, BIN_SCATTER_LDS));
        KZG_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)k_bin_scatter<Rec8>, hipFuncAttributeMaxDynamicSharedMemorySize, BIN_SCATTER_LDS));
        KZG_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)k_bin_scatter_naf<Rec4>, hipFuncAttributeMaxDynamicSharedMemorySize, BIN_SCATTER_LDS));
        KZG_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)k_bin_scatter_naf<Rec8>, hipFuncAttributeMaxDynamicSharedMemorySize, BIN_SCATTER_LDS));
        ctx->attr_msm_set = true;
    }
    const int B = L.B, G = L.G, c = srs->c, W = srs->W;
    uint32_t *blk_hist = (uint32_t *)(base + L.off_blk_hist);
    uint32_t *total = (uint32_t *)(base + L.off_total);
    uint32_t *bucket_start = (uint32_t *)(base + L.off_bucket_start);
    uint32_t *s1 = (uint32_t *)(base + L.off_s1);
    MsmState *state = (MsmState *)(base + L.off_state);
    uint32_t *entries = (uint32_t *)(base + L.off_entries);
    MsmPoint *bufA = (MsmPoint *)(base + L.off_bufA), *bufB = (MsmPoint *)(base + L.off_bufB);
    const Fr *sc = (const Fr *)d_scalars;
    size_t per_block = (n + G - 1) / G;
    const int mode = srs->narrow17 ? 2 : 0;
    size_t lds_bytes = srs->narrow17 ? (size_t)B * 2 : (size_t)B * 4;  // c = 17: the counters / cursors of half the buckets per walk

    // One pass per `rows` windows: the table holds rows 0 .. rows-1 (2^(c w') P), so pass p reduces the digits of windows
    // [p rows, (p+1) rows) to S_p = sum_i (sum_{w'} d_{i, p rows + w'} 2^(c w')) P_i and the result is sum_p 2^(c rows p) S_p.
    // rows == W (the default): one pass, no doubling chain.
    const int rows = srs->naf ? W : srs->rows, passes = (W + rows - 1) / rows;  // positional tables: one pass, always
    MsmPoint *pass_res = (MsmPoint *)(base + L.off_pass);
    // resident threads k_accum_affine is split over; a small MSM inside a pipeline takes a smaller grid (common.h opt_accum_blocks_small)
    const bool in_pipeline = accum_stream && accum_stream != st;
    const int acc_blocks = in_pipeline && ctx->msm_small(L.M_max) && ctx->opt_accum_blocks_small < mm.accum_blocks ? ctx->opt_accum_blocks_small
                                                                                                                   : mm.accum_blocks;
    const uint32_t slots = (uint32_t)acc_blocks * 256u;
    uint32_t *hvp = (uint32_t *)(base + L.off_hv);
    const uint32_t heavy_seq = ++ctx->lanes[lane].heavy_seq;  // (the lane is leased: no other thread touches it)
    uint32_t *lane_heavy = ctx->d_lane_heavy + lane;
    // oversized sort bins in slices: always / never by option, otherwise while some lane's last plan had any
    bool sliced = ctx->opt_heavy_bins == 1;
    {
        const uint64_t now = ctx->msm_count.fetch_add(1, std::memory_order_relaxed) + 1, last = ctx->heavy_last.load(std::memory_order_relaxed);
        if (ctx->opt_heavy_bins == 0 && last != 0 && now - last <= (uint64_t)KZG_HEAVY_WINDOW) sliced = true;
    }
    for (int p = 0; p < passes; p++) {
        const int w_lo = p * rows, w_hi = (p + 1) * rows < W ? (p + 1) * rows : W;
        if (srs->naf) {
            // positional tables: recode (+ level-1 histogram), then the two-level sort on the digit records
            const int G2 = L.G2;
            const size_t per2 = (n + G2 - 1) / G2;
            uint32_t *bins = (uint32_t *)(base + L.off_bins), *bin_base = (uint32_t *)(base + L.off_bin_base);
            uint32_t *ready = (uint32_t *)(base + L.off_agg) + SCAN_SEG;
            uint32_t *recs = (uint32_t *)(base + L.off_recs);
            KZG_LAUNCH(ctx, st, "k_naf_recode", k_naf_recode, G2, 1024, 0, sc, n, sfmt, per2, recs, bins);
            uint32_t *bin_total = bin_base + NBINS + 1;
            KZG_LAUNCH(ctx, st, "k_bin_scan", k_bin_scan, NBINS / 64, 1024, 0, bins, G2, bin_total, ready);
            if ((uint64_t)srs->rows * srs->npad < REC4_MAX_INDEX) {
                uint32_t *rec = (uint32_t *)blk_hist;
                KZG_LAUNCH(ctx, st, "k_bin_scatter", k_bin_scatter_naf<Rec4>, G2, 1024, BIN_SCATTER_LDS, recs, n, per2, bins, bin_total, bin_base,
                           (uint32_t)srs->npad, (uint32_t)offset, rec);
                KZG_TRY(sort2_level2(ctx, st, 4, rec, bin_base, bin_total, hvp, entries, total, bucket_start, nullptr, slots, sliced, lane_heavy, heavy_seq));
            } else {
                uint2 *rec = (uint2 *)blk_hist;
                KZG_LAUNCH(ctx, st, "k_bin_scatter", k_bin_scatter_naf<Rec8>, G2, 1024, BIN_SCATTER_LDS, recs, n, per2, bins, bin_total, bin_base,
                           (uint32_t)srs->npad, (uint32_t)offset, rec);
                KZG_TRY(sort2_level2(ctx, st, 8, rec, bin_base, bin_total, hvp, entries, total, bucket_start, nullptr, slots, sliced, lane_heavy, heavy_seq));
            }
            KZG_LAUNCH(ctx, st, "k_scan_b_bins", k_scan_b_bins, B / SCAN_SEG, SCAN_SEG, 0, total, bin_base, B, B / SCAN_SEG, bucket_start, s1,
                       state, slots, ready);
        } else if (srs->sort20) {
            const int G2 = L.G2;
            uint32_t *bins = (uint32_t *)(base + L.off_bins), *bin_base = (uint32_t *)(base + L.off_bin_base);
            uint32_t *ready = (uint32_t *)(base + L.off_agg) + SCAN_SEG, *bin_total = bin_base + NBINS + 1;
            uint32_t *seg = (uint32_t *)(base + L.off_seg);
            KZG_TRY(sort20_hist(ctx, st, sc, n, sfmt, G2, bins));
            KZG_LAUNCH(ctx, st, "k_bin_scan", k_bin_scan, NBINS / 64, 1024, 0, bins, G2, bin_total, ready);
            KZG_TRY(sort20_place(ctx, st, sc, n, sfmt, G2, bins, bin_total, bin_base, (uint32_t)srs->npad, (uint32_t)offset, blk_hist, entries,
                                 bucket_start, s1, state, slots, seg, seg + 256, seg + 512, hvp, sliced, lane_heavy, heavy_seq));
        } else if (srs->narrow17 && !ctx->opt_sort_single) {
            const int G2 = L.G2;
            const size_t per2 = (n + G2 - 1) / G2;
            uint32_t *bins = (uint32_t *)(base + L.off_bins), *bin_base = (uint32_t *)(base + L.off_bin_base);
            uint32_t *ready = (uint32_t *)(base + L.off_agg) + SCAN_SEG;
            KZG_LAUNCH(ctx, st, "k_bin_hist", k_bin_hist, G2, 1024, 0, sc, n, sfmt, per2, bins, w_lo, w_hi);
            uint32_t *bin_total = bin_base + NBINS + 1;
            KZG_LAUNCH(ctx, st, "k_bin_scan", k_bin_scan, NBINS / 64, 1024, 0, bins, G2, bin_total, ready);
            if ((uint64_t)srs->rows * srs->npad < REC4_MAX_INDEX) {
                uint32_t *rec = (uint32_t *)blk_hist;
                KZG_LAUNCH(ctx, st, "k_bin_scatter", k_bin_scatter<Rec4>, G2, 1024, BIN_SCATTER_LDS, sc, n, sfmt, per2, bins, bin_total,
                           bin_base, (uint32_t)srs->npad, (uint32_t)offset, rec, w_lo, w_hi);
                KZG_TRY(sort2_level2(ctx, st, 4, rec, bin_base, bin_total, hvp, entries, total, bucket_start, nullptr, slots, sliced, lane_heavy, heavy_seq));
            } else {
                uint2 *rec = (uint2 *)blk_hist;
                KZG_LAUNCH(ctx, st, "k_bin_scatter", k_bin_scatter<Rec8>, G2, 1024, BIN_SCATTER_LDS, sc, n, sfmt, per2, bins, bin_total,
                           bin_base, (uint32_t)srs->npad, (uint32_t)offset, rec, w_lo, w_hi);
                KZG_TRY(sort2_level2(ctx, st, 8, rec, bin_base, bin_total, hvp, entries, total, bucket_start, nullptr, slots, sliced, lane_heavy, heavy_seq));
            }
            KZG_LAUNCH(ctx, st, "k_scan_b_bins", k_scan_b_bins, B / SCAN_SEG, SCAN_SEG, 0, total, bin_base, B, B / SCAN_SEG, bucket_start, s1,
                       state, slots, ready);
        } else {
            KZG_LAUNCH(ctx, st, "k_hist", k_hist, G, mm.sort_threads, lds_bytes, sc, n, sfmt, c, W, B, per_block, blk_hist, mode, w_lo,
                       w_hi);
            // s1[0 .. B] = per-bucket start offsets of the round-1 output list
            KZG_TRY(scan_run(ctx, st, blk_hist, G, B, total, (uint32_t *)(base + L.off_local), (uint32_t *)(base + L.off_agg), bucket_start,
                             s1, state, slots));
            KZG_LAUNCH(ctx, st, "k_scatter", k_scatter, G, mm.sort_threads, lds_bytes, sc, n, sfmt, c, W, B, per_block, blk_hist,
                       bucket_start, (uint32_t)srs->npad, (uint32_t)offset, entries, mode, w_lo, w_hi);
        }
        // grid covers ceil(M/E) <= max(ACC_SLOTS, M_max/8) threads
        size_t thr1 = L.M_max / KZG_ACCUM_MIN_CHUNK + 1 < (size_t)slots ? L.M_max / KZG_ACCUM_MIN_CHUNK + 1 : (size_t)slots;
        unsigned grid1 = (unsigned)((thr1 + 255) / 256);
        if (accum_stream && accum_stream != st) {
            // wait / launch / record on the shared FIFO stream is one unit: with concurrent callers the three must not
            // interleave with another lane's (a kernel would pick up the other lane's dependency, a lane the other kernel's)
            KZG_HIP_CHECK(ctx, hipEventRecord(sorted_ev, st));
            std::lock_guard<std::mutex> alk(ctx->accum_mu);
            KZG_HIP_CHECK(ctx, hipStreamWaitEvent(accum_stream, sorted_ev, 0));
            KZG_LAUNCH(ctx, accum_stream, "k_accum_affine", k_accum_affine, grid1, 256, 0, entries, bucket_start, s1, B,
                       (const uint4 *)srs->table30, bufA, state);
            KZG_HIP_CHECK(ctx, hipEventRecord(accum_ev, accum_stream));
            if (defer && passes == 1) {  // the caller enqueues the wait and the tail later (msm_finish)
                static_assert(sizeof(TailLayout) == sizeof(defer->tail_off), "MsmPending::tail_off mirrors TailLayout");
                defer->active = true;
                defer->st = st;
                defer->accum_ev = accum_ev;
                defer->mm = mm;
                defer->part = bufA;
                defer->scratch = bufB;
                defer->s1 = s1;
                defer->B = B;
                defer->expected = expected_partials(L.M_max, slots, B);
                defer->state = state;
                defer->tail_base = base + L.off_tail;
                memcpy(defer->tail_off, &L.tail, sizeof(TailLayout));
                defer->odd = srs->naf != 0;
                *d_result = nullptr;
                return KZG_OK;
            }
            KZG_HIP_CHECK(ctx, hipStreamWaitEvent(st, accum_ev, 0));
        } else {
            KZG_LAUNCH(ctx, st, "k_accum_affine", k_accum_affine, grid1, 256, 0, entries, bucket_start, s1, B,
                       (const uint4 *)srs->table30, bufA, state);
        }
        MsmPoint *res = nullptr;
        KZG_TRY(msm_tail_run(ctx, st, mm, bufA, bufB, s1, B, expected_partials(L.M_max, slots, B), state, base + L.off_tail, L.tail, &res,
                             srs->naf != 0));
        if (passes == 1) {
            *d_result = res;
            return KZG_OK;
        }
        KZG_HIP_CHECK(ctx, hipMemcpyAsync(pass_res + p, res, sizeof(MsmPoint), hipMemcpyDeviceToDevice, st));
    }
    KZG_LAUNCH(ctx, st, "k_combine_passes", k_combine_passes, 1, 64, 0, pass_res, passes, c * rows, pass_res + passes);
    *d_result = pass_res + passes;
    return KZG_OK;
}

int msm_finish(kzg_ctx *ctx, MsmPending &pd, MsmPoint **d_result) {
    if (!pd.active) {
        *d_result = pd.result;
        return KZG_OK;
    }
    pd.active = false;
    KZG_HIP_CHECK(ctx, hipStreamWaitEvent(pd.st, pd.accum_ev, 0));
    TailLayout tl;
    memcpy(&tl, pd.tail_off, sizeof tl);
    return msm_tail_run(ctx, pd.st, pd.mm, pd.part, pd.scratch, pd.s1, pd.B, pd.expected, (MsmState *)pd.state, pd.tail_base, tl, d_result, pd.odd);
}

int msm_run(kzg_ctx *ctx, int lane, const kzg_srs *srs, size_t offset, const void *d_scalars, size_t n, int sfmt,
            MsmPoint **d_result, hipStream_t accum_stream, hipEvent_t sorted_ev, hipEvent_t accum_ev, MsmPending *defer) {
    if (defer) defer->active = false;
    if (srs->device != ctx->device)  // (every MSM of every entry point passes here; a kernel on this GPU cannot read another GPU's table)
        return fail(ctx, KZG_ERR_SHAPE, "the SRS is resident on GPU " + std::to_string(srs->device) + ", this context runs on GPU " +
                                            std::to_string(ctx->device) + ": upload or generate it through this context");
    if (n > srs->n || offset > srs->n - n) return fail(ctx, KZG_ERR_SHAPE, "MSM range exceeds the SRS (reference: slice index panic)");
    if (srs->c > 16 && !srs->narrow17 && !(srs->sort20 && !ctx->opt_sort_single))
        return msm_run_wide(ctx, lane, srs, offset, d_scalars, n, sfmt, d_result);
    if ((uint64_t)srs->rows * srs->npad >= (1ull << 31))
        return fail(ctx, KZG_ERR_SHAPE, "SRS too large for the 31-bit entry index (table rows * points < 2^31)");
    if (srs->naf && offset + n > srs->npad) return fail(ctx, KZG_ERR_SHAPE, "MSM range exceeds the SRS");
    return msm_run_narrow(ctx, lane, srs, offset, d_scalars, n, sfmt, d_result, accum_stream, sorted_ev, accum_ev, defer);
}

}  // namespace kzg
