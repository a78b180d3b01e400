// msm_wide.hip -- kernels of the optional wide-window MSM path (17 < c <= 20): the second sort pass and the multi-block
// scans over the 2^(c-1) buckets (the tail is shared with the narrow path: msm_tail.hip).  Orchestrated from msm.hip (msm_run_wide); see
// msm_internal.h for why they live in their own translation unit.  DESIGN.md section 8 has the measurements.
#include "msm_internal.h"
#include "msm_sort.h"

namespace kzg {

// ---------------------------------------------------------------------------------------------
// wide windows (17 < c <= 20): second sort pass, multi-block scans
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ void hi_thread_range(uint32_t M, uint32_t &p0, uint32_t &p1) {
    const uint32_t chunk = (M + HI_BLOCKS - 1) / HI_BLOCKS;              // items per block
    const uint32_t run = (chunk + HI_THREADS - 1) / HI_THREADS;          // consecutive items per thread
    uint64_t b0 = (uint64_t)blockIdx.x * chunk;
    uint64_t b1 = b0 + chunk < M ? b0 + chunk : M;
    uint64_t q0 = b0 + (uint64_t)threadIdx.x * run;
    uint64_t q1 = q0 + run < b1 ? q0 + run : b1;
    if (q0 > b1) q0 = b1;
    p0 = (uint32_t)q0;
    p1 = (uint32_t)(q1 < q0 ? q0 : q1);
}

// per (block, hi) item counts of the lo-sorted list; blockcnt layout [hi][block]
__global__ __launch_bounds__(HI_THREADS) void k_hi_count(const uint32_t *entries, const MsmState *st, int nhi, uint32_t *blockcnt) {
    __shared__ uint32_t cnt[16];
    if (threadIdx.x < 16) cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t p0, p1;
    hi_thread_range(st->M, p0, p1);
    uint32_t mine[16];
#pragma unroll
    for (int h = 0; h < 16; h++) mine[h] = 0;
    for (uint32_t p = p0; p < p1; p++) {
        uint32_t h = (entries[p] & WIDE_HI_MASK) >> WIDE_HI_SHIFT;
#pragma unroll
        for (int k = 0; k < 16; k++) mine[k] += (h == (uint32_t)k) ? 1u : 0u;
    }
#pragma unroll
    for (int h = 0; h < 16; h++)
        if (mine[h]) atomicAdd(&cnt[h], mine[h]);
    __syncthreads();
    if (threadIdx.x < (unsigned)nhi) blockcnt[threadIdx.x * HI_BLOCKS + blockIdx.x] = cnt[threadIdx.x];
}

// exclusive scan of blockcnt in hi-major order (one block); binbase[hi][block]
__global__ __launch_bounds__(1024) void k_hi_scan(const uint32_t *blockcnt, int nhi, uint32_t *binbase) {
    __shared__ uint32_t lds[1024];
    // nhi * HI_BLOCKS <= 4096 values: 4 per thread
    const int N = nhi * HI_BLOCKS;
    uint32_t v[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int i = threadIdx.x * 4 + k;
        v[k] = i < N ? blockcnt[i] : 0u;
        sum += v[k];
    }
    lds[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t t = threadIdx.x >= (unsigned)off ? lds[threadIdx.x - off] : 0u;
        __syncthreads();
        lds[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = lds[threadIdx.x] - sum;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int i = threadIdx.x * 4 + k;
        if (i < N) binbase[i] = run;
        run += v[k];
    }
}

// Stable partition by hi of the lo-sorted list.  Because the order inside a hi bin stays lo-sorted, the running
// per-bin offsets at the position where lo bucket b begins ARE the starts of the 2^nhi_bits full buckets (hi, b):
// the kernel writes the final entry list and bucket_start[] of all nhi * 2^15 buckets without a histogram.
__global__ __launch_bounds__(HI_THREADS) void k_hi_scatter(const uint32_t *entries, const MsmState *st, int nhi, const uint32_t *binbase,
                                                           const uint32_t *lo_start, int B_lo, uint32_t *out,
                                                           uint32_t *bucket_start) {
    extern __shared__ uint32_t off[];  // [16][HI_THREADS]: per-thread running offsets of every bin
    const uint32_t M = st->M;
    uint32_t p0, p1;
    hi_thread_range(M, p0, p1);
    {
        uint32_t mine[16];
#pragma unroll
        for (int h = 0; h < 16; h++) mine[h] = 0;
        for (uint32_t p = p0; p < p1; p++) {
            uint32_t h = (entries[p] & WIDE_HI_MASK) >> WIDE_HI_SHIFT;
#pragma unroll
            for (int k = 0; k < 16; k++) mine[k] += (h == (uint32_t)k) ? 1u : 0u;
        }
#pragma unroll
        for (int h = 0; h < 16; h++) off[h * HI_THREADS + threadIdx.x] = mine[h];
    }
    __syncthreads();
    // wave w scans bin w over the 1024 threads (16 rounds of 64 lanes)
    {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        if (w < nhi) {
            uint32_t carry = binbase[w * HI_BLOCKS + blockIdx.x];
            for (int r = 0; r < HI_THREADS / 64; r++) {
                uint32_t x = off[w * HI_THREADS + r * 64 + lane], incl = x;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    uint32_t t = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += t;
                }
                off[w * HI_THREADS + r * 64 + lane] = incl - x + carry;
                carry += __shfl(incl, 63, 64);
            }
        }
    }
    __syncthreads();
    // M == 0: bucket_start was zero-filled by the host.  Threads without items have nothing to place; the boundaries
    // that coincide with the end of the list are emitted by the thread that owns the last item.
    if (M == 0 || p0 >= p1) return;
    // first lo bucket whose start is >= p0
    uint32_t bl = 0, bh = (uint32_t)B_lo;  // invariant: lo_start[bl] < p0 (or bl == 0), lo_start[bh] >= p0
    if (lo_start[0] >= p0) {
        bh = 0;
    } else {
        while (bh - bl > 1) {
            uint32_t mid = (bl + bh) >> 1;
            if (lo_start[mid] < p0) bl = mid; else bh = mid;
        }
    }
    uint32_t b = bh;
    for (uint32_t p = p0; p < p1; p++) {
        while (b < (uint32_t)B_lo && lo_start[b] == p) {
            for (int h = 0; h < nhi; h++) bucket_start[(size_t)h * B_lo + b] = off[h * HI_THREADS + threadIdx.x];
            b++;
        }
        uint32_t e = entries[p];
        uint32_t h = (e & WIDE_HI_MASK) >> WIDE_HI_SHIFT;
        uint32_t dst = off[h * HI_THREADS + threadIdx.x]++;
        out[dst] = e & ~WIDE_HI_MASK;
    }
    if (p1 == M) {  // owner of the last item: lo buckets that begin at M are empty in every bin
        while (b < (uint32_t)B_lo) {
            if (lo_start[b] == M)
                for (int h = 0; h < nhi; h++) bucket_start[(size_t)h * B_lo + b] = off[h * HI_THREADS + threadIdx.x];
            b++;
        }
        bucket_start[(size_t)nhi * B_lo] = M;
    }
}

// ---- multi-block exclusive scan over B per-bucket values (B a multiple of SEG or smaller than it) ----

struct ScanS1 {  // flags of the equal-split layout: non-empty bucket whose start is not a multiple of E
    const uint32_t *start;
    const MsmState *st;
    __device__ uint32_t operator()(int b) const {
        uint32_t s0 = start[b], s1 = start[b + 1];
        return (s1 != s0 && (s0 % st->E) != 0) ? 1u : 0u;
    }
    __device__ uint32_t cnt(int) const { return 0; }
};
template <class F>
__global__ __launch_bounds__(256) void k_seg_sums(F f, int B, const MsmState *st, int check_done, uint32_t *sums, uint32_t *maxs) {
    __shared__ uint32_t ssum, smax;
    if (threadIdx.x == 0) {
        ssum = 0;
        smax = 0;
    }
    __syncthreads();
    uint32_t a = 0, m = 0;
    int base = blockIdx.x * SEG;
    for (int k = 0; k < SEG / 256; k++) {
        int b = base + k * 256 + threadIdx.x;
        if (b < B) {
            a += f(b);
            uint32_t c = f.cnt(b);
            m = c > m ? c : m;
        }
    }
    atomicAdd(&ssum, a);
    atomicMax(&smax, m);
    __syncthreads();
    if (threadIdx.x == 0) {
        sums[blockIdx.x] = ssum;
        maxs[blockIdx.x] = smax;
    }
}

// one block: exclusive scan of the segment sums; mode 0 = S1 pass (no state change), mode 1 = fold level bookkeeping
__global__ __launch_bounds__(256) void k_seg_top(uint32_t *sums, const uint32_t *maxs, int nseg, int mode, MsmState *st, uint32_t level,
                                                 uint32_t in_buf, uint32_t *out_total) {
    __shared__ uint32_t lds[256];
    __shared__ uint32_t smax;
    uint32_t v = threadIdx.x < (unsigned)nseg ? sums[threadIdx.x] : 0u;
    uint32_t mx = threadIdx.x < (unsigned)nseg ? maxs[threadIdx.x] : 0u;
    if (threadIdx.x == 0) smax = 0;
    lds[threadIdx.x] = v;
    __syncthreads();
    atomicMax(&smax, mx);
    for (int off = 1; off < 256; off <<= 1) {
        uint32_t t = threadIdx.x >= (unsigned)off ? lds[threadIdx.x - off] : 0u;
        __syncthreads();
        lds[threadIdx.x] += t;
        __syncthreads();
    }
    if (threadIdx.x < (unsigned)nseg) sums[threadIdx.x] = lds[threadIdx.x] - v;
    if (threadIdx.x == 0) {
        uint32_t total = lds[255];
        *out_total = total;
    }
}

// out[b] = exclusive prefix (+ add(b) for the S1 layout), out[B] = total
template <class F>
__global__ __launch_bounds__(256) void k_seg_apply(F f, int B, const MsmState *st, int mode, const uint32_t *sums, const uint32_t *total,
                                                   uint32_t *out) {
    __shared__ uint32_t lds[256];
    int base = blockIdx.x * SEG + threadIdx.x * (SEG / 256);
    uint32_t v[SEG / 256], sum = 0;
#pragma unroll
    for (int k = 0; k < SEG / 256; k++) {
        int b = base + k;
        v[k] = b < B ? f(b) : 0u;
        sum += v[k];
    }
    lds[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        uint32_t t = threadIdx.x >= (unsigned)off ? lds[threadIdx.x - off] : 0u;
        __syncthreads();
        lds[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = lds[threadIdx.x] - sum + sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SEG / 256; k++) {
        int b = base + k;
        if (b < B) out[b] = run;
        run += v[k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[B] = *total;
}

// s1[b] += ceil(start[b] / E)   (b <= B), completing the equal-split layout of round 1
__global__ __launch_bounds__(256) void k_s1_finish(const uint32_t *start, int B, const MsmState *st, uint32_t *s1) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > B) return;
    const uint32_t E = st->E;
    s1[b] += (start[b] + E - 1) / E;
}

// ---------------------------------------------------------------------------------------------
// c = 20: two-level sort (13 windows of 20 bits cover 260 >= 255 bits, so the plain signed recoding never carries out of the top
// window: no balanced scalars).  Record = (entry word, low 9 bucket bits | bin << 16): the bin rides along so that level 1 can
// stage finished records in LDS.
// ---------------------------------------------------------------------------------------------
extern __shared__ __attribute__((aligned(16))) uint32_t lds_u32[];

__global__ __launch_bounds__(1024) void k_bin_hist20(const Fr *scalars, size_t n, int sfmt, size_t per_block, uint32_t *blk_bins) {
    __shared__ uint32_t h[NBINS];
    for (int b = threadIdx.x; b < NBINS; b += blockDim.x) h[b] = 0;
    __syncthreads();
    size_t i0 = (size_t)blockIdx.x * per_block;
    size_t i1 = i0 + per_block < n ? i0 + per_block : n;
    for (size_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        uint32_t s[8];
        load_scalar(scalars, i, sfmt, s);
        for_each_digit_fixed<SORT20_C, SORT20_W>(s, false, [&](int, uint32_t mag, uint32_t) { atomicAdd(&h[(mag - 1) >> SORT20_SHIFT], 1u); });
    }
    __syncthreads();
    for (int b = threadIdx.x; b < NBINS; b += blockDim.x) blk_bins[(size_t)blockIdx.x * NBINS + b] = h[b];
}

// Level 1: as k_bin_scatter of msm.hip (chunks of 1024 scalars, the chunk's <= 13 K records sorted by bin in LDS, then written
// out as runs), with the finished 8-byte records in the stage.
constexpr int BIN_SCATTER20_LDS = (3 * NBINS + 16) * 4 + SORT20_W * 1024 * 8;
__global__ __launch_bounds__(1024) void k_bin_scatter20(const Fr *scalars, size_t n, int sfmt, size_t per_block, const uint32_t *blk_off,
                                                        const uint32_t *bin_total, uint32_t *bin_base, uint32_t row_stride,
                                                        uint32_t idx_base, uint2 *rec) {
    uint32_t *cur = lds_u32, *cnt = cur + NBINS, *off = cnt + NBINS, *wsum = off + NBINS;
    uint2 *stage = reinterpret_cast<uint2 *>(wsum + 16);
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
        uint2 pk[SORT20_W];
        uint32_t rk[SORT20_W], have = 0;
        if (c0 + tid < i1) {
            uint32_t s[8];
            load_scalar(scalars, c0 + tid, sfmt, s);
            const uint32_t e0 = idx_base + (uint32_t)c0 + tid;
            for_each_digit_fixed<SORT20_C, SORT20_W>(s, false, [&](int w, uint32_t mag, uint32_t neg) {
                const uint32_t idx = mag - 1, bin = idx >> SORT20_SHIFT;
                rk[w] = atomicAdd(&cnt[bin], 1u);
                pk[w] = make_uint2(((uint32_t)w * row_stride + e0) | (neg << 31), (idx & (SORT20_BUCKETS - 1)) | (bin << 16));
                have |= 1u << w;
            });
        }
        __syncthreads();
        uint32_t tot;
        off[tid] = block_scan_1024(cnt[tid], wsum, &tot);
        __syncthreads();
#pragma unroll
        for (int w = 0; w < SORT20_W; w++)
            if ((have >> w) & 1u) stage[off[pk[w].y >> 16] + rk[w]] = pk[w];
        __syncthreads();
        for (uint32_t p = tid; p < tot; p += 1024) {
            const uint2 v = stage[p];
            const uint32_t bin = v.y >> 16;
            rec[cur[bin] + p - off[bin]] = v;
        }
        __syncthreads();
        cur[tid] += cnt[tid];
    }
}

// Level 2 of both two-level sorts, and HEAVY bins.
//
// Level 2 sorts a bin with one block, which assumes bins of similar size.  They are not when many digits are small: scalars that
// are bits or bytes put every entry into a handful of buckets of bin 0, u64-valued scalars at c = 20 put the whole top window
// (4 bits + carry) there, all-equal scalars put everything into 13 / 15 buckets -- one block then places up to M records, through
// LDS atomics on a few addresses (measured at 2^20, c = 17: k_bin_sort 0.04 ms for uniform scalars, 1.3 ms for bits, 2.6 ms for
// all-ones).  A bin of more than 4 SL records (SL = max(32768, M / 2048)) is therefore cut into slices of SL records that
// separate blocks sort: k_heavy_count (every block derives the plan from the bin starts, block 0 publishes it as hv[]; per-slice bucket counts),
// k_heavy_scan (per bucket the exclusive scan over the bin's slices; bucket sizes and starts), k_heavy_place.
// hv[0] = slices in total (0: count / scan / place return at once), hv[1] = SL, hv[2 + bin] = first slice of the bin,
// hv[2 + NBINS + bin] = its slice count (0 = not heavy: the one-block kernel takes it).  At most M / SL + M / (4 SL) <= 2560 slices.
constexpr int SORT2_THREADS = 256, SORT2_UNROLL = 8, SORT2_CHUNK = SORT2_THREADS * SORT2_UNROLL;
#ifndef KZG_HEAVY_GRID
#define KZG_HEAVY_GRID 256
#endif
constexpr int HEAVY_MAX_SLICES = 2560, HEAVY_GRID = KZG_HEAVY_GRID, HEAVY_SCAN_GRID = 64;
static_assert(SORT2_HV_WORDS >= 2 + 2 * NBINS, "hv layout");

// exclusive scan of h[0 .. NB) into off[] (NB = 64: one value per thread of the first wave; 512: two per thread); synchronised
// on entry by the caller, NOT on return
template <int NB>
__device__ __forceinline__ void sort2_scan(const uint32_t *h, uint32_t *off, uint32_t *sc) {
    static_assert(NB == 64 || NB == 512, "buckets per bin");
    const uint32_t tid = threadIdx.x;
    if (NB == 512) {
        const uint32_t v0 = h[2 * tid], v1 = h[2 * tid + 1];
        uint32_t tot;
        const uint32_t ex = block_scan_256(v0 + v1, sc, &tot);
        off[2 * tid] = ex;
        off[2 * tid + 1] = ex + v0;
    } else if (tid < NB) {  // NB = 64: the first wave alone, no barrier
        const uint32_t v = h[tid];
        uint32_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = __shfl_up(incl, o, 64);
            if ((int)tid >= o) incl += u;
        }
        off[tid] = incl - v;
    }
}

// bucket counts of records [a, b) into h[] (zeroed here); synchronised on return
template <class REC, int NB>
__device__ __forceinline__ void sort2_count_range(const typename REC::T *rec, uint32_t a, uint32_t b, uint32_t *h) {
    typedef typename REC::T RT;
    const uint32_t tid = threadIdx.x;
    for (uint32_t k = tid; k < NB; k += SORT2_THREADS) h[k] = 0;
    __syncthreads();
    for (uint32_t base = a + tid; base < b; base += SORT2_CHUNK) {
        RT e[SORT2_UNROLL];
#pragma unroll
        for (int k = 0; k < SORT2_UNROLL; k++) {
            const uint32_t r = base + (uint32_t)k * SORT2_THREADS;
            e[k] = r < b ? rec[r] : REC::invalid();
        }
#pragma unroll
        for (int k = 0; k < SORT2_UNROLL; k++)
            if (REC::valid(e[k])) atomicAdd(&h[REC::lo(e[k])], 1u);
    }
    __syncthreads();
}

// records [c_begin, c_end) -> entries[], by bucket, each chunk sorted in LDS first; cur[] = next free position per bucket (set by
// the caller; the first barrier inside orders it)
template <class REC, int NB>
__device__ __forceinline__ void sort2_place_range(const typename REC::T *rec, uint32_t c_begin, uint32_t c_end, uint32_t *entries,
                                                  uint32_t *h, uint32_t *cur, uint32_t *off, uint32_t *sc, typename REC::T *stage) {
    typedef typename REC::T RT;
    const uint32_t tid = threadIdx.x;
    for (uint32_t c0 = c_begin; c0 < c_end; c0 += SORT2_CHUNK) {
        for (uint32_t k = tid; k < NB; k += SORT2_THREADS) h[k] = 0;
        __syncthreads();
        RT e[SORT2_UNROLL];
        uint32_t rk[SORT2_UNROLL];
#pragma unroll
        for (int k = 0; k < SORT2_UNROLL; k++) {
            const uint32_t r = c0 + tid + (uint32_t)k * SORT2_THREADS;
            e[k] = r < c_end ? rec[r] : REC::invalid();
        }
#pragma unroll
        for (int k = 0; k < SORT2_UNROLL; k++)
            if (REC::valid(e[k])) rk[k] = atomicAdd(&h[REC::lo(e[k])], 1u);
        __syncthreads();
        sort2_scan<NB>(h, off, sc);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SORT2_UNROLL; k++)
            if (REC::valid(e[k])) stage[off[REC::lo(e[k])] + rk[k]] = e[k];
        __syncthreads();
        const uint32_t m = c_end - c0 < (uint32_t)SORT2_CHUNK ? c_end - c0 : (uint32_t)SORT2_CHUNK;
        for (uint32_t p = tid; p < m; p += SORT2_THREADS) {
            const RT v = stage[p];
            const uint32_t b = REC::lo(v);
            entries[cur[b] + p - off[b]] = REC::entry(v);
        }
        __syncthreads();
        for (uint32_t k = tid; k < NB; k += SORT2_THREADS) cur[k] += h[k];
    }
}

// One block per bin (not heavy): bucket sizes -> total[] (if asked for) and bucket starts -> bucket_start[] (if asked for), then the
// records -> entries[].  set_state: block 0 also sets the equal-split state (the c = 20 pipeline; c = 17 does it in k_scan_b_bins).
// slice length and "this bin is sorted in slices" from the sizes alone (every block of every kernel derives the same plan)
__device__ __forceinline__ uint32_t heavy_slice_len(uint32_t M) {
    const uint32_t SL = (M + 2047) / 2048;
    return SL < 32768u ? 32768u : SL;
}
__device__ __forceinline__ uint32_t heavy_slices(uint32_t size, uint32_t SL) { return size > 4 * SL ? (size + SL - 1) / SL : 0u; }

template <class REC, int NB>
__global__ __launch_bounds__(SORT2_THREADS) void k_bin_sort2(const typename REC::T *rec, const uint32_t *bin_base, uint32_t *entries,
                                                             uint32_t *total, uint32_t *bucket_start, MsmState *st, uint32_t slots,
                                                             int sliced, uint32_t *lane_heavy, uint32_t heavy_seq) {
    typedef typename REC::T RT;
    __shared__ uint32_t h[NB], cur[NB], off[NB], sc[4];
    __shared__ RT stage[SORT2_CHUNK];
    const uint32_t tid = threadIdx.x;
    const uint32_t r0 = bin_base[blockIdx.x], r1 = bin_base[blockIdx.x + 1];
    if (st && blockIdx.x == 0 && tid == 0) {
        const uint32_t M = bin_base[NBINS];
        uint32_t E = (M + slots - 1) / slots;
        if (E < KZG_ACCUM_MIN_CHUNK) E = KZG_ACCUM_MIN_CHUNK;
        st->M = M;
        st->E = E;
        st->ntasks = (M + E - 1) / E;
        st->ovf_tasks = 0;
        bucket_start[(size_t)NBINS * NB] = M;
    }
    if (heavy_slices(r1 - r0, heavy_slice_len(bin_base[NBINS]))) {  // an oversized bin
        if (tid == 0) *lane_heavy = heavy_seq;  // what the host decides the NEXT calls on
        if (sliced) return;                     // the slice kernels sort it
    }
    sort2_count_range<REC, NB>(rec, r0, r1, h);
    sort2_scan<NB>(h, off, sc);
    __syncthreads();
    for (uint32_t k = tid; k < NB; k += SORT2_THREADS) {
        const uint32_t start = r0 + off[k];
        cur[k] = start;
        if (total) total[(size_t)blockIdx.x * NB + k] = h[k];
        if (bucket_start) bucket_start[(size_t)blockIdx.x * NB + k] = start;
    }
    sort2_place_range<REC, NB>(rec, r0, r1, entries, h, cur, off, sc, stage);
}

// the heavy bin that owns slice `id` (< hv[0]): the last bin whose first slice is <= id (bins that are not heavy share their
// first-slice number with the next heavy one)
__device__ __forceinline__ uint32_t heavy_bin_of(const uint32_t *hv, uint32_t id) {
    uint32_t lo = 0, hi = NBINS;  // slice0[lo] <= id, slice0[hi] > id (hi = NBINS: sentinel)
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (hv[2 + mid] <= id) lo = mid; else hi = mid;
    }
    return lo;
}

template <class REC, int NB>
__global__ __launch_bounds__(SORT2_THREADS) void k_heavy_count(const typename REC::T *rec, const uint32_t *bin_base, uint32_t *hv,
                                                               uint32_t *hcnt) {
    __shared__ uint32_t h[NB], plan[2 + 2 * NBINS], sc[4];
    const uint32_t tid = threadIdx.x;
    {   // the plan: slices per bin, their exclusive scan (four bins per thread)
        const uint32_t SL = heavy_slice_len(bin_base[NBINS]);
        uint32_t ns[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t bin = 4 * tid + k;
            ns[k] = heavy_slices(bin_base[bin + 1] - bin_base[bin], SL);
            sum += ns[k];
        }
        uint32_t tot;
        uint32_t run = block_scan_256(sum, sc, &tot);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            plan[2 + 4 * tid + k] = run;
            plan[2 + NBINS + 4 * tid + k] = ns[k];
            run += ns[k];
        }
        if (tid == 0) {
            plan[0] = tot;
            plan[1] = SL;
        }
        __syncthreads();
        if (blockIdx.x == 0)
            for (uint32_t k = tid; k < 2 + 2 * NBINS; k += SORT2_THREADS) hv[k] = plan[k];
    }
    const uint32_t total = plan[0], SL = plan[1];
    for (uint32_t id = blockIdx.x; id < total; id += gridDim.x) {
        const uint32_t bin = heavy_bin_of(plan, id), j = id - plan[2 + bin];
        const uint32_t r0 = bin_base[bin], r1 = bin_base[bin + 1];
        const uint32_t a = r0 + j * SL, b = a + SL < r1 ? a + SL : r1;
        sort2_count_range<REC, NB>(rec, a, b, h);
        for (uint32_t k = tid; k < NB; k += SORT2_THREADS) hcnt[(size_t)id * NB + k] = h[k];
        __syncthreads();
    }
}

// one block per heavy bin, one thread per bucket: counts -> exclusive offsets over the bin's slices; bucket sizes and starts
template <int NB>
__global__ __launch_bounds__(NB) void k_heavy_scan(const uint32_t *bin_base, const uint32_t *hv, uint32_t *hcnt, uint32_t *total,
                                                   uint32_t *bucket_start) {
    __shared__ uint32_t ws[8];
    if (hv[0] == 0) return;
    for (uint32_t bin = blockIdx.x; bin < NBINS; bin += gridDim.x) {
    const uint32_t ns = hv[2 + NBINS + bin];
    if (ns == 0) continue;  // (block-uniform)
    const uint32_t s0 = hv[2 + bin], b = threadIdx.x;
    uint32_t run = 0, j = 0;
    for (; j + 8 <= ns; j += 8) {  // eight independent loads in flight
        uint32_t t[8];
#pragma unroll
        for (int k = 0; k < 8; k++) t[k] = hcnt[(size_t)(s0 + j + k) * NB + b];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            hcnt[(size_t)(s0 + j + k) * NB + b] = run;
            run += t[k];
        }
    }
    for (; j < ns; j++) {
        const uint32_t t = hcnt[(size_t)(s0 + j) * NB + b];
        hcnt[(size_t)(s0 + j) * NB + b] = run;
        run += t;
    }
    // exclusive scan of the NB bucket sizes
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t u = __shfl_up(incl, o, 64);
        if (lane >= o) incl += u;
    }
    if (lane == 63) ws[wave] = incl;
    __syncthreads();
    uint32_t woff = 0;
#pragma unroll
    for (int w = 0; w < NB / 64; w++)
        if (w < wave) woff += ws[w];
    if (total) total[(size_t)bin * NB + b] = run;
    bucket_start[(size_t)bin * NB + b] = bin_base[bin] + woff + incl - run;
    __syncthreads();
    }
}

template <class REC, int NB>
__global__ __launch_bounds__(SORT2_THREADS) void k_heavy_place(const typename REC::T *rec, const uint32_t *bin_base, const uint32_t *hv,
                                                               const uint32_t *hcnt, const uint32_t *bucket_start, uint32_t *entries) {
    typedef typename REC::T RT;
    __shared__ uint32_t h[NB], cur[NB], off[NB], sc[4];
    __shared__ RT stage[SORT2_CHUNK];
    const uint32_t total = hv[0], SL = hv[1], tid = threadIdx.x;
    for (uint32_t id = blockIdx.x; id < total; id += gridDim.x) {
        const uint32_t bin = heavy_bin_of(hv, id), j = id - hv[2 + bin];
        const uint32_t r0 = bin_base[bin], r1 = bin_base[bin + 1];
        const uint32_t a = r0 + j * SL, b = a + SL < r1 ? a + SL : r1;
        for (uint32_t k = tid; k < NB; k += SORT2_THREADS) cur[k] = bucket_start[(size_t)bin * NB + k] + hcnt[(size_t)id * NB + k];
        sort2_place_range<REC, NB>(rec, a, b, entries, h, cur, off, sc, stage);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// launchers used by msm.hip's wide orchestration
// ---------------------------------------------------------------------------------------------
int wide_sort_pass2(kzg_ctx *ctx, hipStream_t st, const uint32_t *entries1, const MsmState *state, int nhi, uint32_t *blockcnt,
                    uint32_t *binbase, const uint32_t *lo_start, int B_lo, uint32_t *entries2, uint32_t *bucket_start) {
    if (!ctx->attr_wide_set) {
        KZG_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)k_hi_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, 16 * HI_THREADS * 4));
        ctx->attr_wide_set = true;
    }
    KZG_HIP_CHECK(ctx, hipMemsetAsync(bucket_start, 0, ((size_t)nhi * B_lo + 1) * 4, st));
    KZG_LAUNCH(ctx, st, "k_hi_count", k_hi_count, HI_BLOCKS, HI_THREADS, 0, entries1, state, nhi, blockcnt);
    KZG_LAUNCH(ctx, st, "k_hi_scan", k_hi_scan, 1, 1024, 0, blockcnt, nhi, binbase);
    KZG_LAUNCH(ctx, st, "k_hi_scatter", k_hi_scatter, HI_BLOCKS, HI_THREADS, 16 * HI_THREADS * 4, entries1, state, nhi, binbase,
               lo_start, B_lo, entries2, bucket_start);
    return KZG_OK;
}

int sort20_hist(kzg_ctx *ctx, hipStream_t st, const void *d_scalars, size_t n, int sfmt, int G2, uint32_t *bins) {
    const size_t per2 = (n + G2 - 1) / G2;
    KZG_LAUNCH(ctx, st, "k_bin_hist", k_bin_hist20, G2, 1024, 0, (const Fr *)d_scalars, n, sfmt, per2, bins);
    return KZG_OK;
}

// level 2 + heavy bins.  kind: 4 = Rec4, 8 = Rec8 (c = 17: NB = 64, sizes -> total[]), 20 = Rec20 (NB = 512, starts and state)
template <class REC, int NB>
static int sort2_level2_t(kzg_ctx *ctx, hipStream_t st, const void *rec_v, const uint32_t *bin_base, uint32_t *hv, uint32_t *entries,
                          uint32_t *total, uint32_t *bucket_start, MsmState *state, uint32_t slots, bool sliced, uint32_t *lane_heavy,
                          uint32_t heavy_seq) {
    typedef typename REC::T RT;
    const RT *rec = (const RT *)rec_v;
    uint32_t *hcnt = hv + SORT2_HV_WORDS;
    KZG_LAUNCH(ctx, st, "k_bin_sort", (k_bin_sort2<REC, NB>), NBINS, SORT2_THREADS, 0, rec, bin_base, entries, total,
               state ? bucket_start : nullptr, state, slots, sliced ? 1 : 0, lane_heavy, heavy_seq);
    if (!sliced) return KZG_OK;
    KZG_LAUNCH(ctx, st, "k_heavy_count", (k_heavy_count<REC, NB>), HEAVY_GRID, SORT2_THREADS, 0, rec, bin_base, hv, hcnt);
    KZG_LAUNCH(ctx, st, "k_heavy_scan", k_heavy_scan<NB>, HEAVY_SCAN_GRID, NB, 0, bin_base, hv, hcnt, total, bucket_start);
    KZG_LAUNCH(ctx, st, "k_heavy_place", (k_heavy_place<REC, NB>), HEAVY_GRID, SORT2_THREADS, 0, rec, bin_base, hv, hcnt, bucket_start,
               entries);
    return KZG_OK;
}

int sort2_level2(kzg_ctx *ctx, hipStream_t st, int kind, const void *rec, const uint32_t *bin_base, const uint32_t *bin_total, uint32_t *hv,
                 uint32_t *entries, uint32_t *total, uint32_t *bucket_start, MsmState *state, uint32_t slots, bool sliced, uint32_t *lane_heavy,
                 uint32_t heavy_seq) {
    (void)bin_total;
    if (kind == 4) return sort2_level2_t<Rec4, 64>(ctx, st, rec, bin_base, hv, entries, total, bucket_start, nullptr, slots, sliced, lane_heavy, heavy_seq);
    if (kind == 8) return sort2_level2_t<Rec8, 64>(ctx, st, rec, bin_base, hv, entries, total, bucket_start, nullptr, slots, sliced, lane_heavy, heavy_seq);
    return sort2_level2_t<Rec20, SORT20_BUCKETS>(ctx, st, rec, bin_base, hv, entries, nullptr, bucket_start, state, slots, sliced, lane_heavy, heavy_seq);
}

int sort20_place(kzg_ctx *ctx, hipStream_t st, const void *d_scalars, size_t n, int sfmt, int G2, const uint32_t *bins,
                 const uint32_t *bin_total, uint32_t *bin_base, uint32_t row_stride, uint32_t idx_base, void *rec, uint32_t *entries,
                 uint32_t *bucket_start, uint32_t *s1, MsmState *state, uint32_t slots, uint32_t *segsums, uint32_t *segmaxs,
                 uint32_t *segtotal, uint32_t *hv, bool sliced, uint32_t *lane_heavy, uint32_t heavy_seq) {
    if (!ctx->attr_sort20_set) {
        KZG_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)k_bin_scatter20, hipFuncAttributeMaxDynamicSharedMemorySize, BIN_SCATTER20_LDS));
        ctx->attr_sort20_set = true;
    }
    const size_t per2 = (n + G2 - 1) / G2;
    KZG_LAUNCH(ctx, st, "k_bin_scatter", k_bin_scatter20, G2, 1024, BIN_SCATTER20_LDS, (const Fr *)d_scalars, n, sfmt, per2, bins, bin_total,
               bin_base, row_stride, idx_base, (uint2 *)rec);
    KZG_TRY(sort2_level2(ctx, st, 20, rec, bin_base, bin_total, hv, entries, nullptr, bucket_start, state, slots, sliced, lane_heavy, heavy_seq));
    return wide_s1_layout(ctx, st, bucket_start, NBINS * SORT20_BUCKETS, state, segsums, segmaxs, segtotal, s1);
}

int wide_s1_layout(kzg_ctx *ctx, hipStream_t st, const uint32_t *bucket_start, int Btot, MsmState *state, uint32_t *segsums,
                   uint32_t *segmaxs, uint32_t *segtotal, uint32_t *s1_out) {
    const int nseg = (Btot + SEG - 1) / SEG;
    ScanS1 f{bucket_start, state};
    KZG_LAUNCH(ctx, st, "k_seg_sums", k_seg_sums<ScanS1>, nseg, 256, 0, f, Btot, state, 0, segsums, segmaxs);
    KZG_LAUNCH(ctx, st, "k_seg_top", k_seg_top, 1, 256, 0, segsums, segmaxs, nseg, 0, state, 0u, 0u, segtotal);
    KZG_LAUNCH(ctx, st, "k_seg_apply", k_seg_apply<ScanS1>, nseg, 256, 0, f, Btot, state, 0, segsums, segtotal, s1_out);
    KZG_LAUNCH(ctx, st, "k_s1_finish", k_s1_finish, (Btot + 1 + 255) / 256, 256, 0, bucket_start, Btot, state, s1_out);
    return KZG_OK;
}

}  // namespace kzg
