// msm_wide.hip -- kernels of the optional wide-window MSM path (16 < c <= 20): the second sort pass, multi-block scans
// over the 2^(c-1) buckets and the row/column bucket reduction.  Orchestrated from msm.hip (msm_run_wide); see
// msm_internal.h for why they live in their own translation unit.  DESIGN.md section 8 has the measurements.
#include "msm_internal.h"

namespace kzg {

// ---------------------------------------------------------------------------------------------
// wide windows (16 < c <= 20): second sort pass, multi-block scans, row/column bucket reduction
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
struct ScanLevel {  // tasks of a fold level: ceil(partials / L)
    const uint32_t *in_start;
    int L;
    __device__ uint32_t operator()(int b) const { return (in_start[b + 1] - in_start[b] + L - 1) / L; }
    __device__ uint32_t cnt(int b) const { return in_start[b + 1] - in_start[b]; }
};

template <class F>
__global__ __launch_bounds__(256) void k_seg_sums(F f, int B, const MsmState *st, int check_done, uint32_t *sums, uint32_t *maxs) {
    if (check_done && st->done) return;
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
    if (mode == 1 && st->done) return;
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
        if (mode == 1) {
            if (smax <= 1) {
                st->done = 1;
                st->final_level = level;
                st->final_buf = in_buf;
                st->max_cnt = smax;
            } else {
                st->ntasks = total;
                st->max_cnt = smax;
            }
        }
    }
}

// out[b] = exclusive prefix (+ add(b) for the S1 layout), out[B] = total
template <class F>
__global__ __launch_bounds__(256) void k_seg_apply(F f, int B, const MsmState *st, int mode, const uint32_t *sums, const uint32_t *total,
                                                   uint32_t *out) {
    if (mode == 1 && st->done) return;
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

// ---- bucket reduction for B = R x C buckets:  sum (b+1) X_b = C * sum_r r Row_r + sum_c (c+1) Col_c ----
__device__ __forceinline__ MsmPoint final_bucket(const MsmPoint *buf, const uint32_t *start, uint32_t b) {
    uint32_t s = start[b];
    return start[b + 1] > s ? buf[s] : MsmPoint::infinity();
}

// rows: out[r * (C/8) + j] = sum of the 8 consecutive buckets r*C + 8j .. +7
__global__ __launch_bounds__(64) void k_rows8(const MsmPoint *buf0, const MsmPoint *buf1, const uint32_t *starts, int Btot, int C,
                                              const MsmState *st, MsmPoint *out) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint32_t)(Btot / 8)) return;
    const uint32_t *start = starts + (size_t)st->final_level * (Btot + 1);
    const MsmPoint *buf = st->final_buf ? buf1 : buf0;
    MsmPoint acc = MsmPoint::infinity();
    for (int k = 0; k < 8; k++) acc = g1_add30(acc, final_bucket(buf, start, t * 8 + k));
    out[t] = acc;
}

// columns: out[c * (R/8) + g] = sum over the 8 rows 8g .. 8g+7 of bucket (row, c)
__global__ __launch_bounds__(64) void k_cols8(const MsmPoint *buf0, const MsmPoint *buf1, const uint32_t *starts, int Btot, int C,
                                              const MsmState *st, MsmPoint *out) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const int R = Btot / C;
    if (t >= (uint32_t)(C * (R / 8))) return;
    const uint32_t *start = starts + (size_t)st->final_level * (Btot + 1);
    const MsmPoint *buf = st->final_buf ? buf1 : buf0;
    uint32_t c = t % C, g = t / C;  // consecutive threads -> consecutive columns of the same row group (coalesced starts)
    MsmPoint acc = MsmPoint::infinity();
    for (int k = 0; k < 8; k++) acc = g1_add30(acc, final_bucket(buf, start, (g * 8 + k) * C + c));
    out[(size_t)c * (R / 8) + g] = acc;
}

// out[t] = sum_{i in chunk t} (i + 1 + first_weight) * pts[i]   (chunks of CH points; N need not be a multiple)
__global__ __launch_bounds__(64) void k_weighted_chunks(const MsmPoint *pts, int N, int CH, int first_weight, MsmPoint *out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int nchunks = (N + CH - 1) / CH;
    if (t >= nchunks) return;
    int lo = t * CH, hi = lo + CH < N ? lo + CH : N;
    MsmPoint run = MsmPoint::infinity(), acc = MsmPoint::infinity();
    for (int i = hi - 1; i >= lo; i--) {
        run = g1_add30(run, pts[i]);
        acc = g1_add30(acc, run);
    }
    int base = lo + first_weight;  // acc = sum (i - lo + 1) pts[i]; add base * run
    if (base != 0 && !run.inf) {
        MsmPoint m = MsmPoint::infinity();
        for (int bit = 30; bit >= 0; bit--) {
            m = g1_dbl30(m);
            if ((base >> bit) & 1) m = g1_add30(m, run);
        }
        acc = g1_add30(acc, m);
    }
    out[t] = acc;
}

// result = 2^shift * a + b
__global__ __launch_bounds__(64) void k_combine_shifted(const MsmPoint *a, int shift, const MsmPoint *b, MsmPoint *result) {
    MsmPoint m = *a;
    for (int k = 0; k < shift; k++) m = g1_dbl30(m);
    *result = g1_add30(m, *b);
}


// ---------------------------------------------------------------------------------------------
// rare path of the c <= 16 pipeline: all fold levels after the first FAST_LEVELS, in one block
// ---------------------------------------------------------------------------------------------
// The fold levels beyond the first FAST_LEVELS, in one block.  Normal inputs: every bucket already holds one partial and the
// kernel returns after one pass over the counts.  Otherwise the buckets that still hold several partials are few (after the
// grid-wide round only buckets more than LK times the equal-split chunk long: the carry bucket of u64-valued scalars -- half of all
// scalars put a digit 1 into window 4 --, or the handful of buckets of adversarial inputs), so they are compacted into a list
// in LDS once and reduced by fan-in-L trees IN PLACE: level by level between the two partial buffers at the bucket's own
// offset, the single result ending in slot 0 of the bucket's range in the input list, which is where k_bucket_reduce reads it.
// No per-level pass over all 2^15 buckets (that cost 0.55 ms per level: 2.8 ms for the u64 case).  256 threads: the
// additions need ~200 VGPRs (a 1024-thread block is capped at 128 and spills).  More than FOLD_LIST multi-partial buckets:
// the generic level loop (scan of all buckets per level).
constexpr int FOLD_LIST = 1024;
__global__ __launch_bounds__(256) void k_fold_rest(MsmPoint *buf0, MsmPoint *buf1, uint32_t *starts, int B, int L, int level0,
                                                   int max_level, MsmState *st) {
    if (st->done) return;
    __shared__ uint32_t lds[1024];
    __shared__ uint32_t ls[FOLD_LIST], lc[FOLD_LIST], lpre[FOLD_LIST + 1];
    __shared__ uint32_t smax, nmulti, total;
    const uint32_t *start0 = starts + (size_t)level0 * (B + 1);
    if (threadIdx.x == 0) {
        smax = 0;
        nmulti = 0;
    }
    __syncthreads();
    {
        uint32_t mx = 0;
        for (int b = threadIdx.x; b < B; b += blockDim.x) {
            uint32_t s = start0[b], cnt = start0[b + 1] - s;
            mx = cnt > mx ? cnt : mx;
            if (cnt > 1) {
                uint32_t i = atomicAdd(&nmulti, 1u);
                if (i < (uint32_t)FOLD_LIST) {
                    ls[i] = s;
                    lc[i] = cnt;
                }
            }
        }
        atomicMax(&smax, mx);
    }
    __syncthreads();
    if (smax <= 1) {
        if (threadIdx.x == 0) {
            st->done = 1;
            st->final_level = (uint32_t)level0;
            st->final_buf = (uint32_t)(level0 & 1);
            st->max_cnt = smax;
        }
        return;
    }
    if (nmulti <= (uint32_t)FOLD_LIST && level0 >= 1) {
        // level0 >= 1: the other buffer held the (longer) list of the level before, so every offset used below fits it
        const uint32_t nm = nmulti;
        MsmPoint *X = (level0 & 1) ? buf1 : buf0, *Y = (level0 & 1) ? buf0 : buf1;
        MsmPoint *const X0 = X;
        for (;;) {
            if (threadIdx.x == 0) {
                uint32_t run = 0, mx = 0;
                for (uint32_t i = 0; i < nm; i++) {
                    lpre[i] = run;
                    run += (lc[i] + L - 1) / L;
                    mx = lc[i] > mx ? lc[i] : mx;
                }
                lpre[nm] = run;
                total = run;
                smax = mx;
            }
            __syncthreads();
            if (smax <= 1) break;
            const uint32_t T = total;
            for (uint32_t t = threadIdx.x; t < T; t += blockDim.x) {
                uint32_t lo = 0, hi = nm;  // lpre[lo] <= t < lpre[hi]
                while (hi - lo > 1) {
                    uint32_t mid = (lo + hi) >> 1;
                    if (lpre[mid] <= t) lo = mid; else hi = mid;
                }
                const uint32_t j = t - lpre[lo];
                uint32_t s = ls[lo] + j * L, e = ls[lo] + lc[lo];
                e = s + L < e ? s + L : e;
                MsmPoint acc = X[s];
                for (uint32_t k = s + 1; k < e; k++) acc = g1_add30(acc, X[k]);
                Y[ls[lo] + j] = acc;
            }
            __threadfence_block();
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < nm; i += blockDim.x) lc[i] = (lc[i] + L - 1) / L;
            MsmPoint *tmp = X;
            X = Y;
            Y = tmp;
            __syncthreads();
        }
        if (X != X0)  // odd number of levels: the results sit in the other buffer
            for (uint32_t i = threadIdx.x; i < nm; i += blockDim.x) X0[ls[i]] = X[ls[i]];
        if (threadIdx.x == 0) {
            st->done = 1;
            st->final_level = (uint32_t)level0;
            st->final_buf = (uint32_t)(level0 & 1);
            st->max_cnt = 1;
        }
        return;
    }
    // generic level loop
    int level = level0;
    for (;;) {
        const uint32_t *in_start = starts + (size_t)level * (B + 1);
        uint32_t *out_start = starts + (size_t)(level + 1) * (B + 1);
        if (threadIdx.x == 0) smax = 0;
        __syncthreads();
        uint32_t mx = 0;
        for (int b = threadIdx.x; b < B; b += blockDim.x) {
            uint32_t cnt = in_start[b + 1] - in_start[b];
            mx = cnt > mx ? cnt : mx;
        }
        atomicMax(&smax, mx);
        __syncthreads();
        if (smax <= 1 || level >= max_level) {
            if (threadIdx.x == 0) {
                st->done = smax <= 1 ? 1u : 0u;
                st->final_level = (uint32_t)level;
                st->final_buf = (uint32_t)(level & 1);
                st->max_cnt = smax;
            }
            return;
        }
        const uint32_t T = block_exclusive_scan(
            B, [&](int b) { return (in_start[b + 1] - in_start[b] + L - 1) / L; }, out_start, lds);
        const MsmPoint *in = (level & 1) ? buf1 : buf0;
        MsmPoint *out = (level & 1) ? buf0 : buf1;
        for (uint32_t t = threadIdx.x; t < T; t += blockDim.x) {
            uint32_t b, j;
            find_task(out_start, B, t, b, j);
            uint32_t s = in_start[b] + j * L;
            uint32_t e = in_start[b + 1];
            e = s + L < e ? s + L : e;
            MsmPoint acc = in[s];
            for (uint32_t k = s + 1; k < e; k++) acc = g1_add30(acc, in[k]);
            out[t] = acc;
        }
        __threadfence_block();
        __syncthreads();
        level++;
    }
}
int fold_rest_run(kzg_ctx *ctx, hipStream_t st, MsmPoint *buf0, MsmPoint *buf1, uint32_t *starts, int B, int L, int level0,
                  int max_level, MsmState *state) {
    KZG_LAUNCH(ctx, st, "k_fold_rest", k_fold_rest, 1, 256, 0, buf0, buf1, starts, B, L, level0, max_level, state);
    return KZG_OK;
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

int wide_level_scan(kzg_ctx *ctx, hipStream_t st, const uint32_t *in_start, uint32_t *out_start, int Btot, int L, MsmState *state,
                    uint32_t level, uint32_t in_buf, uint32_t *segsums, uint32_t *segmaxs, uint32_t *segtotal, bool apply) {
    const int nseg = (Btot + SEG - 1) / SEG;
    ScanLevel f{in_start, L};
    KZG_LAUNCH(ctx, st, "k_seg_sums", k_seg_sums<ScanLevel>, nseg, 256, 0, f, Btot, state, 1, segsums, segmaxs);
    KZG_LAUNCH(ctx, st, "k_seg_top", k_seg_top, 1, 256, 0, segsums, segmaxs, nseg, 1, state, level, in_buf, segtotal);
    if (apply) KZG_LAUNCH(ctx, st, "k_seg_apply", k_seg_apply<ScanLevel>, nseg, 256, 0, f, Btot, state, 1, segsums, segtotal, out_start);
    return KZG_OK;
}

// sum_level over groups that must not straddle `per` consecutive inputs; returns the array holding one point per group
static MsmPoint *reduce_groups(kzg_ctx *ctx, hipStream_t st, MsmPoint *in, size_t groups, size_t per, MsmPoint *bufs[2]) {
    int which = 0;
    while (per > 1) {
        int Lf = per >= 8 ? 8 : (int)per;
        size_t count = groups * per;
        sum_level_run(ctx, st, in, (uint32_t)count, Lf, bufs[which]);
        in = bufs[which];
        which ^= 1;
        per /= Lf;
    }
    return in;
}

int wide_bucket_reduce(kzg_ctx *ctx, int lane, const MsmPoint *buf0, const MsmPoint *buf1, const uint32_t *starts, int Btot, int C,
                       const MsmState *state, MsmPoint *rows, MsmPoint *cols, MsmPoint *red0, MsmPoint *red1, MsmPoint *chunks,
                       MsmPoint *sum_scratch, MsmPoint *scratch3, MsmPoint *result) {
    hipStream_t st = ctx->lanes[lane].stream;
    const int R = Btot / C;
    MsmPoint *red[2] = {red0, red1};
    KZG_LAUNCH(ctx, st, "k_rows8", k_rows8, (Btot / 8 + 63) / 64, 64, 0, buf0, buf1, starts, Btot, C, state, rows);
    KZG_LAUNCH(ctx, st, "k_cols8", k_cols8, (Btot / 8 + 63) / 64, 64, 0, buf0, buf1, starts, Btot, C, state, cols);
    MsmPoint *rowsum = reduce_groups(ctx, st, rows, (size_t)R, (size_t)C / 8, red);  // R points
    // the column reduction re-uses the ping-pong buffers: park the row sums in `rows` (free now)
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(rows, rowsum, (size_t)R * sizeof(MsmPoint), hipMemcpyDeviceToDevice, st));
    MsmPoint *colsum = reduce_groups(ctx, st, cols, (size_t)C, (size_t)R / 8, red);  // C points
    int nr = (R - 1 + REDUCE_CH - 1) / REDUCE_CH, nc = (C + REDUCE_CH - 1) / REDUCE_CH;
    MsmPoint *sumr = nullptr, *sumc = nullptr;
    MsmPoint *fin = scratch3;  // fin[0] = sum_r r Row_r, fin[1] = sum_c (c+1) Col_c
    if (R > 1) {
        KZG_LAUNCH(ctx, st, "k_weighted_chunks", k_weighted_chunks, (nr + 63) / 64, 64, 0, rows + 1, R - 1, REDUCE_CH, 0, chunks);
        KZG_TRY(sum_points_run(ctx, lane, chunks, nr, sum_scratch, &sumr));
        KZG_HIP_CHECK(ctx, hipMemcpyAsync(fin, sumr, sizeof(MsmPoint), hipMemcpyDeviceToDevice, st));
    } else {
        KZG_TRY(point_set_infinity(ctx, st, fin));
    }
    KZG_LAUNCH(ctx, st, "k_weighted_chunks", k_weighted_chunks, (nc + 63) / 64, 64, 0, colsum, C, REDUCE_CH, 0, chunks);
    KZG_TRY(sum_points_run(ctx, lane, chunks, nc, sum_scratch, &sumc));
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(fin + 1, sumc, sizeof(MsmPoint), hipMemcpyDeviceToDevice, st));
    int shift = 0;
    while ((1 << shift) < C) shift++;
    KZG_LAUNCH(ctx, st, "k_combine_shifted", k_combine_shifted, 1, 1, 0, fin, shift, fin + 1, result);
    return KZG_OK;
}

}  // namespace kzg
