// msm_internal.h -- declarations shared by msm.hip (sort, bucket accumulation, all orchestration), msm_wide.hip (the
// kernels only the wide-window path uses) and msm_tail.hip (everything after the accumulation).  Separate translation units
// so that they do not perturb the code generation of the hot kernel: a same-box A/B showed k_accum_affine 2.5 % slower with
// everything in one TU.
#pragma once
#include "common.h"

// Wave priority of the sort and tail kernels (s_setprio, 0..3): they are the short, latency-bound kernels of the pipeline and run
// next to the long accumulation kernels, whose waves are older and therefore win the VALU arbitration (priority, then age).
#ifndef KZG_SIDE_PRIO
#define KZG_SIDE_PRIO 0
#endif
#if KZG_SIDE_PRIO > 0
#define KZG_SIDE_PRIO_STMT __builtin_amdgcn_s_setprio(KZG_SIDE_PRIO)
#else
#define KZG_SIDE_PRIO_STMT ((void)0)
#endif

// Fewest sorted entries one thread of k_accum_affine folds (the equal-split chunk E = max(ceil(M / resident threads), this)): a thread
// pays a bucket search, its first gathers and a 224-byte partial whatever its share, and every thread beyond the first of a bucket is
// one more addition for the fold.
#ifndef KZG_ACCUM_MIN_CHUNK
#define KZG_ACCUM_MIN_CHUNK 8
#endif

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
// Two-level sort of the widest window (c = 20: 13 windows, 2^19 buckets = 1024 bins of 512), msm_wide.hip.  Same structure as the
// c = 17 sort of msm.hip (level 1 by bin with the chunk sorted in LDS first, level 2 one block per bin), 8-byte records.
constexpr int NBINS = 1024;         // level-1 bins of both two-level sorts
constexpr int SORT20_C = 20, SORT20_W = 13, SORT20_SHIFT = 9, SORT20_BUCKETS = 1 << SORT20_SHIFT;
int sort20_hist(kzg_ctx *ctx, hipStream_t st, const void *d_scalars, size_t n, int sfmt, int G2, uint32_t *bins);
// after k_bin_scan (msm.hip): records by bin, then entries by bucket; bucket_start[0 .. B], state (M, E, ntasks), s1[0 .. B]
int sort20_place(kzg_ctx *ctx, hipStream_t st, const void *d_scalars, size_t n, int sfmt, int G2, const uint32_t *bins,
                 const uint32_t *bin_total, uint32_t *bin_base, uint32_t row_stride, uint32_t idx_base, void *rec, uint32_t *entries,
                 uint32_t *bucket_start, uint32_t *s1, MsmState *state, uint32_t slots, uint32_t *segsums, uint32_t *segmaxs,
                 uint32_t *segtotal, uint32_t *hv, bool sliced, uint32_t *lane_heavy, uint32_t heavy_seq);
// Level 2 of both two-level sorts (one block per bin) + the bins that are sorted in slices (msm_wide.hip).  kind: 4 / 8 = the
// c = 17 records of msm.hip (64 buckets per bin; bucket sizes -> total[], k_scan_b_bins follows), 20 = c = 20 (512 buckets per
// bin; bucket starts and the equal-split state).  bucket_start is scratch for the heavy bins in the first case as well.
// hv: descriptor (2 + 2 NBINS words, padded; written by block 0 of k_heavy_count) followed by the per-slice bucket counts
// (<= 2560 slices x buckets per bin).  sliced = false: every bin is sorted by its own block and the slice kernels are not enqueued.
// A block that meets an oversized bin writes heavy_seq to *lane_heavy either way (kzg_ctx::d_lane_heavy).
constexpr size_t SORT2_HV_WORDS = 2112;
constexpr size_t sort2_hv_bytes(int buckets_per_bin) { return (SORT2_HV_WORDS + (size_t)2560 * buckets_per_bin) * 4; }
int sort2_level2(kzg_ctx *ctx, hipStream_t st, int kind, const void *rec, const uint32_t *bin_base, const uint32_t *bin_total, uint32_t *hv,
                 uint32_t *entries, uint32_t *total, uint32_t *bucket_start, MsmState *state, uint32_t slots, bool sliced, uint32_t *lane_heavy,
                 uint32_t heavy_seq);

// msm_tail.hip: everything after the bucket accumulation (fold to one point per bucket, sum (b+1) B_b)
struct TailLayout {
    size_t off_dense, off_rows, off_cols, off_Q, off_tasks, off_arrive, off_result, bytes;
};
TailLayout tail_layout(int B, size_t T1_max);
int msm_tail_run(kzg_ctx *ctx, hipStream_t st, const MsmMode &mode, const MsmPoint *part, MsmPoint *scratch, const uint32_t *s1, int B,
                 size_t expected_partials, MsmState *state, char *tail_base, const TailLayout &L, MsmPoint **d_result,
                 bool odd_weights = false);

}  // namespace kzg
