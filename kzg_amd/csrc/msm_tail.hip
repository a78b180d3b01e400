// msm_tail.hip -- everything of an MSM after the bucket accumulation kernel: fold the per-thread partial sums into ONE point
// per bucket (a dense array), then sum_b (b + 1) B_b and the final point.  Shared by the narrow (c <= 17) and the wide
// (18 <= c <= 20) pipelines of msm.hip.  Own translation unit: these kernels are latency-bound (a few waves, every
// instruction executed once), and what suits them is not what suits k_accum_affine.
//
// Latency, not work, is what this stage costs (one XYZZ addition is ~6300 instructions = ~10 us for a lone wave), so every
// kernel here is organised for DEPTH: lanes of a wave cooperate through shuffles, partial results are combined by butterfly
// reductions, and the weighted sum is decomposed so that no thread ever runs a long chain:
//
//   k_fold_dense     bucket b's partials (a contiguous run of the round-1 list) are summed by a group of G lanes -- strided
//                    sequential sums of <= 8, then log2 G butterfly steps -- into dense[b].  G is chosen on the host from the
//                    expected partials per bucket; a bucket with more than 8 G partials is handed to
//   k_fold_overflow  as slices of 512 partials, one wave per slice; the wave that completes a bucket's last slice (atomic
//                    arrival counter, no spinning) adds the slice sums.  Returns at once when there is no such bucket, which is
//                    the normal case; skewed inputs (the carry bucket of u64-valued scalars, all-equal scalars) stay bounded:
//                    depth <= 8 + 6 + ceil(slices / 64) + 6 additions.
//   k_rc_sums        with b = r C + c:  sum_b (b+1) B_b = C sum_r r Row_r + sum_c (c+1) Col_c.  One wave per row sum and one
//                    per column sum: C/64 (R/64) sequential additions per lane + 6 butterfly steps.
//   k_weighted_bits  sum_i i X_i = sum_j 2^j (sum_{i : bit j of i} X_i): one wave per bit of the row index, per bit of the
//                    column index, and one for the plain total of the columns (the "+1").
//   k_reduce_final   <= 21 lanes: lane i doubles its masked sum shift_i times, a 32-lane butterfly adds them up.
//
// For c = 17 (256 x 256 buckets): 3 + 9 + 8 + 15 ~ 35 additions deep, against 43 (8-bucket running sums + a 31-step
// scalar multiplication per chunk) + 21 (seven tree-sum launches) + two fold rounds before.
#include "msm_internal.h"

// Experiment switch: the 256-thread tail kernels compiled for KZG_TAIL_WAVES waves per SIMD (VGPR budget 512 / waves), so that they
// fit next to a 3-wave accumulation kernel (-DKZG_ACCUM_WAVES=3).  Default: no bound beyond the block size.
#if defined(KZG_TAIL_WAVES)
#define KZG_TAIL_LB __launch_bounds__(256, KZG_TAIL_WAVES)
#else
#define KZG_TAIL_LB __launch_bounds__(256)
#endif

namespace kzg {

constexpr int FOLD_SEQ = 8;     // sequential additions per lane in k_fold_dense before a bucket counts as overflowing
constexpr int OVF_SLICE = 512;  // partials per overflow task (one wave: 8 per lane + butterfly)

__device__ __forceinline__ MsmPoint shfl_xor_point(const MsmPoint &p, int mask) {
    MsmPoint r;
#pragma unroll
    for (int i = 0; i < F30_N; i++) {
        r.x.v[i] = __shfl_xor(p.x.v[i], mask, 64);
        r.y.v[i] = __shfl_xor(p.y.v[i], mask, 64);
        r.zz.v[i] = __shfl_xor(p.zz.v[i], mask, 64);
        r.zzz.v[i] = __shfl_xor(p.zzz.v[i], mask, 64);
    }
    r.inf = (uint32_t)__shfl_xor((int)p.inf, mask, 64);
    r.pad[0] = r.pad[1] = r.pad[2] = 0;
    return r;
}

// sum over `width` consecutive lanes (a power of two); every lane of the group ends with the total
template <int WIDTH>
__device__ __forceinline__ MsmPoint butterfly_sum(MsmPoint acc) {
#pragma nounroll
    for (int off = WIDTH / 2; off >= 1; off >>= 1) acc = g1_add30(acc, shfl_xor_point(acc, off));
    return acc;
}

// sum of p[lo .. hi) by one wave; every lane returns the total
__device__ __forceinline__ MsmPoint wave_sum(const MsmPoint *p, uint32_t lo, uint32_t hi, int lane) {
    MsmPoint acc = MsmPoint::infinity();
    for (uint32_t k = lo + (uint32_t)lane; k < hi; k += 64) acc = g1_add30(acc, p[k]);
    return butterfly_sum<64>(acc);
}

template <int G>
__global__ KZG_TAIL_LB void k_fold_dense(const MsmPoint *part, const uint32_t *s1, int B, MsmPoint *dense, MsmState *st,
                                                    uint32_t *tasks, uint32_t *arrive) {
    KZG_SIDE_PRIO_STMT;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = gid / G, g = gid % G;
    if (b >= (uint32_t)B) return;
    const uint32_t s = s1[b], cnt = s1[b + 1] - s;
    if (cnt > (uint32_t)(FOLD_SEQ * G)) {  // uniform over the group
        if (g == 0) {
            const uint32_t nsl = (cnt + OVF_SLICE - 1) / OVF_SLICE;
            const uint32_t base = atomicAdd(&st->ovf_tasks, nsl);
            for (uint32_t k = 0; k < nsl; k++) tasks[base + k] = (b << 12) | k;
            arrive[b] = 0;
        }
        return;
    }
    MsmPoint acc = MsmPoint::infinity();
    for (uint32_t k = g; k < cnt; k += G) acc = g1_add30(acc, part[s + k]);
    acc = butterfly_sum<G>(acc);
    if (g == 0) dense[b] = acc;
}

__global__ KZG_TAIL_LB void k_fold_overflow(const MsmPoint *part, MsmPoint *scratch, const uint32_t *s1, MsmPoint *dense,
                                                       const MsmState *st, const uint32_t *tasks, uint32_t *arrive) {
    KZG_SIDE_PRIO_STMT;
    const uint32_t T = st->ovf_tasks;
    if (T == 0) return;
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t t = wave; t < T; t += nwaves) {
        const uint32_t e = tasks[t], b = e >> 12, sl = e & 0xfffu;
        const uint32_t s0 = s1[b], cnt = s1[b + 1] - s0, nsl = (cnt + OVF_SLICE - 1) / OVF_SLICE;
        const uint32_t lo = s0 + sl * OVF_SLICE, end = s0 + cnt, hi = lo + OVF_SLICE < end ? lo + OVF_SLICE : end;
        MsmPoint acc = wave_sum(part, lo, hi, lane);
        if (nsl == 1) {
            if (lane == 0) dense[b] = acc;
            continue;
        }
        // slice sums go to the bucket's own offsets in the scratch list (nsl <= cnt); the last wave to arrive adds them
        if (lane == 0) scratch[s0 + sl] = acc;
        __threadfence();
        uint32_t old = 0;
        if (lane == 0) old = atomicAdd(&arrive[b], 1u);
        old = (uint32_t)__shfl((int)old, 0, 64);
        if (old == nsl - 1) {
            __threadfence();
            acc = wave_sum(scratch, s0, s0 + nsl, lane);
            if (lane == 0) dense[b] = acc;
        }
    }
}

__global__ KZG_TAIL_LB void k_rc_sums(const MsmPoint *dense, int Rn, int Cn, MsmPoint *rows, MsmPoint *cols) {
    KZG_SIDE_PRIO_STMT;
    const int lane = threadIdx.x & 63;
    const int wv = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (wv >= Rn + Cn) return;
    MsmPoint acc = MsmPoint::infinity();
    if (wv < Rn) {
        const MsmPoint *row = dense + (size_t)wv * Cn;
        for (int c = lane; c < Cn; c += 64) acc = g1_add30(acc, row[c]);
    } else {
        const MsmPoint *col = dense + (wv - Rn);
        for (int r = lane; r < Rn; r += 64) acc = g1_add30(acc, col[(size_t)r * Cn]);
    }
    acc = butterfly_sum<64>(acc);
    if (lane == 0) (wv < Rn ? rows[wv] : cols[wv - Rn]) = acc;
}

// Throughput form of k_rc_sums for a deep batched pipeline (two or more MSMs per lane), where lane-time counts and depth does not: RC_LANES lanes per row /
// column sum, each adding Cn / RC_LANES points in sequence before a log2(RC_LANES)-level butterfly -- 35 wave-additions per 8
// sums instead of 72 (a butterfly level costs a full addition for half the useful work of the level before).
constexpr int RC_LANES = 8;
__global__ KZG_TAIL_LB void k_rc_sums_t(const MsmPoint *dense, int Rn, int Cn, MsmPoint *rows, MsmPoint *cols) {
    KZG_SIDE_PRIO_STMT;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int sum = gid / RC_LANES, l = gid % RC_LANES;
    if (sum >= Rn + Cn) return;  // uniform over the lane group
    MsmPoint acc = MsmPoint::infinity();
    if (sum < Rn) {
        const MsmPoint *row = dense + (size_t)sum * Cn;
#pragma nounroll
        for (int c = l; c < Cn; c += RC_LANES) acc = g1_add30(acc, row[c]);
    } else {
        const MsmPoint *col = dense + (sum - Rn);
#pragma nounroll
        for (int r = l; r < Rn; r += RC_LANES) acc = g1_add30(acc, col[(size_t)r * Cn]);
    }
    acc = butterfly_sum<RC_LANES>(acc);
    if (l == 0) (sum < Rn ? rows[sum] : cols[sum - Rn]) = acc;
}

// Q[j] (j < lr): sum of rows whose index has bit j set; Q[lr + j] (j < lc): the same for the columns; Q[lr + lc]: all columns
__global__ KZG_TAIL_LB void k_weighted_bits(const MsmPoint *rows, const MsmPoint *cols, int lr, int lc, MsmPoint *Q) {
    KZG_SIDE_PRIO_STMT;
    const int lane = threadIdx.x & 63;
    const int wv = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (wv > lr + lc) return;
    const MsmPoint *X = wv < lr ? rows : cols;
    const int size = wv < lr ? (1 << lr) : (1 << lc);
    const int j = wv < lr ? wv : (wv < lr + lc ? wv - lr : -1);
    MsmPoint acc = MsmPoint::infinity();
    if (j >= 0) {
        for (int t = lane; t < size / 2; t += 64) {
            const int i = ((t >> j) << (j + 1)) | (1 << j) | (t & ((1 << j) - 1));
            acc = g1_add30(acc, X[i]);
        }
    } else {
        for (int i = lane; i < size; i += 64) acc = g1_add30(acc, X[i]);
    }
    acc = butterfly_sum<64>(acc);
    if (lane == 0) Q[wv] = acc;
}

// result = sum_{j < lr} 2^(j + lc) Q[j] + sum_{j < lc} 2^j Q[lr + j] + Q[lr + lc]        (lr + lc + 1 <= 32 lanes)
//        = sum_b b B_b + sum_b B_b = sum_b (b + 1) B_b.
// odd (positional tables: bucket b holds the digits of magnitude 2 b + 1): 2 sum_b b B_b + sum_b B_b -- every bit slice one
// doubling more, the plain total as it is.
__global__ __launch_bounds__(64) void k_reduce_final(const MsmPoint *Q, int lr, int lc, MsmPoint *result, int odd) {
    KZG_SIDE_PRIO_STMT;
    const int lane = threadIdx.x;
    const int cnt = lr + lc + 1;
    MsmPoint p = lane < cnt ? Q[lane] : MsmPoint::infinity();
    const int shift = lane < lr ? lane + lc + odd : (lane < lr + lc ? lane - lr + odd : 0);
    const int maxshift = (lr > 0 ? lr + lc - 1 : (lc > 0 ? lc - 1 : 0)) + odd;
    for (int k = 0; k < maxshift; k++)
        if (k < shift && lane < cnt) p = g1_dbl30(p);
    p = butterfly_sum<32>(p);
    if (lane == 0) *result = p;
}

// ---------------------------------------------------------------------------------------------
// latency mode: four lanes per point operation
// ---------------------------------------------------------------------------------------------
// A lone MSM (KZGProver::commit as one blocking call) pays the tail's DEPTH: ~35 point operations of ~16 us each, one wave per
// SIMD, nothing to overlap with.  An XYZZ addition is 14 field multiplications but only 4 deep, a doubling 9 and 3 deep, so a
// QUAD of four consecutive lanes holding identical copies of the operands computes one point operation in 4 (3) multiplication
// rounds: every lane runs the same mul30 on operands selected by its role (lane & 3), the four products are broadcast inside
// the quad with DPP quad_perm moves, and the cheap linear steps are done redundantly by all four.  Same formulas, same
// intermediate values, hence the same bits as g1_add30 / g1_dbl30 (the squarings become plain multiplications of equal
// operands, which produce identical column sums).  ~2.4x lower latency for 4x the lanes: used when one MSM has the GPU to itself
// (the batched pipeline keeps the one-lane-per-point kernels above, where lane-time is what counts).  Measured at c = 17, same
// box: k_weighted_bits 0.137 -> 0.086 ms, k_reduce_final 0.204 -> 0.126 ms.
template <int SRC>
__device__ __forceinline__ Fq30 quad_bcast(const Fq30 &v) {
    Fq30 r;
#pragma unroll
    for (int i = 0; i < F30_N; i++) r.v[i] = __builtin_amdgcn_update_dpp(0, v.v[i], SRC * 0x55, 0xf, 0xf, false);
    return r;
}

__device__ __forceinline__ Fq30 sel4(int role, const Fq30 &a0, const Fq30 &a1, const Fq30 &a2, const Fq30 &a3) {
    Fq30 r;
#pragma unroll
    for (int i = 0; i < F30_N; i++) {
        int32_t v = a0.v[i];
        v = role == 1 ? a1.v[i] : v;
        v = role == 2 ? a2.v[i] : v;
        v = role == 3 ? a3.v[i] : v;
        r.v[i] = v;
    }
    return r;
}

// one multiplication round: lane `role` multiplies (a_role, b_role); every lane gets all four products
__device__ __forceinline__ void quad_round(int role, const Fq30 &a0, const Fq30 &b0, const Fq30 &a1, const Fq30 &b1, const Fq30 &a2,
                                           const Fq30 &b2, const Fq30 &a3, const Fq30 &b3, Fq30 &p0, Fq30 &p1, Fq30 &p2, Fq30 &p3) {
    const Fq30 t = mul30(sel4(role, a0, a1, a2, a3), sel4(role, b0, b1, b2, b3));
    p0 = quad_bcast<0>(t);
    p1 = quad_bcast<1>(t);
    p2 = quad_bcast<2>(t);
    p3 = quad_bcast<3>(t);
}

// the rare same-x case of an addition (doubling or infinity): the one-lane code, run redundantly by all four lanes
__device__ __noinline__ void add_same_x(const MsmPoint *p, const MsmPoint *q, MsmPoint *r) { *r = g1_add30(*p, *q); }

// g1_dbl30 by a quad (all four lanes hold p; all four return the result)
__device__ __forceinline__ MsmPoint qdbl(const MsmPoint &p, int role) {
    if (p.inf) return p;
    const Fq30 U = times2_30(p.y);
    Fq30 V, XX, d2, d3;
    quad_round(role, U, U, p.x, p.x, U, U, p.x, p.x, V, XX, d2, d3);
    if (is_zero30(V)) return MsmPoint::infinity();
    const Fq30 Mm = times3_30(XX);
    Fq30 W, S, MM, ZZ3;
    quad_round(role, U, V, p.x, V, Mm, Mm, V, p.zz, W, S, MM, ZZ3);
    MsmPoint r;
    r.inf = 0;
    r.pad[0] = r.pad[1] = r.pad[2] = 0;
    r.x = sub30(MM, times2_30(S));
    Fq30 MY, WY, ZZZ3;
    quad_round(role, Mm, sub30(S, r.x), W, p.y, W, p.zzz, W, p.zzz, MY, WY, ZZZ3, d3);
    r.y = sub30(MY, WY);
    r.zz = ZZ3;
    r.zzz = ZZZ3;
    return r;
}

// g1_add30 by a quad
__device__ __forceinline__ MsmPoint qadd(const MsmPoint &p, const MsmPoint &q, int role) {
    if (q.inf) return p;
    if (p.inf) return q;
    Fq30 U1, U2, S1, S2;
    quad_round(role, p.x, q.zz, q.x, p.zz, p.y, q.zzz, q.y, p.zzz, U1, U2, S1, S2);
    const Fq30 Pp = sub30(U2, U1), R = sub30(S2, S1);
    Fq30 PP, RR, ZZ12, ZZZ12;
    quad_round(role, Pp, Pp, R, R, p.zz, q.zz, p.zzz, q.zzz, PP, RR, ZZ12, ZZZ12);
    if (is_zero30(PP)) {  // same x: doubling or infinity (copies: only they live in scratch memory, and only on this path)
        MsmPoint pc = p, qc = q, r;
        add_same_x(&pc, &qc, &r);
        return r;
    }
    Fq30 PPP, Q, ZZ3, d3;
    quad_round(role, Pp, PP, U1, PP, ZZ12, PP, ZZ12, PP, PPP, Q, ZZ3, d3);
    MsmPoint r;
    r.inf = 0;
    r.pad[0] = r.pad[1] = r.pad[2] = 0;
    r.x = sub30(RR, add2x30(PPP, Q));
    Fq30 Bm, RY, ZZZ3;
    quad_round(role, S1, PPP, R, sub30(Q, r.x), ZZZ12, PPP, ZZZ12, PPP, Bm, RY, ZZZ3, d3);
    r.y = sub30(RY, Bm);
    r.zz = ZZ3;
    r.zzz = ZZZ3;
    return r;
}

// sum over the quads of a wave whose quad index differs in the bits of (QUADS - 1); every participating quad gets the total
template <int QUADS>
__device__ __forceinline__ MsmPoint quad_butterfly(MsmPoint acc, int role) {
#pragma nounroll
    for (int off = QUADS / 2; off >= 1; off >>= 1) acc = qadd(acc, shfl_xor_point(acc, 4 * off), role);
    return acc;
}

// sum over all quads of a block of WAVES waves (lds: WAVES points); every quad of the block returns the total
template <int WAVES>
__device__ __forceinline__ MsmPoint block_quad_sum(MsmPoint acc, MsmPoint *lds, int role) {
    acc = quad_butterfly<16>(acc, role);
    if (WAVES == 1) return acc;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ql = lane >> 2;
    if (lane == 0) lds[wave] = acc;
    __syncthreads();
    MsmPoint t = ql < WAVES ? lds[ql] : MsmPoint::infinity();
    return quad_butterfly<WAVES>(t, role);
}

__global__ KZG_TAIL_LB void k_weighted_bits_q(const MsmPoint *rows, const MsmPoint *cols, int lr, int lc, MsmPoint *Q) {
    KZG_SIDE_PRIO_STMT;
    __shared__ MsmPoint lds[4];
    const int role = threadIdx.x & 3, qd = threadIdx.x >> 2;
    const int wv = blockIdx.x;  // <= lr + lc
    const MsmPoint *X = wv < lr ? rows : cols;
    const int size = wv < lr ? (1 << lr) : (1 << lc);
    const int j = wv < lr ? wv : (wv < lr + lc ? wv - lr : -1);
    MsmPoint acc = MsmPoint::infinity();
    if (j >= 0) {
        for (int t = qd; t < size / 2; t += 64) {
            const int i = ((t >> j) << (j + 1)) | (1 << j) | (t & ((1 << j) - 1));
            acc = qadd(acc, X[i], role);
        }
    } else {
        for (int i = qd; i < size; i += 64) acc = qadd(acc, X[i], role);
    }
    acc = block_quad_sum<4>(acc, lds, role);
    if (threadIdx.x == 0) Q[wv] = acc;
}

// 32 quads (two waves): quad i doubles Q[i] shift_i times, then the block sum
__global__ __launch_bounds__(128) void k_reduce_final_q(const MsmPoint *Q, int lr, int lc, MsmPoint *result, int odd) {
    KZG_SIDE_PRIO_STMT;
    __shared__ MsmPoint lds[2];
    const int role = threadIdx.x & 3, qd = threadIdx.x >> 2;
    const int cnt = lr + lc + 1;
    MsmPoint p = qd < cnt ? Q[qd] : MsmPoint::infinity();
    const int shift = qd < lr ? qd + lc + odd : (qd < lr + lc ? qd - lr + odd : 0);
    const int maxshift = (lr > 0 ? lr + lc - 1 : (lc > 0 ? lc - 1 : 0)) + odd;
#pragma nounroll
    for (int k = 0; k < maxshift; k++)
        if (k < shift && qd < cnt) p = qdbl(p, role);
    p = block_quad_sum<2>(p, lds, role);
    if (threadIdx.x == 0) *result = p;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
void tail_shape(int B, int *Rn, int *Cn, int *lr, int *lc) {
    int lb = 0;
    while ((1 << lb) < B) lb++;
    *lc = (lb + 1) / 2;
    *lr = lb - *lc;
    *Cn = 1 << *lc;
    *Rn = 1 << *lr;
}

TailLayout tail_layout(int B, size_t T1_max) {
    TailLayout L;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t r = o;
        o = align_up(o + bytes, 256);
        return r;
    };
    int Rn, Cn, lr, lc;
    tail_shape(B, &Rn, &Cn, &lr, &lc);
    L.off_dense = take((size_t)B * sizeof(MsmPoint));
    L.off_rows = take((size_t)Rn * sizeof(MsmPoint));
    L.off_cols = take((size_t)Cn * sizeof(MsmPoint));
    L.off_Q = take(32 * sizeof(MsmPoint));
    L.off_tasks = take((T1_max / OVF_SLICE + (size_t)B + 64) * 4);
    L.off_arrive = take((size_t)B * 4);
    L.off_result = take(sizeof(MsmPoint));
    L.bytes = o;
    return L;
}

// part: the round-1 partial list (ordered by bucket, bucket b = [s1[b], s1[b+1])); scratch: a list of the same capacity;
// expected_partials: how many partials round 1 is expected to emit (chooses the lane-group width of the fold)
int msm_tail_run(kzg_ctx *ctx, hipStream_t st, const MsmMode &mode, const MsmPoint *part, MsmPoint *scratch, const uint32_t *s1, int B,
                 size_t expected_partials, MsmState *state, char *tail_base, const TailLayout &L, MsmPoint **d_result, bool odd_weights) {
    MsmPoint *dense = (MsmPoint *)(tail_base + L.off_dense), *rows = (MsmPoint *)(tail_base + L.off_rows);
    MsmPoint *cols = (MsmPoint *)(tail_base + L.off_cols), *Q = (MsmPoint *)(tail_base + L.off_Q);
    uint32_t *tasks = (uint32_t *)(tail_base + L.off_tasks), *arrive = (uint32_t *)(tail_base + L.off_arrive);
    MsmPoint *result = (MsmPoint *)(tail_base + L.off_result);
    int Rn, Cn, lr, lc;
    tail_shape(B, &Rn, &Cn, &lr, &lc);
    // lane-group width: twice the expected partials per bucket still fit FOLD_SEQ sequential additions per lane 4 times over
    // (deep batched pipeline: one lane per bucket as long as the expected run fits its FOLD_SEQ sequential additions twice over --
    // a butterfly level is a full addition for every lane of the group, wasted lane-time when depth does not matter)
    size_t avg = expected_partials / (size_t)B + 1;
    int G = 1;
    if (!mode.tail_wide) {
        while (G < 64 && (size_t)G * 2 < avg) G *= 2;
    } else {
        while (G < 64 && (size_t)G * (FOLD_SEQ / 2) < avg) G *= 2;
    }
    const int wpb = TAIL_THREADS / 64;
    const unsigned grid = (unsigned)(((size_t)B * G + TAIL_THREADS - 1) / TAIL_THREADS);
#define KZG_FOLD(GG)                                                                                                      \
    case GG:                                                                                                              \
        KZG_LAUNCH(ctx, st, "k_fold_dense", k_fold_dense<GG>, grid, TAIL_THREADS, 0, part, s1, B, dense, state, tasks, arrive); \
        break;
    switch (G) {
        KZG_FOLD(1) KZG_FOLD(2) KZG_FOLD(4) KZG_FOLD(8) KZG_FOLD(16) KZG_FOLD(32) KZG_FOLD(64)
    }
#undef KZG_FOLD
    KZG_LAUNCH(ctx, st, "k_fold_overflow", k_fold_overflow, 256, TAIL_THREADS, 0, part, scratch, s1, dense, state, tasks, arrive);
    if (mode.tail_wide && Rn >= RC_LANES && Cn >= RC_LANES) {
        KZG_LAUNCH(ctx, st, "k_rc_sums", k_rc_sums_t, ((Rn + Cn) * RC_LANES + 255) / 256, 256, 0, dense, Rn, Cn, rows, cols);
    } else {
        KZG_LAUNCH(ctx, st, "k_rc_sums", k_rc_sums, (Rn + Cn + wpb - 1) / wpb, TAIL_THREADS, 0, dense, Rn, Cn, rows, cols);
    }
    if (mode.tail_quads) {
        // one MSM alone on the GPU: the two depth-bound kernels run four lanes per point operation.  (Fold and row / column
        // sums are bound by their ~2 additions per bucket, not by depth: quads were measured slower there, 0.25 against 0.07 ms
        // and 0.17 against 0.14 ms at c = 17: a quad addition takes ~10 us against 16 for one lane, so a row sum by 32 or 64
        // quads -- 8 or 4 sequential additions + 6 levels -- loses to one wave's 3 + 6.  Two one-lane waves per row sum, 1 + 6 + 1
        // deep, measured 0.20 ms: the dispatcher does not put the 1024 waves on 1024 different SIMDs.)
        KZG_LAUNCH(ctx, st, "k_weighted_bits", k_weighted_bits_q, lr + lc + 1, 256, 0, rows, cols, lr, lc, Q);
        KZG_LAUNCH(ctx, st, "k_reduce_final", k_reduce_final_q, 1, 128, 0, Q, lr, lc, result, odd_weights ? 1 : 0);
    } else {
        KZG_LAUNCH(ctx, st, "k_weighted_bits", k_weighted_bits, (lr + lc + 1 + wpb - 1) / wpb, TAIL_THREADS, 0, rows, cols, lr, lc, Q);
        KZG_LAUNCH(ctx, st, "k_reduce_final", k_reduce_final, 1, 64, 0, Q, lr, lc, result, odd_weights ? 1 : 0);
    }
    *d_result = result;
    return KZG_OK;
}

}  // namespace kzg
