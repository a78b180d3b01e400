// capi.hip -- the extern "C" entry points declared in include/kzg_mi355x.h that mirror KZGProver / KZGProverEvalForm / EvaluationDomain
// method by method (MSM / commit, the batched pipeline, witnesses, NTT, polynomial helpers, verify_poly, the unit-test hooks), with the
// staging and result helpers they share.  The context, lanes, streams and profiling they run on live in runtime.hip.
#include <algorithm>

#include "common.h"
#include "emit.h"

namespace kzg {

// ---------------------------------------------------------------------------------------------
// helpers shared by the entry points
// ---------------------------------------------------------------------------------------------
static int host_scalar(kzg_ctx *ctx, const void *s, int sfmt, Fr *mont) {
    Fr v;
    memcpy(v.v, s, 32);
    if (sfmt == KZG_FR_CANONICAL_LE_32) {
        if (!is_canonical(v)) return fail(ctx, KZG_ERR_SHAPE, "scalar not canonical (>= r)");
        v = to_mont(v);
    } else if (sfmt != KZG_FR_MONT_LE_32) {
        return fail(ctx, KZG_ERR_SHAPE, "unknown scalar format");
    }
    *mont = v;
    return KZG_OK;
}

static void scalar_out(const Fr &mont, int sfmt, void *dst) {
    Fr v = sfmt == KZG_FR_CANONICAL_LE_32 ? from_mont(mont) : mont;
    memcpy(dst, v.v, 32);
}

// bring `bytes` of input to the device (no-op for device-resident input)
static int stage_in(kzg_ctx *ctx, int lane, const void *src, size_t bytes, int flags, const void **dptr) {
    if (flags & KZG_IN_DEVICE) {
        *dptr = src;
        return KZG_OK;
    }
    void *d = lane_alloc(ctx, lane, bytes ? bytes : 16);
    if (!d) return fail(ctx, KZG_ERR_ALLOC, "input staging not reserved");
    if (bytes) KZG_HIP_CHECK(ctx, hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, ctx->lanes[lane].stream));
    *dptr = d;
    return KZG_OK;
}

static size_t stage_bytes(size_t bytes, int flags) { return (flags & KZG_IN_DEVICE) ? 0 : align_up(bytes + 256, 256); }

// The lane's last sort with the result (oversized bins: what the next calls decide on, common.h heavy_last): 4 bytes into the
// lane's pinned buffer before the wait, looked at after it.
static int heavy_pickup_enqueue(kzg_ctx *ctx, int lane) {
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(ctx->lanes[lane].pinned + 512, ctx->d_lane_heavy + lane, 4, hipMemcpyDeviceToHost, ctx->lanes[lane].stream));
    return KZG_OK;
}
static void heavy_pickup_read(kzg_ctx *ctx, int lane) {
    uint32_t hseq;
    memcpy(&hseq, ctx->lanes[lane].pinned + 512, 4);
    Lane &l = ctx->lanes[lane];
    if (l.heavy_seq == l.heavy_seen) return;  // no MSM on this lane since the last look
    l.heavy_seen = l.heavy_seq;
    if (hseq == l.heavy_seq) ctx->heavy_last.store(ctx->msm_count.load(std::memory_order_relaxed), std::memory_order_relaxed);
}

// result point: XYZZ on device -> ofmt at `out` (host or device)
int finish_point(kzg_ctx *ctx, int lane, const MsmPoint *d_pt, void *out, int ofmt, int flags) {
    size_t psz = point_format_bytes(ofmt);
    if (!psz) return fail(ctx, KZG_ERR_SHAPE, "unknown G1 output format");
    hipStream_t st = ctx->lanes[lane].stream;
    KZG_TRY(lane_pinned(ctx, lane, 4096));
    if (flags & KZG_OUT_DEVICE) {
        KZG_TRY(emit_point(ctx, lane, d_pt, out, ofmt));
        KZG_TRY(heavy_pickup_enqueue(ctx, lane));
        KZG_HIP_CHECK(ctx, hipStreamSynchronize(st));
    } else {
        if (ctx->opt_host_affine) {
            // A lone result for the host: copy the XYZZ point out and convert it on the calling thread with the same code
            // (emit.h, compiled for the host): a CPU core does the Fq inversion of to_affine in a few microseconds, one GPU lane
            // needs ~90 us for it, and this sits on the critical path of every blocking commit / create_witness.
            KZG_HIP_CHECK(ctx, hipMemcpyAsync(ctx->lanes[lane].pinned, d_pt, sizeof(MsmPoint), hipMemcpyDeviceToHost, st));
            KZG_TRY(heavy_pickup_enqueue(ctx, lane));
            KZG_HIP_CHECK(ctx, hipStreamSynchronize(st));
            MsmPoint pt;
            memcpy(&pt, ctx->lanes[lane].pinned, sizeof pt);
            alignas(16) uint8_t buf[144];
            emit_one(pt, buf, ofmt);
            memcpy(out, buf, psz);
        } else {
            // the kernel writes the <= 144 bytes straight into the lane's pinned host buffer (device-mapped, coherent)
            KZG_TRY(emit_point(ctx, lane, d_pt, ctx->lanes[lane].pinned, ofmt));
            KZG_TRY(heavy_pickup_enqueue(ctx, lane));
            KZG_HIP_CHECK(ctx, hipStreamSynchronize(st));
            memcpy(out, ctx->lanes[lane].pinned, psz);
        }
    }
    heavy_pickup_read(ctx, lane);
    if (ctx->prof) prof_collect(ctx);
    return KZG_OK;
}

static int check_sfmt(kzg_ctx *ctx, int sfmt) {
    if (sfmt != KZG_FR_MONT_LE_32 && sfmt != KZG_FR_CANONICAL_LE_32) return fail(ctx, KZG_ERR_SHAPE, "unknown scalar format");
    return KZG_OK;
}

static bool is_pow2(size_t x) { return x && !(x & (x - 1)); }
// element counts whose byte sizes cannot overflow the arena arithmetic (2^40 Fr elements = 32 TiB: far beyond any device)
static bool count_ok(size_t n) { return n <= ((size_t)1 << 40); }

#ifdef KZG_TEST_HOOKS
// ---------------------------------------------------------------------------------------------
// device unit-test kernels (only in the -DKZG_TEST_HOOKS build: kzg_amd/libkzg_mi355x_hooks.so, include/kzg_mi355x_test.h)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_test_fr_mul(const Fr *a, const Fr *b, size_t n, Fr *o) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = mul(a[i], b[i]);
}
__global__ __launch_bounds__(256) void k_test_fq_mul(const Fq *a, const Fq *b, size_t n, Fq *o) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = mul(a[i], b[i]);
}
__global__ __launch_bounds__(256) void k_test_fr_inv(const Fr *a, size_t n, Fr *o) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = a[i].is_zero() ? Fr::zero() : inv(a[i]);
}
__global__ __launch_bounds__(256) void k_test_g1_add(const G1Affine *a, const G1Affine *b, size_t n, G1Affine *o) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // exercise both the mixed and the general addition on de-normalised operands
    G1Xyzz p = g1_madd(G1Xyzz::from_affine(a[i]), b[i]);
    G1Xyzz pa = g1_madd(g1_dbl(G1Xyzz::from_affine(a[i])), g1_neg(a[i]));
    G1Xyzz pb = g1_madd(g1_dbl(G1Xyzz::from_affine(b[i])), g1_neg(b[i]));
    G1Xyzz q = g1_add(pa, pb);
    G1Affine r1 = g1_to_affine(p), r2 = g1_to_affine(q);
    bool same = (r1.x == r2.x) && (r1.y == r2.y);
    o[i] = same ? r1 : G1Affine{Fq::one(), Fq::one()};  // (1,1) is not on the curve: flags a mismatch
}
__global__ __launch_bounds__(256) void k_test_g1_mul(const G1Affine *p, const Fr *k, size_t n, G1Affine *o) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr kk = k[i];
    o[i] = g1_to_affine(g1_scalar_mul(p[i], kk.v));
}

#endif  // KZG_TEST_HOOKS

}  // namespace kzg

using namespace kzg;

// ---------------------------------------------------------------------------------------------
// MSM
// ---------------------------------------------------------------------------------------------
extern "C" int kzg_msm_g1(kzg_ctx *ctx, const kzg_srs *srs, size_t offset, const void *scalars, size_t n, int sfmt,
                          int flags, void *out, int ofmt) {
    if (!ctx || !srs || !out || (!scalars && n)) return KZG_ERR_SHAPE;
    Lease ls;
    KZG_TRY(lease_lane(ctx, &ls));
    const int lane = ls.lane;
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    KZG_TRY(check_sfmt(ctx, sfmt));
    if (!point_format_bytes(ofmt)) return fail(ctx, KZG_ERR_SHAPE, "unknown G1 output format");
    if (n > srs->n || offset > srs->n - n) return fail(ctx, KZG_ERR_SHAPE, "MSM range exceeds the SRS (reference: slice index panic)");
    KZG_TRY(lane_reserve(ctx, lane, msm_workspace_bytes(srs, n) + stage_bytes(n * 32, flags) + 8192));
    const void *d_sc = nullptr;
    KZG_TRY(stage_in(ctx, lane, scalars, n * 32, flags, &d_sc));
    MsmPoint *res = nullptr;
    KZG_TRY(lease_msm(ctx, ls, srs, offset, d_sc, n, sfmt, &res));
    return finish_point(ctx, lane, res, out, ofmt, flags);
}

extern "C" int kzg_commit_coeff(kzg_ctx *ctx, const kzg_srs *srs, const void *coeffs, size_t n, int sfmt, int flags,
                                void *out, int ofmt) {
    return kzg_msm_g1(ctx, srs, 0, coeffs, n, sfmt, flags, out, ofmt);
}

extern "C" int kzg_commit_eval(kzg_ctx *ctx, const kzg_srs *lagrange, const void *evals, size_t d, int sfmt, int flags,
                               void *out, int ofmt) {
    if (!ctx || !lagrange) return KZG_ERR_SHAPE;
    if (d != lagrange->n) return fail(ctx, KZG_ERR_SHAPE, "assert!(self.d == evals.d) (src/eval_form.rs:115)");
    return kzg_msm_g1(ctx, lagrange, 0, evals, d, sfmt, flags, out, ofmt);
}

// ---- the batched pipeline shared by kzg_msm_g1_batch and kzg_witness_coeff_many -------------------------------------------
// Item b runs on lane b % nl; every bucket-accumulation kernel goes to one of the dedicated FIFO streams (the plan:
// plan_pipeline in runtime.hip).
struct BatchPipe {
    int nl = 1, nas = 0;
    uint8_t *d_out = nullptr;
    bool out_dev = false;
};

static int batch_begin(kzg_ctx *ctx, size_t batch, size_t out_bytes, void *out, int flags, BatchPipe *bp) {
    bp->out_dev = (flags & KZG_OUT_DEVICE) != 0;
    if (bp->out_dev) {
        bp->d_out = (uint8_t *)out;
    } else {
        // grow-only device staging for the results: a hipMalloc / hipFree pair per call costs a device-wide sync
        if (ctx->batch_out_bytes < out_bytes + 256) {
            if (ctx->batch_out) hipFree(ctx->batch_out);
            ctx->batch_out = nullptr;
            ctx->batch_out_bytes = 0;
            KZG_HIP_CHECK(ctx, hipMalloc((void **)&ctx->batch_out, out_bytes + 256));
            ctx->batch_out_bytes = out_bytes + 256;
        }
        bp->d_out = (uint8_t *)ctx->batch_out;
    }
    KZG_TRY(plan_pipeline(ctx, (int)std::min<size_t>(batch, (size_t)ctx->opt_streams), &bp->nl, &bp->nas));
    // at least two MSMs per lane: the lanes never run dry, so the tail is organised for lane-time instead of depth
    for (int l = 0; l < bp->nl; l++) set_lane_mode(ctx, l, bp->nl > 1, batch >= 2 * (size_t)bp->nl);
    return KZG_OK;
}

static int batch_msm(kzg_ctx *ctx, const BatchPipe &bp, size_t b, int lane, const kzg_srs *srs, size_t offset, const void *d_sc,
                     size_t n, int sfmt, MsmPoint **res, MsmPending *defer = nullptr) {
    // (Staggering the first round -- lane b starting its sort when lane b - 2 has sorted, so that the first accumulation kernel
    // does not wait for sixteen contending sorts -- measured 460 against 472 commitments/s same-box, profiles/r03_ab_kernel.txt:
    // the up-front burst of sorts is the better start.)
    if (bp.nas) {
        // a lane's two MSMs in flight (deferred tails) use different events
        const int ev = defer ? lane + (int)((b / (size_t)bp.nl) & 1) * bp.nl : lane;
        return msm_run(ctx, lane, srs, offset, d_sc, n, sfmt, res, ctx->accum_streams[b % (size_t)accum_streams_for(ctx, bp.nas, srs, n)],
                       ctx->sorted_events[ev], ctx->accum_events[ev], defer);
    }
    return msm_run(ctx, lane, srs, offset, d_sc, n, sfmt, res);
}

static int batch_end(kzg_ctx *ctx, const BatchPipe &bp, int rc, void *out, size_t out_bytes) {
    for (int l = 0; l < bp.nl; l++) hipStreamSynchronize(ctx->lanes[l].stream);
    for (int l = 0; l < bp.nl; l++) set_lane_mode(ctx, l, false, false);
    if (rc == KZG_OK && !bp.out_dev) {
        hipError_t e = hipMemcpy(out, bp.d_out, out_bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(ctx, KZG_ERR_HIP, hipGetErrorString(e));
    }
#if !defined(KZG_TIMING_NO_PICKUP)
    if (rc == KZG_OK) {  // the lanes' last sort plans (common.h heavy_last)
        uint32_t hv[KZG_MAX_LANES];
        if (hipMemcpy(hv, ctx->d_lane_heavy, sizeof hv, hipMemcpyDeviceToHost) == hipSuccess)
            for (int l = 0; l < bp.nl; l++) {
                Lane &ln = ctx->lanes[l];
                if (ln.heavy_seq == ln.heavy_seen) continue;
                ln.heavy_seen = ln.heavy_seq;
                if (hv[l] == ln.heavy_seq) ctx->heavy_last.store(ctx->msm_count.load(std::memory_order_relaxed), std::memory_order_relaxed);
            }
    }
#endif
    if (ctx->prof) prof_collect(ctx);
    return rc;
}

namespace kzg {
// kzg_msm_g1_batch with an explicit distance between consecutive scalar vectors (the sharded commit hands every device the
// slice [lo, hi) of each polynomial: stride = whole-polynomial bytes, n = hi - lo)
int msm_batch_strided(kzg_ctx *ctx, const kzg_srs *srs, size_t offset, const void *scalars, size_t n, size_t batch,
                      size_t stride_bytes, int sfmt, int flags, void *out, int ofmt) {
    Guard g(ctx);
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    KZG_TRY(check_sfmt(ctx, sfmt));
    size_t psz = point_format_bytes(ofmt);
    if (!psz) return fail(ctx, KZG_ERR_SHAPE, "unknown G1 output format");
    if (n > srs->n || offset > srs->n - n) return fail(ctx, KZG_ERR_SHAPE, "MSM range exceeds the SRS");
    if (batch == 0) return KZG_OK;
    BatchPipe bp;
    KZG_TRY(batch_begin(ctx, batch, batch * psz, out, flags, &bp));
    size_t per = align_up(msm_workspace_bytes(srs, n) + stage_bytes(n * 32, flags) + 8192, 4096);
    int rc = KZG_OK;
    // Software pipelining inside a lane (deep batches on FIFO accumulation streams): MSM b's tail is enqueued AFTER the sort of the
    // lane's next MSM (b + nl), so the accumulation queue gets its next kernel before the lane spends milliseconds in latency-bound
    // tail kernels.  Two MSMs of a lane are in flight then: two workspaces (the arena's halves) and two event pairs per lane.
    const bool defer = bp.nas > 0 && batch > (size_t)bp.nl && !ctx->opt_no_defer_tail;
    std::vector<MsmPending> pend(defer ? 2 * (size_t)bp.nl : 0);
    for (int l = 0; l < bp.nl && rc == KZG_OK; l++) rc = lane_reserve(ctx, l, defer ? 2 * per : per);
    auto finish = [&](size_t b) {
        const int l = (int)(b % bp.nl);
        MsmPoint *res = nullptr;
        int r = msm_finish(ctx, pend[(size_t)l + ((b / (size_t)bp.nl) & 1) * (size_t)bp.nl], &res);
        if (r == KZG_OK) r = emit_point(ctx, l, res, bp.d_out + b * psz, ofmt);
        return r;
    };
    for (size_t b = 0; b < batch && rc == KZG_OK; b++) {
        int l = (int)(b % bp.nl);
        const size_t half = defer ? ((b / (size_t)bp.nl) & 1) : 0;
        ctx->lanes[l].arena_used = half * per;  // stream order makes re-use of the lane arena (of this half) safe
        ctx->lanes[l].arena_limit = defer ? (half + 1) * per : 0;  // an under-estimated workspace fails (KZG_ERR_ALLOC) instead of running
                                                                   // into the other half, which belongs to the lane's other MSM in flight
        const void *d_sc = nullptr;
        rc = stage_in(ctx, l, (const uint8_t *)scalars + b * stride_bytes, n * 32, flags, &d_sc);
        MsmPoint *res = nullptr;
        if (defer) {
            MsmPending &pd = pend[(size_t)l + half * (size_t)bp.nl];
            if (rc == KZG_OK) rc = batch_msm(ctx, bp, b, l, srs, offset, d_sc, n, sfmt, &res, &pd);
            if (rc == KZG_OK && !pd.active) pd.result = res;  // (not deferred: multi-pass / wide path) finish() emits it
            if (rc == KZG_OK && b >= (size_t)bp.nl) rc = finish(b - bp.nl);
            continue;
        }
        if (rc == KZG_OK) rc = batch_msm(ctx, bp, b, l, srs, offset, d_sc, n, sfmt, &res);
        if (rc == KZG_OK) rc = emit_point(ctx, l, res, bp.d_out + b * psz, ofmt);
    }
    if (defer)
        for (size_t b = batch > (size_t)bp.nl ? batch - bp.nl : 0; b < batch && rc == KZG_OK; b++) rc = finish(b);
    for (int l = 0; l < bp.nl; l++) ctx->lanes[l].arena_limit = 0;
    return batch_end(ctx, bp, rc, out, batch * psz);
}
}  // namespace kzg

extern "C" int kzg_msm_g1_batch(kzg_ctx *ctx, const kzg_srs *srs, size_t offset, const void *scalars, size_t n,
                                size_t batch, int sfmt, int flags, void *out, int ofmt) {
    if (!ctx || !srs || !out || (!scalars && n && batch)) return KZG_ERR_SHAPE;
    if (n > SIZE_MAX / 32) return KZG_ERR_SHAPE;
    return msm_batch_strided(ctx, srs, offset, scalars, n, batch, n * 32, sfmt, flags, out, ofmt);
}

extern "C" int kzg_witness_coeff_many(kzg_ctx *ctx, const kzg_srs *srs, const void *coeffs, size_t n, const void *xs,
                                      const void *ys, size_t count, int sfmt, int flags, void *out, int ofmt, int *status) {
    // `count` calls of KZGProver::create_witness (src/coeff_form.rs:66-81) on ONE polynomial, pipelined: opening j computes its
    // quotient (Horner scan) and its MSM on lane j % streams.  status[j] = 0 or KZG_ERR_POINT_NOT_ON_POLY (p(x_j) != y_j).
    if (!ctx || !srs || !coeffs || !out || n == 0 || ((!xs || !ys) && count)) return KZG_ERR_SHAPE;
    Guard g(ctx);
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    KZG_TRY(check_sfmt(ctx, sfmt));
    size_t psz = point_format_bytes(ofmt);
    if (!psz) return fail(ctx, KZG_ERR_SHAPE, "unknown G1 output format");
    if (n - 1 > srs->n) return fail(ctx, KZG_ERR_SHAPE, "quotient longer than the SRS (reference: slice index panic)");
    if (count == 0) return KZG_OK;
    std::vector<Fr> xm(count);
    for (size_t j = 0; j < count; j++) KZG_TRY(host_scalar(ctx, (const uint8_t *)xs + 32 * j, sfmt, &xm[j]));
    BatchPipe bp;
    KZG_TRY(batch_begin(ctx, count, count * psz, out, flags, &bp));
    int rc = KZG_OK;
    // shared inputs / outputs live in the arena of one extra lane: the coefficients (when they come from the host) and p(x_j)
    const int sl = bp.nl;
    rc = ensure_lanes(ctx, sl + 1);
    if (rc == KZG_OK) rc = lane_reserve(ctx, sl, stage_bytes(n * 32, flags) + count * 32 + 4096);
    const void *d_coeffs = nullptr;
    Fr *d_px = nullptr;
    if (rc == KZG_OK) rc = stage_in(ctx, sl, coeffs, n * 32, flags, &d_coeffs);
    if (rc == KZG_OK && !(d_px = (Fr *)lane_alloc(ctx, sl, count * 32))) rc = fail(ctx, KZG_ERR_ALLOC, "workspace");
    if (rc == KZG_OK && hipStreamSynchronize(ctx->lanes[sl].stream) != hipSuccess) rc = fail(ctx, KZG_ERR_HIP, "stage-in of the coefficients");
    size_t per = msm_workspace_bytes(srs, n - 1) + n * 32 + (n / 2048 + 4) * 64 + 65536;
    for (int l = 0; l < bp.nl && rc == KZG_OK; l++) rc = lane_reserve(ctx, l, per);
    for (size_t j = 0; j < count && rc == KZG_OK; j++) {
        int l = (int)(j % bp.nl);
        ctx->lanes[l].arena_used = 0;
        Fr *dq = (Fr *)lane_alloc(ctx, l, n * 32);
        if (!dq) rc = fail(ctx, KZG_ERR_ALLOC, "workspace");
        if (rc == KZG_OK) rc = quotient_linear_run(ctx, l, (const Fr *)d_coeffs, n, xm[j], dq, d_px + j);
        MsmPoint *res = nullptr;
        if (rc == KZG_OK) rc = batch_msm(ctx, bp, j, l, srs, 0, dq, n - 1, sfmt, &res);
        if (rc == KZG_OK) rc = emit_point(ctx, l, res, bp.d_out + j * psz, ofmt);
    }
    rc = batch_end(ctx, bp, rc, out, count * psz);
    if (rc != KZG_OK) return rc;
    std::vector<uint8_t> px(count * 32);
    KZG_HIP_CHECK(ctx, hipMemcpy(px.data(), d_px, count * 32, hipMemcpyDeviceToHost));
    int worst = KZG_OK;
    for (size_t j = 0; j < count; j++) {
        // remainder of (p - y)/(X - x) is p(x) - y: Some(_) => Err(PointNotOnPolynomial)
        int sj = memcmp(px.data() + 32 * j, (const uint8_t *)ys + 32 * j, 32) != 0 ? KZG_ERR_POINT_NOT_ON_POLY : KZG_OK;
        if (status) status[j] = sj;
        if (sj) worst = sj;
    }
    if (worst && !status) return fail(ctx, KZG_ERR_POINT_NOT_ON_POLY, "point not on polynomial!");
    return KZG_OK;
}

extern "C" int kzg_witness_eval_many(kzg_ctx *ctx, const kzg_srs *lagrange, const void *evals, size_t d, const size_t *indices,
                                     size_t count, int sfmt, int flags, void *out, int ofmt) {
    // `count` calls of KZGProverEvalForm::create_witness (src/eval_form.rs:124-140) on ONE evaluation vector, pipelined
    if (!ctx || !lagrange || !evals || !out || (!indices && count)) return KZG_ERR_SHAPE;
    Guard g(ctx);
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    KZG_TRY(check_sfmt(ctx, sfmt));
    size_t psz = point_format_bytes(ofmt);
    if (!psz) return fail(ctx, KZG_ERR_SHAPE, "unknown G1 output format");
    if (!is_pow2(d)) return fail(ctx, KZG_ERR_SHAPE, "evaluation domain size must be a power of two");
    for (size_t j = 0; j < count; j++)
        if (indices[j] >= d) return fail(ctx, KZG_ERR_SHAPE, "evaluation index out of range (reference: index panic)");
    if (d > lagrange->n) return fail(ctx, KZG_ERR_SHAPE, "evaluations longer than the Lagrange SRS (reference: slice panic)");
    uint32_t log_d = (uint32_t)ilog2_ceil(d);
    if (log_d >= FR_TWO_ADICITY) return fail(ctx, KZG_ERR_DEGREE_TOO_LARGE, "domain too large");
    if (count == 0) return KZG_OK;
    BatchPipe bp;
    KZG_TRY(batch_begin(ctx, count, count * psz, out, flags, &bp));
    const int sl = bp.nl;  // shared arena: the evaluations when they come from the host
    int rc = ensure_lanes(ctx, sl + 1);
    if (rc == KZG_OK) rc = lane_reserve(ctx, sl, stage_bytes(d * 32, flags) + 4096);
    const void *de = nullptr;
    if (rc == KZG_OK) rc = stage_in(ctx, sl, evals, d * 32, flags, &de);
    if (rc == KZG_OK) rc = eval_tables_ready(ctx, sl, log_d);  // also waits for the stage-in
    size_t per = msm_workspace_bytes(lagrange, d) + d * 32 + (d / 256 + 4) * 32 + 65536;
    for (int l = 0; l < bp.nl && rc == KZG_OK; l++) rc = lane_reserve(ctx, l, per);
    for (size_t j = 0; j < count && rc == KZG_OK; j++) {
        int l = (int)(j % bp.nl);
        ctx->lanes[l].arena_used = 0;
        Fr *dq = (Fr *)lane_alloc(ctx, l, d * 32);
        if (!dq) rc = fail(ctx, KZG_ERR_ALLOC, "workspace");
        if (rc == KZG_OK) rc = quotient_eval_run(ctx, l, (const Fr *)de, log_d, indices[j], sfmt, dq);
        MsmPoint *res = nullptr;
        if (rc == KZG_OK) rc = batch_msm(ctx, bp, j, l, lagrange, 0, dq, d, sfmt, &res);
        if (rc == KZG_OK) rc = emit_point(ctx, l, res, bp.d_out + j * psz, ofmt);
    }
    return batch_end(ctx, bp, rc, out, count * psz);
}

extern "C" int kzg_g1_sum(kzg_ctx *ctx, const void *points, size_t count, int pfmt, int flags, void *out, int ofmt) {
    if (!ctx || !out || (!points && count)) return KZG_ERR_SHAPE;
    Guard g(ctx);
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    size_t psz = point_format_bytes(pfmt);
    if (!psz || !point_format_bytes(ofmt)) return fail(ctx, KZG_ERR_SHAPE, "unknown G1 point format");
    size_t cnt = count ? count : 1;
    size_t need = stage_bytes(cnt * psz, flags) + cnt * sizeof(G1Xyzz) + (cnt + 2 * sum_points_scratch_count(cnt) + 4) * sizeof(MsmPoint) + 8192;
    KZG_TRY(lane_reserve(ctx, 0, need));
    hipStream_t st = ctx->lanes[0].stream;
    G1Xyzz *dec = (G1Xyzz *)lane_alloc(ctx, 0, cnt * sizeof(G1Xyzz));
    MsmPoint *pts = (MsmPoint *)lane_alloc(ctx, 0, cnt * sizeof(MsmPoint));
    MsmPoint *scratch = (MsmPoint *)lane_alloc(ctx, 0, 2 * sum_points_scratch_count(cnt) * sizeof(MsmPoint));
    int *bad = (int *)lane_alloc(ctx, 0, 256);
    if (!dec || !pts || !scratch || !bad) return fail(ctx, KZG_ERR_ALLOC, "sum workspace not reserved");
    KZG_HIP_CHECK(ctx, hipMemsetAsync(bad, 0, sizeof(int), st));
    MsmPoint *res = pts;
    if (count == 0) {
        KZG_TRY(point_set_infinity(ctx, st, pts));
    } else {
        const void *d_raw = nullptr;
        KZG_TRY(stage_in(ctx, 0, points, count * psz, flags, &d_raw));
        KZG_TRY(decode_points(ctx, st, d_raw, count, pfmt, dec, bad, POINTS_ON_CURVE));
        KZG_TRY(points_to30(ctx, st, dec, pts, count));
        KZG_TRY(sum_points_run(ctx, 0, pts, count, scratch, &res));
    }
    int hbad = 0;
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, st));
    KZG_TRY(finish_point(ctx, 0, res, out, ofmt, flags));
    if (hbad) return fail(ctx, KZG_ERR_BAD_POINT, "a G1 point failed to decode or is not on the curve");
    return KZG_OK;
}

namespace kzg {
int g1_sum_batch_strided(kzg_ctx *ctx, const void *points, size_t count, size_t groups, size_t gstride, size_t istride, int pfmt,
                         int flags, void *out, int ofmt, int level) {
    Guard g(ctx);
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    size_t psz = point_format_bytes(pfmt), osz = point_format_bytes(ofmt);
    if (!psz || !osz) return fail(ctx, KZG_ERR_SHAPE, "unknown G1 point format");
    if (count > (1u << 20) || groups > (1u << 24)) return fail(ctx, KZG_ERR_SHAPE, "kzg_g1_sum_batch: count <= 2^20, groups <= 2^24");
    // every source slot the strides reach: count * groups of them for dense groups, more when the records carry other slots
    // between the points -- the gathered [world][batch + 1] records of a device group (mgpu.hip) hold a status slot after each
    // rank's batch partials, so rank r's partial of polynomial b sits at r (batch + 1) + b
    size_t total = count && groups ? (groups - 1) * gstride + (count - 1) * istride + 1 : 0;
    KZG_TRY(lane_reserve(ctx, 0, stage_bytes(total * psz, flags) + total * sizeof(G1Xyzz) + (total + groups + 4) * sizeof(MsmPoint) + groups * 144 + 65536));
    hipStream_t st = ctx->lanes[0].stream;
    G1Xyzz *dec = (G1Xyzz *)lane_alloc(ctx, 0, total * sizeof(G1Xyzz));
    MsmPoint *pts = (MsmPoint *)lane_alloc(ctx, 0, total * sizeof(MsmPoint));
    MsmPoint *tmp = (MsmPoint *)lane_alloc(ctx, 0, groups * sizeof(MsmPoint));
    int *bad = (int *)lane_alloc(ctx, 0, 256);
    void *d_out = (flags & KZG_OUT_DEVICE) ? out : lane_alloc(ctx, 0, groups * osz);
    if (!dec || !pts || !tmp || !bad || !d_out) return fail(ctx, KZG_ERR_ALLOC, "sum workspace not reserved");
    KZG_HIP_CHECK(ctx, hipMemsetAsync(bad, 0, sizeof(int), st));
    const void *d_raw = nullptr;
    KZG_TRY(stage_in(ctx, 0, points, total * psz, flags, &d_raw));
    KZG_TRY(decode_points(ctx, st, d_raw, total, pfmt, dec, bad, level));
    KZG_TRY(points_to30(ctx, st, dec, pts, total));
    // up to four host-bound sums (the combine step of a sharded commit has one): converted on the calling thread, as in finish_point
    const bool on_host = !(flags & KZG_OUT_DEVICE) && ctx->opt_host_affine && groups <= 4;
    KZG_TRY(sum_groups_emit(ctx, 0, pts, count, groups, gstride, istride, tmp, on_host ? nullptr : d_out, ofmt));
    int hbad = 0;
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, st));
    MsmPoint hsum[4];
    if (on_host) KZG_HIP_CHECK(ctx, hipMemcpyAsync(hsum, tmp, groups * sizeof(MsmPoint), hipMemcpyDeviceToHost, st));
    else if (!(flags & KZG_OUT_DEVICE)) KZG_HIP_CHECK(ctx, hipMemcpyAsync(out, d_out, groups * osz, hipMemcpyDeviceToHost, st));
    KZG_HIP_CHECK(ctx, hipStreamSynchronize(st));
    if (on_host)
        for (size_t gi = 0; gi < groups; gi++) {
            alignas(16) uint8_t buf[144];
            emit_one(hsum[gi], buf, ofmt);
            memcpy((uint8_t *)out + gi * osz, buf, osz);
        }
    if (ctx->prof) prof_collect(ctx);
    if (hbad) return fail(ctx, KZG_ERR_BAD_POINT, "a G1 point failed to decode or is not on the curve");
    return KZG_OK;
}
}  // namespace kzg

extern "C" int kzg_g1_sum_batch(kzg_ctx *ctx, const void *points, size_t count, size_t groups, int pfmt, int flags,
                                void *out, int ofmt) {
    if (!ctx || !out || !points || count == 0 || groups == 0) return KZG_ERR_SHAPE;
    return g1_sum_batch_strided(ctx, points, count, groups, count, 1, pfmt, flags, out, ofmt);
}

// ---------------------------------------------------------------------------------------------
// NTT
// ---------------------------------------------------------------------------------------------
extern "C" int kzg_compute_omega(size_t d, size_t *m_out, uint32_t *exp_out, void *omega, int sfmt) {
    // EvaluationDomain::compute_omega (src/ft.rs:55-76)
    size_t m = 1;
    uint32_t exp = 0;
    while (m < d) {
        m *= 2;
        exp += 1;
        if (exp >= FR_TWO_ADICITY) return KZG_ERR_DEGREE_TOO_LARGE;
    }
    if (m_out) *m_out = m;
    if (exp_out) *exp_out = exp;
    if (omega) scalar_out(host_omega(exp), sfmt, omega);
    return KZG_OK;
}

extern "C" int kzg_ntt_fr(kzg_ctx *ctx, void *data, uint32_t log_n, int inverse, int flags) {
    if (!ctx || !data) return KZG_ERR_SHAPE;
    Lease ls;  // EvaluationDomain::fft is a `&mut self` method of the caller's own vector: many threads, one context
    KZG_TRY(lease_lane(ctx, &ls));
    const int lane = ls.lane;
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (log_n >= FR_TWO_ADICITY) return fail(ctx, KZG_ERR_DEGREE_TOO_LARGE, "polynomial degree too large");
    if (log_n > 28) return fail(ctx, KZG_ERR_SHAPE, "NTT sizes above 2^28 are not supported (2^24-point two-pass transforms under one 16-point outer level)");
    size_t n = (size_t)1 << log_n;
    // above 2^24: the transposed copy of the whole vector + the inner transform's 2^24-element scratch + the outer twiddle tables
    KZG_TRY(lane_reserve(ctx, lane, ntt_workspace_bytes(log_n) + stage_bytes(n * 32, flags) + 8192));
    hipStream_t st = ctx->lanes[lane].stream;
    const void *d = nullptr;
    KZG_TRY(stage_in(ctx, lane, data, n * 32, flags, &d));
    KZG_TRY(ntt_run(ctx, lane, (Fr *)d, log_n, inverse));
    if (!(flags & KZG_IN_DEVICE)) KZG_HIP_CHECK(ctx, hipMemcpyAsync(data, d, n * 32, hipMemcpyDeviceToHost, st));
    KZG_HIP_CHECK(ctx, hipStreamSynchronize(st));
    if (ctx->prof) prof_collect(ctx);
    return KZG_OK;
}

// ---------------------------------------------------------------------------------------------
// Fr polynomial helpers
// ---------------------------------------------------------------------------------------------
extern "C" int kzg_poly_eval(kzg_ctx *ctx, const void *coeffs, size_t n, const void *x, int sfmt, int flags, void *y_out) {
    if (!ctx || !coeffs || !x || !y_out || n == 0 || !count_ok(n)) return KZG_ERR_SHAPE;
    Lease ls;
    KZG_TRY(lease_lane(ctx, &ls));
    const int lane = ls.lane;
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    KZG_TRY(check_sfmt(ctx, sfmt));
    Fr xm;
    KZG_TRY(host_scalar(ctx, x, sfmt, &xm));
    KZG_TRY(lane_reserve(ctx, lane, stage_bytes(n * 32, flags) + (n / 2048 + 4) * 64 + 65536));
    hipStream_t st = ctx->lanes[lane].stream;
    const void *d = nullptr;
    KZG_TRY(stage_in(ctx, lane, coeffs, n * 32, flags, &d));
    Fr *dy = (Fr *)lane_alloc(ctx, lane, 256);
    if (!dy) return fail(ctx, KZG_ERR_ALLOC, "workspace");
    KZG_TRY(poly_eval_run(ctx, lane, (const Fr *)d, n, xm, dy));
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(y_out, dy, 32, hipMemcpyDeviceToHost, st));
    KZG_HIP_CHECK(ctx, hipStreamSynchronize(st));
    if (ctx->prof) prof_collect(ctx);
    return KZG_OK;
}

extern "C" int kzg_quotient_linear(kzg_ctx *ctx, const void *coeffs, size_t n, const void *x, const void *y, int sfmt,
                                   int flags, void *q_out) {
    if (!ctx || !coeffs || !x || !y || n == 0 || (!q_out && n > 1) || !count_ok(n)) return KZG_ERR_SHAPE;
    Lease ls;
    KZG_TRY(lease_lane(ctx, &ls));
    const int lane = ls.lane;
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    KZG_TRY(check_sfmt(ctx, sfmt));
    Fr xm;
    KZG_TRY(host_scalar(ctx, x, sfmt, &xm));
    KZG_TRY(lane_reserve(ctx, lane, stage_bytes(n * 32, flags) * 2 + (n / 2048 + 4) * 64 + 65536));
    hipStream_t st = ctx->lanes[lane].stream;
    const void *d = nullptr;
    KZG_TRY(stage_in(ctx, lane, coeffs, n * 32, flags, &d));
    bool out_dev = (flags & KZG_OUT_DEVICE) != 0;
    Fr *dq = out_dev ? (Fr *)q_out : (Fr *)lane_alloc(ctx, lane, n * 32);
    Fr *dpx = (Fr *)lane_alloc(ctx, lane, 256);
    if (!dq || !dpx) return fail(ctx, KZG_ERR_ALLOC, "workspace");
    KZG_TRY(quotient_linear_run(ctx, lane, (const Fr *)d, n, xm, dq, dpx));
    Fr px;
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(px.v, dpx, 32, hipMemcpyDeviceToHost, st));
    if (!out_dev && n > 1) KZG_HIP_CHECK(ctx, hipMemcpyAsync(q_out, dq, (n - 1) * 32, hipMemcpyDeviceToHost, st));
    KZG_HIP_CHECK(ctx, hipStreamSynchronize(st));
    if (ctx->prof) prof_collect(ctx);
    if (memcmp(px.v, y, 32) != 0) return fail(ctx, KZG_ERR_POINT_NOT_ON_POLY, "point not on polynomial!");
    return KZG_OK;
}

extern "C" int kzg_quotient_eval(kzg_ctx *ctx, const void *evals, size_t d, size_t i, int sfmt, int flags, void *q_out) {
    if (!ctx || !evals || !q_out) return KZG_ERR_SHAPE;
    Lease ls;
    KZG_TRY(lease_lane(ctx, &ls));
    const int lane = ls.lane;
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    KZG_TRY(check_sfmt(ctx, sfmt));
    if (!is_pow2(d)) return fail(ctx, KZG_ERR_SHAPE, "evaluation domain size must be a power of two");
    if (i >= d) return fail(ctx, KZG_ERR_SHAPE, "evaluation index out of range (reference: index panic)");
    uint32_t log_d = (uint32_t)ilog2_ceil(d);
    if (log_d >= FR_TWO_ADICITY) return fail(ctx, KZG_ERR_DEGREE_TOO_LARGE, "domain too large");
    KZG_TRY(lane_reserve(ctx, lane, stage_bytes(d * 32, flags) * 2 + (d / 256 + 4) * 32 + 65536));
    hipStream_t st = ctx->lanes[lane].stream;
    const void *de = nullptr;
    KZG_TRY(stage_in(ctx, lane, evals, d * 32, flags, &de));
    bool out_dev = (flags & KZG_OUT_DEVICE) != 0;
    Fr *dq = out_dev ? (Fr *)q_out : (Fr *)lane_alloc(ctx, lane, d * 32);
    if (!dq) return fail(ctx, KZG_ERR_ALLOC, "workspace");
    KZG_TRY(quotient_eval_run(ctx, lane, (const Fr *)de, log_d, i, sfmt, dq));
    if (!out_dev) KZG_HIP_CHECK(ctx, hipMemcpyAsync(q_out, dq, d * 32, hipMemcpyDeviceToHost, st));
    KZG_HIP_CHECK(ctx, hipStreamSynchronize(st));
    if (ctx->prof) prof_collect(ctx);
    return KZG_OK;
}

// ---------------------------------------------------------------------------------------------
// witnesses
// ---------------------------------------------------------------------------------------------
extern "C" int kzg_witness_coeff(kzg_ctx *ctx, const kzg_srs *srs, const void *coeffs, size_t n, const void *x,
                                 const void *y, int sfmt, int flags, void *out, int ofmt) {
    // KZGProver::create_witness (src/coeff_form.rs:66-81)
    if (!ctx || !srs || !coeffs || !x || !y || !out || n == 0) return KZG_ERR_SHAPE;
    Lease ls;
    KZG_TRY(lease_lane(ctx, &ls));
    const int lane = ls.lane;
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    KZG_TRY(check_sfmt(ctx, sfmt));
    if (n - 1 > srs->n) return fail(ctx, KZG_ERR_SHAPE, "quotient longer than the SRS (reference: slice index panic)");
    Fr xm;
    KZG_TRY(host_scalar(ctx, x, sfmt, &xm));
    size_t need = msm_workspace_bytes(srs, n - 1) + stage_bytes(n * 32, flags) + n * 32 + (n / 2048 + 4) * 64 + 65536;
    KZG_TRY(lane_reserve(ctx, lane, need));
    hipStream_t st = ctx->lanes[lane].stream;
    const void *d = nullptr;
    KZG_TRY(stage_in(ctx, lane, coeffs, n * 32, flags, &d));
    Fr *dq = (Fr *)lane_alloc(ctx, lane, n * 32);
    Fr *dpx = (Fr *)lane_alloc(ctx, lane, 256);
    if (!dq || !dpx) return fail(ctx, KZG_ERR_ALLOC, "workspace");
    KZG_TRY(quotient_linear_run(ctx, lane, (const Fr *)d, n, xm, dq, dpx));
    // p(x) goes to the lane's PINNED staging buffer (behind the 144 bytes finish_point uses): a device-to-host copy into pageable
    // memory -- the stack variable this used to be -- makes hipMemcpyAsync wait for the stream, so the host sat out the three quotient
    // kernels before it could enqueue the MSM's fourteen, and the GPU then idled between those short kernels (0.4-0.5 ms of a lone
    // create_witness; profiles/r06_prof_witness_coeff.txt)
    KZG_TRY(lane_pinned(ctx, lane, 4096));
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(ctx->lanes[lane].pinned + 1024, dpx, 32, hipMemcpyDeviceToHost, st));
    MsmPoint *res = nullptr;
    KZG_TRY(lease_msm(ctx, ls, srs, 0, dq, n - 1, sfmt, &res));
    KZG_TRY(finish_point(ctx, lane, res, out, ofmt, flags));  // synchronises the stream
    Fr px;
    memcpy(px.v, ctx->lanes[lane].pinned + 1024, 32);
    // remainder of (p - y)/(X - x) is p(x) - y: Some(_) => Err(PointNotOnPolynomial)
    if (memcmp(px.v, y, 32) != 0) return fail(ctx, KZG_ERR_POINT_NOT_ON_POLY, "point not on polynomial!");
    return KZG_OK;
}

extern "C" int kzg_witness_eval(kzg_ctx *ctx, const kzg_srs *lagrange, const void *evals, size_t d, size_t i, int sfmt,
                                int flags, void *out, int ofmt) {
    // KZGProverEvalForm::create_witness (src/eval_form.rs:124-140)
    if (!ctx || !lagrange || !evals || !out) return KZG_ERR_SHAPE;
    Lease ls;
    KZG_TRY(lease_lane(ctx, &ls));
    const int lane = ls.lane;
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    KZG_TRY(check_sfmt(ctx, sfmt));
    if (!is_pow2(d)) return fail(ctx, KZG_ERR_SHAPE, "evaluation domain size must be a power of two");
    if (i >= d) return fail(ctx, KZG_ERR_SHAPE, "evaluation index out of range (reference: index panic)");
    if (d > lagrange->n) return fail(ctx, KZG_ERR_SHAPE, "evaluations longer than the Lagrange SRS (reference: slice panic)");
    uint32_t log_d = (uint32_t)ilog2_ceil(d);
    if (log_d >= FR_TWO_ADICITY) return fail(ctx, KZG_ERR_DEGREE_TOO_LARGE, "domain too large");
    size_t need = msm_workspace_bytes(lagrange, d) + stage_bytes(d * 32, flags) + d * 32 + (d / 256 + 4) * 32 + 65536;
    KZG_TRY(lane_reserve(ctx, lane, need));
    const void *de = nullptr;
    KZG_TRY(stage_in(ctx, lane, evals, d * 32, flags, &de));
    Fr *dq = (Fr *)lane_alloc(ctx, lane, d * 32);
    if (!dq) return fail(ctx, KZG_ERR_ALLOC, "workspace");
    KZG_TRY(quotient_eval_run(ctx, lane, (const Fr *)de, log_d, i, sfmt, dq));
    MsmPoint *res = nullptr;
    KZG_TRY(lease_msm(ctx, ls, lagrange, 0, dq, d, sfmt, &res));
    return finish_point(ctx, lane, res, out, ofmt, flags);
}

static int verify_against(kzg_ctx *ctx, int lane, const MsmPoint *res, const void *commitment, int pfmt, int *ok) {
    size_t psz = point_format_bytes(pfmt);
    if (!psz || pfmt == KZG_G1_JACOBIAN_MONT_144)
        return fail(ctx, KZG_ERR_SHAPE, "verify_poly takes the commitment in an affine format (KZGCommitment = G1Affine)");
    uint8_t mine[96];
    KZG_TRY(finish_point(ctx, lane, res, mine, pfmt, 0));
    *ok = memcmp(mine, commitment, psz) == 0;
    return KZG_OK;
}

extern "C" int kzg_verify_poly_coeff(kzg_ctx *ctx, const kzg_srs *srs, const void *commitment, int pfmt,
                                     const void *coeffs, size_t n, int sfmt, int flags, int *ok) {
    // KZGVerifier::verify_poly (src/coeff_form.rs:119-124)
    if (!ctx || !srs || !commitment || !ok || (!coeffs && n)) return KZG_ERR_SHAPE;
    Lease ls;
    KZG_TRY(lease_lane(ctx, &ls));
    const int lane = ls.lane;
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    KZG_TRY(check_sfmt(ctx, sfmt));
    if (n > srs->n) return fail(ctx, KZG_ERR_SHAPE, "polynomial longer than the SRS (reference: slice index panic)");
    KZG_TRY(lane_reserve(ctx, lane, msm_workspace_bytes(srs, n) + stage_bytes(n * 32, flags) + 8192));
    const void *d = nullptr;
    KZG_TRY(stage_in(ctx, lane, coeffs, n * 32, flags, &d));
    MsmPoint *res = nullptr;
    KZG_TRY(lease_msm(ctx, ls, srs, 0, d, n, sfmt, &res));
    return verify_against(ctx, lane, res, commitment, pfmt, ok);
}

extern "C" int kzg_verify_poly_eval(kzg_ctx *ctx, const kzg_srs *monomial, const void *commitment, int pfmt,
                                    const void *evals, size_t d, int sfmt, int flags, int *ok) {
    // KZGVerifierEvalForm::verify_poly (src/eval_form.rs:162-171): ifft, then the monomial-basis MSM
    if (!ctx || !monomial || !commitment || !ok || !evals) return KZG_ERR_SHAPE;
    Lease ls;
    KZG_TRY(lease_lane(ctx, &ls));
    const int lane = ls.lane;
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    KZG_TRY(check_sfmt(ctx, sfmt));
    if (!is_pow2(d)) return fail(ctx, KZG_ERR_SHAPE, "evaluation domain size must be a power of two");
    if (d > monomial->n) return fail(ctx, KZG_ERR_SHAPE, "polynomial longer than the SRS (reference: slice index panic)");
    uint32_t log_d = (uint32_t)ilog2_ceil(d);
    if (log_d >= FR_TWO_ADICITY) return fail(ctx, KZG_ERR_DEGREE_TOO_LARGE, "domain too large");
    KZG_TRY(lane_reserve(ctx, lane, msm_workspace_bytes(monomial, d) + d * 32 + ntt_workspace_bytes(log_d) + 65536));
    hipStream_t st = ctx->lanes[lane].stream;
    Fr *work = (Fr *)lane_alloc(ctx, lane, d * 32);
    if (!work) return fail(ctx, KZG_ERR_ALLOC, "workspace");
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(work, evals, d * 32, (flags & KZG_IN_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    KZG_TRY(ntt_run(ctx, lane, work, log_d, 1));
    MsmPoint *res = nullptr;
    KZG_TRY(lease_msm(ctx, ls, monomial, 0, work, d, sfmt, &res));
    return verify_against(ctx, lane, res, commitment, pfmt, ok);
}

#ifdef KZG_TEST_HOOKS
#include "../../include/kzg_mi355x_test.h"
// ---------------------------------------------------------------------------------------------
// test hooks
// ---------------------------------------------------------------------------------------------
template <class K, class... Args>
static int run_test_kernel(kzg_ctx *ctx, const char *name, K kern, size_t n, size_t in_elem, int n_in, const void *a,
                           const void *b, size_t in_elem_b, void *out, size_t out_elem) {
    Guard g(ctx);
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    KZG_TRY(lane_reserve(ctx, 0, n * (in_elem + in_elem_b + out_elem) + 65536));
    hipStream_t st = ctx->lanes[0].stream;
    void *da = lane_alloc(ctx, 0, n * in_elem + 16), *db = lane_alloc(ctx, 0, n * in_elem_b + 16),
         *dout = lane_alloc(ctx, 0, n * out_elem + 16);
    if (!da || !db || !dout) return fail(ctx, KZG_ERR_ALLOC, "workspace");
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(da, a, n * in_elem, hipMemcpyHostToDevice, st));
    if (n_in > 1) KZG_HIP_CHECK(ctx, hipMemcpyAsync(db, b, n * in_elem_b, hipMemcpyHostToDevice, st));
    kern(st, da, db, dout);
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(out, dout, n * out_elem, hipMemcpyDeviceToHost, st));
    KZG_HIP_CHECK(ctx, hipStreamSynchronize(st));
    KZG_HIP_CHECK(ctx, hipGetLastError());
    return KZG_OK;
}

extern "C" int kzg_test_fr_mul(kzg_ctx *ctx, const void *a, const void *b, size_t n, void *out) {
    if (!ctx || !n) return KZG_ERR_SHAPE;
    unsigned grid = (unsigned)((n + 255) / 256);
    return run_test_kernel(ctx, "k_test_fr_mul", [&](hipStream_t st, void *da, void *db, void *dout) {
        hipLaunchKernelGGL(k_test_fr_mul, dim3(grid), dim3(256), 0, st, (const Fr *)da, (const Fr *)db, n, (Fr *)dout);
    }, n, 32, 2, a, b, 32, out, 32);
}
extern "C" int kzg_test_fq_mul(kzg_ctx *ctx, const void *a, const void *b, size_t n, void *out) {
    if (!ctx || !n) return KZG_ERR_SHAPE;
    unsigned grid = (unsigned)((n + 255) / 256);
    return run_test_kernel(ctx, "k_test_fq_mul", [&](hipStream_t st, void *da, void *db, void *dout) {
        hipLaunchKernelGGL(k_test_fq_mul, dim3(grid), dim3(256), 0, st, (const Fq *)da, (const Fq *)db, n, (Fq *)dout);
    }, n, 48, 2, a, b, 48, out, 48);
}
extern "C" int kzg_test_fr_inv(kzg_ctx *ctx, const void *a, size_t n, void *out) {
    if (!ctx || !n) return KZG_ERR_SHAPE;
    unsigned grid = (unsigned)((n + 255) / 256);
    return run_test_kernel(ctx, "k_test_fr_inv", [&](hipStream_t st, void *da, void *, void *dout) {
        hipLaunchKernelGGL(k_test_fr_inv, dim3(grid), dim3(256), 0, st, (const Fr *)da, n, (Fr *)dout);
    }, n, 32, 1, a, nullptr, 0, out, 32);
}
extern "C" int kzg_test_g1_add(kzg_ctx *ctx, const void *a, const void *b, size_t n, void *out) {
    if (!ctx || !n) return KZG_ERR_SHAPE;
    unsigned grid = (unsigned)((n + 255) / 256);
    return run_test_kernel(ctx, "k_test_g1_add", [&](hipStream_t st, void *da, void *db, void *dout) {
        hipLaunchKernelGGL(k_test_g1_add, dim3(grid), dim3(256), 0, st, (const G1Affine *)da, (const G1Affine *)db, n,
                           (G1Affine *)dout);
    }, n, 96, 2, a, b, 96, out, 96);
}
// pretend an SRS lives on another GPU (one-GPU test boxes): the device check of msm_run
extern "C" int kzg_test_srs_set_device(kzg_srs *srs, int device) {
    if (!srs) return KZG_ERR_SHAPE;
    srs->device = device;
    return KZG_OK;
}
extern "C" int kzg_test_g1_mul(kzg_ctx *ctx, const void *p, const void *k, size_t n, void *out) {
    if (!ctx || !n) return KZG_ERR_SHAPE;
    unsigned grid = (unsigned)((n + 255) / 256);
    return run_test_kernel(ctx, "k_test_g1_mul", [&](hipStream_t st, void *da, void *db, void *dout) {
        hipLaunchKernelGGL(k_test_g1_mul, dim3(grid), dim3(256), 0, st, (const G1Affine *)da, (const Fr *)db, n,
                           (G1Affine *)dout);
    }, n, 96, 2, p, k, 32, out, 96);
}
#endif  // KZG_TEST_HOOKS

