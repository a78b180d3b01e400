// common.h -- context, workspace arena, error handling and profiled kernel launches shared by the
// translation units of libkzg_mi355x.so.
#pragma once
#ifndef KZG_NTT_KERNEL_DEFAULT
#define KZG_NTT_KERNEL_DEFAULT 1
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <array>
#include <atomic>
#include <condition_variable>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/kzg_mi355x.h"
#include "curve.h"
#include "curve30.h"

namespace kzg {

struct ProfEntry {
    uint64_t launches = 0;
    double total_ms = 0.0;
};

struct PendingEvent {
    std::string name;
    hipEvent_t start, stop;
};

#ifndef KZG_ACCUM_WAVES
#define KZG_ACCUM_WAVES 2  // waves per SIMD k_accum_affine is compiled for (VGPR budget 512 / waves)
#endif

// How one MSM is shaped for what else is on the GPU (set per lane by the entry point that submits it).
struct MsmMode {
    int accum_blocks = 256 * KZG_ACCUM_WAVES;  // k_accum_affine grid (a lone MSM: every SIMD holds its KZG_ACCUM_WAVES waves)
    int sort_threads = 1024;                   // threads per k_hist / k_scatter block
    bool tail_quads = true;   // four lanes per point operation in the depth-bound tail kernels (latency mode, msm_tail.hip)
    bool tail_wide = false;   // a pipeline whose lanes never run dry: work-efficient fold width and row/column sums
};

// One independent execution lane: a HIP stream plus a bump-allocated scratch arena in HBM.
// Exclusive calls use lane 0; kzg_msm_g1_batch pipelines independent MSMs over several lanes (two in flight per lane); concurrent
// blocking callers (commit / create_witness / create_witness_batched / fft ... from many host threads) each lease one lane for the
// duration of their call.
struct Lane {
    MsmMode mode;
    hipStream_t stream = nullptr;
    char *arena = nullptr;
    size_t arena_bytes = 0;
    size_t arena_used = 0;
    size_t arena_limit = 0;  // 0 = the whole arena; else lane_alloc fails beyond this offset (two MSMs in flight share the arena as halves)
    uint32_t heavy_seq = 0, heavy_seen = 0;  // MSMs sorted on this lane / the one last looked at (kzg_ctx::d_lane_heavy)
    char *pinned = nullptr;  // small pinned host staging buffer for results
    size_t pinned_bytes = 0;
};

struct NttPlan;
struct FixedBaseTable;
struct EvalDomainTables;

// Who may use the context when.  Two kinds of callers:
//   exclusive  everything that works on all lanes at once or rebuilds shared state (the *_batch / *_many entry points, SRS
//              construction, the pairing verifier, options, profiling, kzg_sync / kzg_dev_free): one at a time.  lock() / unlock() --
//              the type is a BasicLockable, so std::lock_guard<CtxGate> is the old "Guard".
//   shared     every blocking call of the reference that works on ONE lane (KZGProver::commit / create_witness /
//              create_witness_batched, both forms, verify_poly, EvaluationDomain::fft and friends: `&self` methods of Clone types,
//              src/coeff_form.rs:59-124, src/eval_form.rs:114-171, src/ft.rs:111-271).  Each leases a free lane under this short lock
//              (kzg::Lease), submits its kernels there (accumulation kernels on the shared FIFO streams) and waits for its own
//              stream only -- N host threads keep the pipeline as full as kzg_msm_g1_batch does.  The caches such calls share (NTT
//              plans, coset tables, evaluation-domain tables) are guarded by kzg_ctx::cache_mu.
// Exclusive callers wait for the leased lanes to drain and hold new leases off while they wait (no starvation).
struct CtxGate {
    std::mutex m;
    std::condition_variable cv;
    int shared_active = 0;
    int excl_waiting = 0;
    bool exclusive = false;
    uint32_t lane_busy = 0;  // bit l: lane l is leased
    void lock() {
        std::unique_lock<std::mutex> lk(m);
        excl_waiting++;
        cv.wait(lk, [&] { return !exclusive && shared_active == 0; });
        excl_waiting--;
        exclusive = true;
    }
    void unlock() {
        {
            std::lock_guard<std::mutex> lk(m);
            exclusive = false;
        }
        cv.notify_all();
    }
};
constexpr int KZG_HEAVY_WINDOW = 64;  // MSMs for which one oversized sort bin keeps the slice kernels enqueued (kzg_ctx::heavy_last)
constexpr int KZG_MAX_LANES = 24;  // lanes.reserve(): leased lanes are indexed while other threads append

}  // namespace kzg

struct kzg_ctx;
namespace kzg {
struct StreamPool;                     // runtime.hip: the process' streams of one device, shared by its contexts
struct PointSetCache;
void point_sets_free(kzg_ctx *ctx);  // witness.hip
void point_set_stats(kzg_ctx *ctx, uint64_t *hits, uint64_t *misses);
}
struct kzg_ctx {
    int device = 0;
    kzg::CtxGate mu;
    std::mutex err_mu;   // err
    std::mutex prof_mu;  // prof_map, prof_pending, event_pool
    std::mutex cache_mu; // ntt_plans, coset_tabs, eval_tabs: built by one leased caller at a time, published after the builder's stream is synchronised
    std::mutex accum_mu; // wait / launch / record on a shared accumulation stream is one unit
    std::atomic<uint32_t> accum_rr{0};  // round robin over the accumulation streams (leased lanes)
    // the last plan of the batched pipeline (plan_pipeline): what was asked for, what the hardware-queue pool of the process allowed.
    // narrowed (plan_lanes < plan_want_lanes or plan_accum < plan_want_accum) = other streams of the process -- a second context, an
    // RCCL communicator, the host's own -- hold queues: reported by kzg_ctx_info / kzg_mctx_info and once on stderr.
    int plan_want_lanes = 0, plan_want_accum = 0, plan_lanes = 0, plan_accum = 0, plan_queues = 0;
    bool plan_warned = false;
    bool pipe_planned = false;          // lanes + accumulation streams created, probed and ordered for concurrent callers
    int pipe_lanes = 1, pipe_accum = 0;
    std::string err;
    std::vector<kzg::Lane> lanes;
    kzg::StreamPool *pool = nullptr;   // where the lanes' and accumulation streams come from (runtime.hip)
    int opt_window_bits = 0;  // 0 = auto
    int opt_streams = 13;  // lanes: depth of the batched pipeline and of the pool concurrent blocking callers lease from (13 + 4
                           // accumulation streams leave an RCCL communicator its ~6 hardware queues of the 24: runtime.hip, StreamPool)
    int opt_accum_blocks = 0;          // k_accum_affine grid for a single MSM (0 = every SIMD holds its KZG_ACCUM_WAVES waves)
    int opt_hw_queues = 0;             // hardware queues to plan the batched pipeline for (0 = measure: probe_queues in runtime.hip)
    int probed_queues = 0;             // hardware queues the probed streams were found on (0 = not measured yet)
    int probed_lanes = 0, probed_accum = 0;  // which streams that measurement covered: lanes [0, probed_lanes), accumulation streams
    std::vector<hipStream_t> probed_order;  // the probed streams, one per queue first
    int opt_window_rows = 0;           // 0 = every window has its table row; r > 0: keep r rows (low-memory SRS, multi-pass MSM)
    int opt_naf_window = 0;            // SRSs created afterwards: 18 = positional tables + width-18 NAF digits (17x the table; measured
                                       // +3.7 % batched throughput at 2^20, nothing below 2^19, +1.3 ms on a lone commit: off by default)
    int opt_trusted_points = 0;        // 1: caller vouches for its points (skip the subgroup check of uploads / verifier inputs)
    int opt_ntt_vec_log = 2;           // NTT passes: 2^v adjacent columns / rows per LDS tile
    int opt_ntt_kernel = KZG_NTT_KERNEL_DEFAULT;  // 0: three-phase passes (k_ntt_pass1/2); 1: load/store fused into the first/last stage pair (k_ntt_tile); 2: 1 + two butterflies per thread
    int opt_ntt_three_from = 23;       // NTT sizes from 2^this on take three passes of <= 2^8 points (ntt_run3); 0 = never
    int opt_ntt_vec2_log = 1;          // pass 2: at most 2^v rows per tile (rows are contiguous: narrow tiles cost no coalescing on the load side)
    int opt_ntt_xcd = 1;               // XCD-aware tile order: bit 0 = pass 1, bit 1 = pass 2 (ntt.hip, xcd_tile)
    int opt_accum_blocks_batch = 0;    // batched MSMs (0 = auto): leave 1/16 of the wave slots to the latency-bound tail and sort
                                       // kernels of the neighbouring MSMs in flight (measured +6 % throughput)
    int opt_tail_quads = 1;            // single MSMs: four lanes per point operation in the tail kernels (latency mode, msm_tail.hip)
    // kzg_msm_g1_batch: every k_accum_affine runs on one of this many dedicated streams, in submission order (0 = on its lane's
    // stream).  With the accumulation on the lanes' own streams the lanes fall into a convoy -- all sorting, then up to nine
    // accumulation kernels resident at once, then all in their tails -- and no accumulation kernel is resident 6 % of the time
    // (rocprofv3 kernel trace); two FIFO streams keep exactly the next one or two queued behind the running one: +3.5 %.
    // Needs its own hardware queues (GPU_MAX_HW_QUEUES >= lanes + 2): sharing a queue with a lane serialises them (-10 %).
    int opt_accum_streams = 2;
    // Small MSMs in a pipeline (sorted entries W * n <= opt_small_entries: up to 2^17 points): the per-thread costs of the equal
    // split (a partial sum written and folded per thread and per bucket boundary) weigh more the shorter a thread's chunk, so such
    // an MSM gets a smaller accumulation grid and the pipeline more accumulation streams to keep the chip full -- measured
    // (profiles/r04_small_grid_ab.txt) 160 blocks on 4 streams against 480 on 2: +49 % at 2^14, +39 % at 2^15, +20 % at 2^16, +9 % at
    // 2^17; at 2^20 four streams measure -1 %, so larger MSMs keep two.
    int opt_accum_streams_small = 4;
    int opt_accum_blocks_small = 160;
    int64_t opt_small_entries = 2 << 20;
    int planned_accum = 0;   // accumulation streams of the current plan (plan_pipeline): with fewer than three (a process short of
                             // hardware queues narrows the pipeline) the small grid would leave most of the chip idle, so the rule is off
    bool msm_small(size_t entries) const { return opt_accum_blocks_small > 0 && planned_accum >= 3 && (int64_t)entries <= opt_small_entries; }
    hipStream_t accum_streams[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> sorted_events, accum_events;  // per lane
    int opt_sort_threads = 1024;       // threads per k_hist / k_scatter block for a single MSM (one block per CU: 128 KiB of LDS)
    int opt_sort_threads_batch = 1024; // ... for batched MSMs
    int opt_host_affine = 1;           // a lone result bound for host memory is converted to affine / serialised on the host (emit.h)
    int opt_heavy_bins = 0;            // level 2 of the two-level sorts: 0 = slices for oversized bins once such bins have been seen (below), 1 = always, 2 = never
    bool opt_no_defer_tail = false;    // option defer_tail = 0: kzg_msm_g1_batch enqueues every MSM's tail right behind its accumulation (A/B)
    int opt_sort_single = 0;           // 1: c = 17 sorts in one pass (2^16 cursors, two walks) instead of the two-level sort
    int accum_blocks_single() const { return opt_accum_blocks ? opt_accum_blocks : 256 * KZG_ACCUM_WAVES; }
    int accum_blocks_batch() const { return opt_accum_blocks_batch ? opt_accum_blocks_batch : 240 * KZG_ACCUM_WAVES; }
    int num_cus = 256;
    std::atomic<bool> attr_msm_set{false}, attr_ntt_set{false}, attr_wide_set{false}, attr_sort20_set{false}, attr_horner_set{false};  // > 64 KiB dynamic-LDS opt-in done for this device
    // profiling
    bool prof = false;
    bool prof_only_accum = false;  // events around k_accum_affine only (bench.py's timed region: two events per kernel launch cost 1.4 % there)
    std::map<std::string, kzg::ProfEntry> prof_map;
    std::vector<kzg::PendingEvent> prof_pending;
    std::vector<hipEvent_t> event_pool;
    // caches
    std::map<uint32_t, kzg::NttPlan *> ntt_plans;          // key = log_n * 2 + inverse
    std::map<uint32_t, kzg::EvalDomainTables *> eval_tabs;  // key = log_d
    kzg::FixedBaseTable *fixed_base = nullptr;
    // distribute_powers (coset NTTs): per coset generator g two device tables g^j and g^(1024 j), j < 1024 (witness.hip)
    std::vector<std::pair<std::array<uint32_t, 8>, void *>> coset_tabs;
    // create_witness_batched: what depends on the opening POINTS only (Z, 1 / Z'(x_i), the coset shift, 1 / Z on the coset), kept
    // per point set for callers that open many polynomials at the same points (witness.hip, PointSetCache).  One pool of equal slots,
    // allocated by the first call that can use it; option "witness_cache_slots" (default 16, 0 = off; read before the pool exists).
    kzg::PointSetCache *point_sets = nullptr;
    int opt_witness_cache_slots = 16;
    // Oversized sort bins (msm_wide.hip).  A level-2 block that finds its bin oversized writes the MSM's sequence number (per lane,
    // Lane::heavy_seq) into the lane's word of d_lane_heavy; the host picks the words up with the results (finish_point,
    // batch_end) and notes the context's MSM count at that moment in heavy_last.  The three slice kernels are only enqueued for
    // the next KZG_HEAVY_WINDOW MSMs after such a pick-up (or option heavy_bins = 1): enqueued for nothing they cost uniform
    // scalars 0.8 % of the batched rate.  Setting the option clears the history.
    std::atomic<uint64_t> msm_count{0}, heavy_last{0};
    uint32_t *d_lane_heavy = nullptr;
    void *batch_out = nullptr;  // device staging of kzg_msm_g1_batch results (grow-only)
    size_t batch_out_bytes = 0;
};

struct kzg_srs {
    size_t n = 0;        // points
    size_t npad = 0;     // row stride (points)
    int c = 0;           // window bits (signed digits, 2^(c-1) buckets)
    int W = 0;           // windows = ceil(256 / c) (15 in the c = 17 single-pass mode); table row w holds 2^(c*w) * P_i
    int rows = 0;        // table rows resident (= W unless option window_rows asked for fewer: then an MSM takes ceil(W / rows) passes)
    bool narrow17 = false;  // c = 17: 15 windows, balanced scalars; two-level sort (or, option sort_single_pass, one pass walking the scalars twice)
    bool sort20 = false;    // c = 20: 13 windows, 2^19 buckets, two-level sort (msm_wide.hip); option sort_single_pass: the older two-pass path
    // Positional tables (naf = 18): the table holds 2^j P_i for EVERY bit position j (255 rows) instead of every c-th, and a
    // scalar is recoded in width-18 non-adjacent form: odd digits |d| < 2^17 at arbitrary positions, at least 18 apart -- 13.9
    // non-zero digits per 255-bit scalar on average instead of the 15 of fixed 17-bit windows (-7.4 % bucket additions) into
    // the same 2^16 buckets (bucket (|d| - 1) / 2, weight 2 b + 1).  c = 17 / W = 15 then describe the bucket count and the
    // maximum digits per scalar; rows = 255.  Costs 17x the table (32.6 KiB per point: 34 GB at 2^20), and the random gathers
    // over 34 GB cost 10 % more per entry than over 2 GB (profiles/r03_naf_tables.txt): opt-in (option naf_window = 18).
    int naf = 0;
    int row_shift = 0;   // row w holds 2^(row_shift w) P: c for window tables, 1 for positional tables
    kzg::G1Affine *table = nullptr;  // [W][npad], affine Montgomery
    void *table30 = nullptr;         // [W][npad] G1Affine30 (2 x 13 x 30-bit limbs + pad, KZG_ROW_BYTES = 128 B): what k_accum_affine gathers
    int device = 0;
};

namespace kzg {

struct Guard {  // exclusive use of the context for the scope
    std::lock_guard<CtxGate> lk;
    explicit Guard(kzg_ctx *c) : lk(c->mu) {}
};

// Shared use of the context for the scope: one leased lane (stream + arena).  Every blocking prover / verifier / transform call
// that works on ONE lane takes a Lease instead of a Guard (runtime.hip: lease_lane), so N host threads inside create_witness_batched,
// commit, fft, verify_poly ... run side by side on one context, as the reference's `&self` methods do (src/coeff_form.rs:59-111).
struct Lease {
    kzg_ctx *ctx = nullptr;
    int lane = -1;
    bool pipelined = false;  // others were in flight when the lane was leased: the call's accumulation goes to a FIFO accumulation stream
    uint32_t slot = 0;       // ... which one: slot modulo the number of streams MSMs of its size are spread over (lease_msm)
    ~Lease();
};
int lease_lane(kzg_ctx *ctx, Lease *ls);

// validation of decoded points (what blstrs' G1Affine / G2Affine deserialisation enforces upstream):
//   POINTS_TRUSTED   nothing (the engine's own intermediate results)
//   POINTS_ON_CURVE  limbs canonical (< q) and on the curve
//   POINTS_SUBGROUP  + in the r-torsion subgroup ([r]P == O)
enum { POINTS_TRUSTED = 0, POINTS_ON_CURVE = 1, POINTS_SUBGROUP = 2 };

#define KZG_HIP_CHECK(ctx, expr)                                                                   \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return kzg::fail((ctx), KZG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

#define KZG_TRY(expr)                                                                              \
    do {                                                                                           \
        int _s = (expr);                                                                           \
        if (_s != KZG_OK) return _s;                                                               \
    } while (0)

// The message goes to the context (for callers that ask later from another thread) and to the calling thread's own slot:
// with several threads inside one context kzg_last_error returns the failure of the caller's own last call.
struct ThreadErr {
    const kzg_ctx *ctx = nullptr;
    std::string msg;
};
inline ThreadErr &thread_err() {
    static thread_local ThreadErr te;
    return te;
}
inline int fail(kzg_ctx *ctx, int code, const std::string &msg) {
    {
        std::lock_guard<std::mutex> lk(ctx->err_mu);
        ctx->err = msg;
    }
    ThreadErr &te = thread_err();
    te.ctx = ctx;
    te.msg = msg;
    return code;
}

// ---- arena ---------------------------------------------------------------------------------
int lane_reserve(kzg_ctx *ctx, int lane, size_t bytes);  // ensure arena >= bytes and reset it
void *lane_alloc(kzg_ctx *ctx, int lane, size_t bytes);  // 256-B aligned bump allocation (nullptr if exhausted)
int lane_pinned(kzg_ctx *ctx, int lane, size_t bytes);   // ensure pinned staging >= bytes
// runtime.hip: the pipeline plan of a context (lanes + accumulation streams on hardware queues of their own) and the shape of a lane
int ensure_lanes(kzg_ctx *ctx, int want);                                     // exclusive callers only
int plan_pipeline(kzg_ctx *ctx, int want, int *nl_out, int *nas_out);
void set_lane_mode(kzg_ctx *ctx, int lane, bool pipelined, bool deep);
int accum_streams_for(kzg_ctx *ctx, int planned, const kzg_srs *srs, size_t n);  // accumulation streams MSMs of this size are spread over

// ---- profiling -----------------------------------------------------------------------------
struct ProfScope {
    kzg_ctx *ctx;
    hipStream_t stream;
    hipEvent_t start = nullptr, stop = nullptr;
    const char *name;
    ProfScope(kzg_ctx *c, hipStream_t s, const char *n);
    ~ProfScope();
};
void prof_collect(kzg_ctx *ctx);

#define KZG_LAUNCH(ctx, stream, name, kern, grid, block, shmem, ...)                               \
    do {                                                                                           \
        kzg::ProfScope _ps((ctx), (stream), (name));                                               \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(block), (shmem), (stream), __VA_ARGS__);         \
    } while (0)

inline int untrusted_level(const kzg_ctx *ctx) { return ctx->opt_trusted_points ? POINTS_ON_CURVE : POINTS_SUBGROUP; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline int ilog2_ceil(size_t x) {
    int l = 0;
    while (((size_t)1 << l) < x) l++;
    return l;
}

// The point type the MSM pipeline computes in and hands between its kernels: extended Jacobian coordinates in
// the signed 30-bit field representation (curve30.h).  Canonical encodings are produced by emit_point.
typedef G1Xyzz30 MsmPoint;

// ---- cross-TU entry points -------------------------------------------------------------------
// msm.hip
size_t msm_workspace_bytes(const kzg_srs *srs, size_t n);
// one MSM result (device) -> `ofmt` at `out` (host, or device with KZG_OUT_DEVICE); synchronises the lane's stream (capi.hip)
int finish_point(kzg_ctx *ctx, int lane, const MsmPoint *d_pt, void *out, int ofmt, int flags);
// d_scalars: device pointer to n scalars (sfmt); result: one device MsmPoint in the lane arena
// accum_stream (optional): run k_accum_affine there instead of on the lane's stream, ordered by the two caller-owned events
// `defer` (batched pipeline): when the accumulation kernel went to a FIFO stream, the part behind it (wait for it, fold, bucket
// reduction) is NOT enqueued: it is described in *defer and enqueued later by msm_finish -- after the lane's NEXT sort, so that the
// accumulation queue is fed before the lane's latency-bound tail kernels run (two MSMs in flight per lane, two workspaces).
struct MsmPending {
    bool active = false;
    hipStream_t st = nullptr;
    hipEvent_t accum_ev = nullptr;
    MsmMode mm;
    const MsmPoint *part = nullptr;
    MsmPoint *scratch = nullptr, *result = nullptr;
    const uint32_t *s1 = nullptr;
    int B = 0;
    size_t expected = 0;
    void *state = nullptr;
    char *tail_base = nullptr;
    size_t tail_off[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // TailLayout (msm_internal.h), opaque here
    bool odd = false;
};
int msm_run(kzg_ctx *ctx, int lane, const kzg_srs *srs, size_t offset, const void *d_scalars, size_t n, int sfmt,
            MsmPoint **d_result, hipStream_t accum_stream = nullptr, hipEvent_t sorted_ev = nullptr, hipEvent_t accum_ev = nullptr,
            MsmPending *defer = nullptr);
int msm_finish(kzg_ctx *ctx, MsmPending &pd, MsmPoint **d_result);  // the deferred part of an MSM (its result if it was not deferred)
// runtime.hip: one MSM on a leased lane (its accumulation kernel on the lease's FIFO stream when other calls are in flight)
int lease_msm(kzg_ctx *ctx, const Lease &ls, const kzg_srs *srs, size_t offset, const void *d_sc, size_t n, int sfmt, MsmPoint **res);
// d_points: count points -> one point (plain sum)
int sum_points_run(kzg_ctx *ctx, int lane, MsmPoint *d_points, size_t count, MsmPoint *d_scratch, MsmPoint **d_result);
// conversions between the canonical saturated XYZZ form and MsmPoint (device arrays)
int points_to30(kzg_ctx *ctx, hipStream_t st, const G1Xyzz *d_in, MsmPoint *d_out, size_t n);
int points_from30(kzg_ctx *ctx, hipStream_t st, const MsmPoint *d_in, G1Xyzz *d_out, size_t n);
int point_set_infinity(kzg_ctx *ctx, hipStream_t st, MsmPoint *d_pt);
size_t sum_points_scratch_count(size_t count);
// writes one point in `ofmt` (from XYZZ) to d_out (device); single thread incl. the Fq inversion
int emit_point(kzg_ctx *ctx, int lane, const MsmPoint *d_point, void *d_out, int ofmt);
size_t point_format_bytes(int fmt);
// out[g] = sum_i pts[g * gstride + i * istride], written in ofmt
int sum_groups_emit(kzg_ctx *ctx, int lane, const MsmPoint *d_pts, size_t count, size_t groups, size_t gstride, size_t istride,
                    MsmPoint *d_tmp, void *d_out, int ofmt);

// capi.hip
int msm_batch_strided(kzg_ctx *ctx, const kzg_srs *srs, size_t offset, const void *scalars, size_t n, size_t batch,
                      size_t stride_bytes, int sfmt, int flags, void *out, int ofmt);
// out[g] = sum_i points[g * gstride + i * istride] (strides in points); points in pfmt, host or device per flags
int g1_sum_batch_strided(kzg_ctx *ctx, const void *points, size_t count, size_t groups, size_t gstride, size_t istride, int pfmt,
                         int flags, void *out, int ofmt, int level = POINTS_ON_CURVE);

// srs.hip
int srs_choose_window(int opt_window_bits, size_t n);
void srs_shape(int opt_window_bits, int opt_window_rows, size_t n, int *c, int *W, int *rows, bool *narrow17, int opt_naf = 0, int *naf = nullptr);
// scalars' digits one MSM pass turns into sorted entries, at most
inline size_t srs_entries_per_scalar(const kzg_srs *s) { return s->naf ? (size_t)s->W : (size_t)s->rows; }
int srs_alloc(kzg_ctx *ctx, size_t n, kzg_srs **out);
int srs_finish_from_xyzz(kzg_ctx *ctx, kzg_srs *srs, G1Xyzz *d_row0_xyzz);  // batch-affine row 0 then precompute rows
int srs_precompute(kzg_ctx *ctx, kzg_srs *srs);                              // rows 1..W-1 from row 0
int batch_to_affine(kzg_ctx *ctx, hipStream_t stream, const G1Xyzz *d_in, G1Affine *d_out, size_t n);
int decode_points(kzg_ctx *ctx, hipStream_t st, const void *d_raw, size_t n, int fmt, G1Xyzz *d_out, int *d_bad,
                  int level = POINTS_SUBGROUP);

int fixed_base_mul(kzg_ctx *ctx, hipStream_t stream, const Fr *d_scalars_mont, size_t n, G1Xyzz *d_out);
int powers_run(kzg_ctx *ctx, hipStream_t st, const Fr &base_mont, size_t first, size_t n, Fr *d_out);  // base^(first+i)
int lagrange_scalars_run(kzg_ctx *ctx, hipStream_t st, const Fr &tau_mont, size_t d, Fr *d_out);      // L_i(tau)

// witness.hip
struct WitnessSink {
    const kzg_srs *srs;    // the SRS (or SRS shard) the quotient MSM runs against
    size_t first, len;     // it holds the points of the quotient coefficients [first, first + len)
    size_t total;          // length of the whole SRS (the reference's slice bound)
    void *d_partial;       // nullptr: witness to the caller; else the 144-byte Jacobian partial goes here (device)
};
int witness_coeff_batched_run(kzg_ctx *ctx, const WitnessSink &sink, const void *coeffs, size_t n, const void *xs, const void *ys,
                              size_t k, int sfmt, int flags, void *out_w, int ofmt, void *out_r, size_t *out_r_len);
int vanishing_poly_run(kzg_ctx *ctx, hipStream_t st, const Fr *d_xs_mont, size_t k, Fr *d_z, Fr *d_tmp);  // k+1 coeffs each

// cached per-context tables, released by kzg_ctx_destroy
void ntt_plans_free(kzg_ctx *ctx);   // ntt.hip
void eval_tabs_free(kzg_ctx *ctx);   // poly.hip
void fixed_base_free(kzg_ctx *ctx);  // srs.hip

// ntt.hip
int ntt_run(kzg_ctx *ctx, int lane, Fr *d_data, uint32_t log_n, int inverse, size_t nnz = (size_t)-1);  // nnz: d_data[nnz ..) is zero (assumed, not read)
// arena bytes ONE ntt_run of 2^log_n points takes from its lane (never returned before the call ends): the two-pass scratch; above
// 2^24 the transposed copy, the inner transform's scratch and the outer twiddle tables
inline size_t ntt_workspace_bytes(uint32_t log_n) {
    const size_t n = (size_t)1 << log_n;
    return log_n > 24 ? n * 32 + ((size_t)1 << 24) * 32 + (2 << 20) : n * 32 + 4096;
}
bool ntt_short_input_ok(uint32_t log_n, size_t nnz);  // ntt_run(..., nnz) will not read d_data[nnz ..)
int pow_table(kzg_ctx *ctx, hipStream_t stream, const Fr &base_mont, const Fr &scale_mont, size_t count, Fr *d_out);
Fr host_omega(uint32_t exp);  // Montgomery-form 2^exp-th root of unity per compute_omega (src/ft.rs:73)

// poly.hip
int fr_convert(kzg_ctx *ctx, hipStream_t stream, Fr *d_data, size_t n, int to_mont);  // in place
int batch_inverse(kzg_ctx *ctx, hipStream_t stream, const Fr *d_in, Fr *d_out, size_t n);
// d_a[i] *= 1 / d_c[i] (zeros of c: a[i] = 0 and *d_flag |= 1); d_tmp: n elements of scratch
int batch_inverse_mul(kzg_ctx *ctx, hipStream_t stream, const Fr *d_c, Fr *d_a, Fr *d_tmp, size_t n, int *d_flag);
// Horner machinery: eval and linear quotient. d_coeffs in Montgomery or canonical form (linear ops:
// the result is in the same form provided x_mont is Montgomery).
int poly_eval_run(kzg_ctx *ctx, int lane, const Fr *d_coeffs, size_t n, const Fr &x_mont, Fr *d_y_out);
int eval_tables_ready(kzg_ctx *ctx, int lane, uint32_t log_d);
int quotient_linear_run(kzg_ctx *ctx, int lane, const Fr *d_coeffs, size_t n, const Fr &x_mont, Fr *d_q_out,
                        Fr *d_px_out);
int quotient_eval_run(kzg_ctx *ctx, int lane, const Fr *d_evals, uint32_t log_d, size_t m, int sfmt, Fr *d_q_out);

}  // namespace kzg
