// capi.hip -- the extern "C" boundary declared in include/kzg_mi355x.h: context / lanes / profiling
// and the entry points that mirror KZGProver / KZGProverEvalForm / EvaluationDomain method by method.
#include <dlfcn.h>

#include <algorithm>

#include "common.h"
#include "emit.h"

namespace kzg {

// ---------------------------------------------------------------------------------------------
// lanes
// ---------------------------------------------------------------------------------------------
int lane_reserve(kzg_ctx *ctx, int lane, size_t bytes) {
    Lane &l = ctx->lanes[lane];
    bytes = align_up(bytes + 4096, 1 << 20);
    if (l.arena_bytes < bytes) {
        KZG_HIP_CHECK(ctx, hipStreamSynchronize(l.stream));
        if (l.arena) KZG_HIP_CHECK(ctx, hipFree(l.arena));
        l.arena = nullptr;
        l.arena_bytes = 0;
        hipError_t e = hipMalloc((void **)&l.arena, bytes);
        if (e != hipSuccess) return fail(ctx, KZG_ERR_ALLOC, std::string("hipMalloc(workspace): ") + hipGetErrorString(e));
        l.arena_bytes = bytes;
    }
    l.arena_used = 0;
    return KZG_OK;
}

void *lane_alloc(kzg_ctx *ctx, int lane, size_t bytes) {
    Lane &l = ctx->lanes[lane];
    size_t off = align_up(l.arena_used, 256);
    if (off + bytes > (l.arena_limit ? std::min(l.arena_limit, l.arena_bytes) : l.arena_bytes)) return nullptr;
    l.arena_used = off + bytes;
    return l.arena + off;
}

int lane_pinned(kzg_ctx *ctx, int lane, size_t bytes) {
    Lane &l = ctx->lanes[lane];
    if (l.pinned_bytes >= bytes) return KZG_OK;
    if (l.pinned) hipHostFree(l.pinned);
    l.pinned = nullptr;
    l.pinned_bytes = 0;
    bytes = align_up(bytes, 4096);
    KZG_HIP_CHECK(ctx, hipHostMalloc((void **)&l.pinned, bytes, hipHostMallocDefault));
    l.pinned_bytes = bytes;
    return KZG_OK;
}

// ---------------------------------------------------------------------------------------------
// profiling: HIP events recorded on the stream each kernel is launched on
// ---------------------------------------------------------------------------------------------
static hipEvent_t get_event(kzg_ctx *ctx) {  // prof_mu held
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
}

ProfScope::ProfScope(kzg_ctx *c, hipStream_t s, const char *n) : ctx(c), stream(s), name(n) {
    if (!ctx->prof) return;
    if (ctx->prof_only_accum && strcmp(n, "k_accum_affine") != 0) return;  // kzg_prof_enable(ctx, 2): the dominant kernel only
    {
        std::lock_guard<std::mutex> lk(ctx->prof_mu);
        start = get_event(ctx);
        stop = get_event(ctx);
    }
    hipEventRecord(start, stream);
}

ProfScope::~ProfScope() {
    if (!start) return;
    hipEventRecord(stop, stream);
    std::lock_guard<std::mutex> lk(ctx->prof_mu);
    ctx->prof_pending.push_back(PendingEvent{name, start, stop});
}

void prof_collect(kzg_ctx *ctx) {
    // the pending list is taken under the lock and waited for outside it: with concurrent leased callers a thread collecting
    // must not hold every other thread's ProfScope (and their kernels still in flight) behind prof_mu
    std::vector<PendingEvent> pend;
    {
        std::lock_guard<std::mutex> lk(ctx->prof_mu);
        pend.swap(ctx->prof_pending);
    }
    std::vector<float> ms(pend.size(), -1.f);
    for (size_t i = 0; i < pend.size(); i++) {
        hipEventSynchronize(pend[i].stop);
        float t = 0.f;
        if (hipEventElapsedTime(&t, pend[i].start, pend[i].stop) == hipSuccess) ms[i] = t;
    }
    std::lock_guard<std::mutex> lk(ctx->prof_mu);
    for (size_t i = 0; i < pend.size(); i++) {
        if (ms[i] >= 0.f) {
            ProfEntry &e = ctx->prof_map[pend[i].name];
            e.launches++;
            e.total_ms += ms[i];
        }
        ctx->event_pool.push_back(pend[i].start);
        ctx->event_pool.push_back(pend[i].stop);
    }
}

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
// context
// ---------------------------------------------------------------------------------------------
extern "C" const char *kzg_version(void) { return "kzg-mi355x 0.1.0 (gfx950)"; }

// The pipelined paths want one hardware queue per stream (14 lanes + 4 accumulation streams, and ~6 for an RCCL communicator); the HIP runtime sizes its queue
// pool from GPU_MAX_HW_QUEUES (default 4) when it initialises, i.e. at the first HIP call of the process.  The library does not
// touch the host's environment on its own: the host either exports GPU_MAX_HW_QUEUES itself, or calls kzg_init_hw_queues()
// before its first HIP call, or sets KZG_SET_HW_QUEUES=<n> to let the load-time constructor below do it.  Without any of these
// the pipeline measures the queues it has (probe_queues) and narrows itself (4 queues: 3 lanes + 1 accumulation stream; loss
// in INTEGRATION.md section 6, profiles/r03_hw_queues.txt).
extern "C" int kzg_init_hw_queues(int queues) {
    if (queues < 0 || queues > 64) return KZG_ERR_SHAPE;
    char buf[16];
    snprintf(buf, sizeof buf, "%d", queues ? queues : 24);
    return setenv("GPU_MAX_HW_QUEUES", buf, 0) == 0 ? KZG_OK : KZG_ERR_INTERNAL;  // a value the host exported is kept
}
__attribute__((constructor)) static void kzg_optional_hw_queues() {
    const char *e = getenv("KZG_SET_HW_QUEUES");
    if (e && atoi(e) > 0) kzg_init_hw_queues(atoi(e) == 1 ? 0 : atoi(e));
}

// Which HIP runtime is this library bound to?  A process may hold two (PyTorch wheels ship their own libamdhip64 next to
// /opt/rocm's); the dynamic loader binds this library to whichever copy with the matching SONAME was loaded first, so the answer
// depends on the host's import order.  bench.py puts the string into its result line.
extern "C" int kzg_runtime_info(char *buf, size_t buflen) {
    if (!buf || !buflen) return KZG_ERR_SHAPE;
    Dl_info di;
    const char *file = "?";
    if (dladdr((const void *)&hipStreamSynchronize, &di) && di.dli_fname) file = di.dli_fname;
    int rt = 0, drv = 0;
    hipRuntimeGetVersion(&rt);
    hipDriverGetVersion(&drv);
    snprintf(buf, buflen, "hip=%s runtime_version=%d driver_version=%d", file, rt, drv);
    return KZG_OK;
}

// "device=<d> lanes=<n> accum_streams=<m> hw_queues_found=<q> narrowed_from=<L>+<A>|none witness_cache_slots=<s>": the batched
// pipeline's current plan (all zero before the first batched / concurrent call) and whether the process' hardware-queue pool forced
// it below what was asked for.
extern "C" int kzg_ctx_info(kzg_ctx *ctx, char *buf, size_t buflen) {
    if (!ctx || !buf || !buflen) return KZG_ERR_SHAPE;
    Guard g(ctx);
    char nf[32] = "none";
    if (ctx->plan_lanes < ctx->plan_want_lanes || ctx->plan_accum < ctx->plan_want_accum)
        snprintf(nf, sizeof nf, "%d+%d", ctx->plan_want_lanes, ctx->plan_want_accum);
    snprintf(buf, buflen, "device=%d lanes=%d accum_streams=%d hw_queues_found=%d narrowed_from=%s witness_cache_slots=%d", ctx->device,
             ctx->plan_lanes, ctx->plan_accum, ctx->plan_queues, nf, ctx->opt_witness_cache_slots);
    return KZG_OK;
}

extern "C" int kzg_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 0) return 0;
    return count;
}

// ---- one pool of streams per device and PROCESS ----------------------------------------------------------------------------
// Every context used to create its own lanes (16, or 14 in a device group) + 4 accumulation streams, and the runtime multiplexes all streams of a process onto
// ONE pool of hardware queues (GPU_MAX_HW_QUEUES).  A second context -- a device group's beside a plain prover's, what INTEGRATION.md
// section 5b describes -- then found most of its streams sharing queues with the first one's, narrowed its pipeline to what was
// left and lost 19 % of its batched rate (384.8 against 475.6 commitments/s; profiles/r05_engine_and_group.txt).  Streams are only
// ordered queues: contexts of one device now take THE SAME streams from this pool (lane i of every context is pool lane i), so a
// process holds 18 streams however many contexts it has, each on a queue of its own.  Work of two contexts interleaves on a
// stream in submission order; every wait is on an event that the same host thread submitted EARLIER in real time, so the streams'
// FIFO order cannot close a cycle.  An RCCL communicator needs about six queues of the same pool (24 by default): with 16 lanes + 4
// accumulation streams beside one the exchange's kernels queue behind the pipeline's (336.6 against 469.9 commitments/s), which is
// why every context plans 14 + 4 (option "streams"; same-box 474.2 against 471.2 commitments/s for a plain context: no loss) -- a
// prover context, a device group and its communicator then fit the pool together: the group path beside a live plain context
// 471.1 against 473.5 alone, both committing at once 512-517 in total (profiles/r05_engine_and_group.txt).  Not isolated: a
// collective that never leaves the group's exchange stream (lane 0; after a failed ncclCommAbort) blocks that pool stream for the
// device's other contexts too -- by then the process has lost its RCCL anyway (mgpu.hip, rccl_mark_wedged).
// The pool's streams live as long as some context of the device does: when the last one is destroyed they are returned (a process that
// has used many streams costs OTHER processes on the GPU dearly even when idle: a child process measured 43 instead of 460
// commitments/s beside a parent that had run one batch and closed its engine, 317 beside one that had only created a context --
// and hipDeviceReset in the parent does not give the queues back; bench.py therefore runs its child BEFORE it touches the GPU).
namespace kzg {
struct StreamPool {
    std::mutex mu;
    std::vector<hipStream_t> lanes;
    hipStream_t accum[4] = {nullptr, nullptr, nullptr, nullptr};
    // the last queue measurement over pool streams: class (= hardware queue) of each measured stream
    std::map<hipStream_t, int> cls;
    int refs = 0;  // contexts of this device alive in the process
};
static std::mutex g_pools_mu;
static std::map<int, StreamPool *> g_pools;
static StreamPool *pool_for(int device) {
    std::lock_guard<std::mutex> lk(g_pools_mu);
    auto it = g_pools.find(device);
    if (it != g_pools.end()) return it->second;
    StreamPool *p = new StreamPool();
    g_pools[device] = p;
    return p;
}
static void pool_ref(StreamPool *p) {
    std::lock_guard<std::mutex> lk(p->mu);
    p->refs++;
}
static void pool_unref(StreamPool *p) {  // the caller has synchronised its streams and set the device
    std::lock_guard<std::mutex> lk(p->mu);
    if (--p->refs > 0) return;
    for (auto st : p->lanes) hipStreamDestroy(st);
    for (auto &st : p->accum)
        if (st) {
            hipStreamDestroy(st);
            st = nullptr;
        }
    p->lanes.clear();
    p->cls.clear();
}
static hipError_t pool_lane(StreamPool *p, int i, hipStream_t *out) {
    std::lock_guard<std::mutex> lk(p->mu);
    while ((int)p->lanes.size() <= i) {
        hipStream_t st = nullptr;
        hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (e != hipSuccess) return e;
        p->lanes.push_back(st);
    }
    *out = p->lanes[i];
    return hipSuccess;
}
static hipError_t pool_accum(StreamPool *p, int i, hipStream_t *out) {
    std::lock_guard<std::mutex> lk(p->mu);
    if (!p->accum[i]) {
        hipError_t e = hipStreamCreateWithFlags(&p->accum[i], hipStreamNonBlocking);
        if (e != hipSuccess) return e;
    }
    *out = p->accum[i];
    return hipSuccess;
}
}  // namespace kzg

extern "C" int kzg_ctx_create(int device, kzg_ctx **out) {
    if (!out) return KZG_ERR_SHAPE;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return KZG_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return KZG_ERR_NO_DEVICE;
    kzg_ctx *ctx = new kzg_ctx();
    ctx->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cus = prop.multiProcessorCount;
    ctx->lanes.reserve(KZG_MAX_LANES);  // never reallocated: leased lanes are indexed while an exclusive caller appends
    ctx->lanes.resize(1);
    ctx->pool = pool_for(device);
    pool_ref(ctx->pool);
    if (pool_lane(ctx->pool, 0, &ctx->lanes[0].stream) != hipSuccess) {
        pool_unref(ctx->pool);
        delete ctx;
        return KZG_ERR_HIP;
    }
    if (hipMalloc((void **)&ctx->d_lane_heavy, KZG_MAX_LANES * 4) != hipSuccess ||
        hipMemset(ctx->d_lane_heavy, 0, KZG_MAX_LANES * 4) != hipSuccess) {
        if (ctx->d_lane_heavy) hipFree(ctx->d_lane_heavy);
        pool_unref(ctx->pool);
        delete ctx;
        return KZG_ERR_ALLOC;
    }
    *out = ctx;
    return KZG_OK;
}

static int ensure_lanes(kzg_ctx *ctx, int want) {  // exclusive callers only
    if (want > KZG_MAX_LANES) return fail(ctx, KZG_ERR_INTERNAL, "lane count");
    while ((int)ctx->lanes.size() < want) {
        Lane l;
        KZG_HIP_CHECK(ctx, pool_lane(ctx->pool, (int)ctx->lanes.size(), &l.stream));  // (this context's lane i = the pool's lane i until a probe re-orders them)
        ctx->lanes.push_back(l);
    }
    return KZG_OK;
}

extern "C" void kzg_ctx_destroy(kzg_ctx *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    for (auto &l : ctx->lanes) {
        if (l.stream) hipStreamSynchronize(l.stream);
        if (l.arena) hipFree(l.arena);
        if (l.pinned) hipHostFree(l.pinned);
    }
    // (the streams belong to the process' pool)
    for (auto e : ctx->event_pool) hipEventDestroy(e);
    for (auto e : ctx->sorted_events) hipEventDestroy(e);
    for (auto e : ctx->accum_events) hipEventDestroy(e);
    if (ctx->batch_out) hipFree(ctx->batch_out);
    if (ctx->d_lane_heavy) hipFree(ctx->d_lane_heavy);
    for (auto &ct : ctx->coset_tabs)
        if (ct.second) hipFree(ct.second);
    ntt_plans_free(ctx);
    eval_tabs_free(ctx);
    fixed_base_free(ctx);
    point_sets_free(ctx);
    for (auto st : ctx->accum_streams)
        if (st) hipStreamSynchronize(st);
    pool_unref(ctx->pool);  // the last context of the device returns the pool's streams
    delete ctx;
}

extern "C" const char *kzg_last_error(kzg_ctx *ctx) {
    // copied under the context's lock into a per-thread buffer: another thread failing on the same context cannot
    // invalidate the returned pointer (it stays valid until this thread's next kzg_last_error call)
    // the calling thread's own last failure on this context if it had one, else the context's last message
    if (!ctx) return "null context";
    static thread_local std::string tl_err;
    ThreadErr &te = thread_err();
    if (te.ctx == ctx) {
        tl_err = te.msg;
    } else {
        std::lock_guard<std::mutex> lk(ctx->err_mu);
        tl_err = ctx->err;
    }
    return tl_err.c_str();
}

extern "C" int kzg_sync(kzg_ctx *ctx) {
    if (!ctx) return KZG_ERR_SHAPE;
    Guard g(ctx);
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    for (auto &l : ctx->lanes) KZG_HIP_CHECK(ctx, hipStreamSynchronize(l.stream));
    return KZG_OK;
}

extern "C" int kzg_ctx_set_option(kzg_ctx *ctx, const char *key, int64_t value) {
    if (!ctx || !key) return KZG_ERR_SHAPE;
    Guard g(ctx);
    std::string k(key);
    if (k == "window_bits") {
        if (value != 0 && (value < 4 || value > 20)) return fail(ctx, KZG_ERR_SHAPE, "window_bits must be 0 or 4..20");
        ctx->opt_window_bits = (int)value;
    } else if (k == "streams") {
        if (value < 1 || value > 16) return fail(ctx, KZG_ERR_SHAPE, "streams must be 1..16");
        ctx->opt_streams = (int)value;
        ctx->pipe_planned = false;
    } else if (k == "accum_blocks" || k == "accum_blocks_batch") {
        if (value != 0 && (value < 64 || value > 256 * KZG_ACCUM_WAVES))
            return fail(ctx, KZG_ERR_SHAPE, "accum_blocks must be 0 (auto) or 64..256 x waves per SIMD");
        (k == "accum_blocks" ? ctx->opt_accum_blocks : ctx->opt_accum_blocks_batch) = (int)value;
    } else if (k == "scan_threads" || k == "scan_threads_batch") {
        // accepted for compatibility: the bucket scans are multi-block kernels now (k_scan_a / k_scan_b)
    } else if (k == "sort_threads" || k == "sort_threads_batch") {
        if (value != 256 && value != 512 && value != 1024) return fail(ctx, KZG_ERR_SHAPE, "sort_threads must be 256, 512 or 1024");
        (k == "sort_threads" ? ctx->opt_sort_threads : ctx->opt_sort_threads_batch) = (int)value;
    } else if (k == "accum_streams") {
        if (value < 0 || value > 4) return fail(ctx, KZG_ERR_SHAPE, "accum_streams must be 0..4");
        ctx->opt_accum_streams = (int)value;
        ctx->pipe_planned = false;
    } else if (k == "accum_streams_small") {
        if (value < 0 || value > 4) return fail(ctx, KZG_ERR_SHAPE, "accum_streams_small must be 0..4");
        ctx->opt_accum_streams_small = (int)value;
        ctx->pipe_planned = false;
    } else if (k == "accum_blocks_small") {
        if (value != 0 && (value < 64 || value > 256 * KZG_ACCUM_WAVES)) return fail(ctx, KZG_ERR_SHAPE, "accum_blocks_small must be 0 (off) or 64..256 x waves per SIMD");
        ctx->opt_accum_blocks_small = (int)value;
        ctx->pipe_planned = false;
    } else if (k == "small_entries") {
        if (value < 0) return fail(ctx, KZG_ERR_SHAPE, "small_entries must be >= 0");
        ctx->opt_small_entries = value;
    } else if (k == "host_affine") {
        ctx->opt_host_affine = value != 0;
    } else if (k == "heavy_bins") {
        if (value < 0 || value > 2) return fail(ctx, KZG_ERR_SHAPE, "heavy_bins: 0 (adaptive), 1 (always slice oversized sort bins), 2 (never)");
        ctx->opt_heavy_bins = (int)value;
        ctx->heavy_last.store(0, std::memory_order_relaxed);
    } else if (k == "sort_single_pass") {
        ctx->opt_sort_single = value != 0;
    } else if (k == "defer_tail") {
        ctx->opt_no_defer_tail = value == 0;
    } else if (k == "tail_quads") {
        ctx->opt_tail_quads = value != 0;
    } else if (k == "hw_queues") {
        if (value < 0 || value > 64) return fail(ctx, KZG_ERR_SHAPE, "hw_queues must be 0 (GPU_MAX_HW_QUEUES or the ROCm default of 4) or 1..64");
        ctx->opt_hw_queues = (int)value;
        ctx->pipe_planned = false;
    } else if (k == "window_rows") {
        if (value < 0 || value > 64) return fail(ctx, KZG_ERR_SHAPE, "window_rows must be 0 (one table row per window) or 1..64");
        ctx->opt_window_rows = (int)value;
    } else if (k == "naf_window") {
        if (value != 0 && value != 18) return fail(ctx, KZG_ERR_SHAPE, "naf_window must be 0 (window tables) or 18 (positional tables, width-18 NAF digits)");
        ctx->opt_naf_window = (int)value;
    } else if (k == "trusted_points") {
        ctx->opt_trusted_points = value != 0;
    } else if (k == "witness_cache_slots") {
        if (value < 0 || value > 256) return fail(ctx, KZG_ERR_SHAPE, "witness_cache_slots must be 0..256");
        if (ctx->point_sets) return fail(ctx, KZG_ERR_SHAPE, "witness_cache_slots: the cache already exists (set the option before the first create_witness_batched)");
        ctx->opt_witness_cache_slots = (int)value;
    } else if (k == "ntt_xcd") {
        if (value < 0 || value > 3) return fail(ctx, KZG_ERR_SHAPE, "ntt_xcd must be 0..3");
        ctx->opt_ntt_xcd = (int)value;
    } else if (k == "ntt_kernel") {
        if (value < 0 || value > 2) return fail(ctx, KZG_ERR_SHAPE, "ntt_kernel must be 0..2");
        ctx->opt_ntt_kernel = (int)value;
    } else if (k == "ntt_three_from") {
        if (value != 0 && (value < 20 || value > 24)) return fail(ctx, KZG_ERR_SHAPE, "ntt_three_from must be 0 (never) or 20..24");
        ctx->opt_ntt_three_from = (int)value;
    } else if (k == "ntt_vec2_log") {
        if (value < 0 || value > 2) return fail(ctx, KZG_ERR_SHAPE, "ntt_vec2_log must be 0..2");
        ctx->opt_ntt_vec2_log = (int)value;
    } else if (k == "ntt_vec_log") {
        if (value < 0 || value > 2) return fail(ctx, KZG_ERR_SHAPE, "ntt_vec_log must be 0..2");
        ctx->opt_ntt_vec_log = (int)value;
    } else {
        return fail(ctx, KZG_ERR_SHAPE, "unknown option " + k);
    }
    return KZG_OK;
}

// ---------------------------------------------------------------------------------------------
// device memory + profiling
// ---------------------------------------------------------------------------------------------
// kzg_dev_alloc / upload / download do not take the context exclusively (the HIP calls are thread-safe by themselves): a host
// thread staging its next polynomial must not drain the other threads' commits.  Every entry point has synchronised its own
// work before it returned, so a plain copy sees the results of all completed calls.
extern "C" int kzg_dev_alloc(kzg_ctx *ctx, size_t bytes, void **out) {
    if (!ctx || !out) return KZG_ERR_SHAPE;
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(out, bytes ? bytes : 16);
    if (e != hipSuccess) return fail(ctx, KZG_ERR_ALLOC, hipGetErrorString(e));
    return KZG_OK;
}
extern "C" int kzg_dev_free(kzg_ctx *ctx, void *p) {
    if (!ctx) return KZG_ERR_SHAPE;
    Guard g(ctx);
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    for (auto &l : ctx->lanes) hipStreamSynchronize(l.stream);
    if (p) KZG_HIP_CHECK(ctx, hipFree(p));
    return KZG_OK;
}
extern "C" int kzg_dev_upload(kzg_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes) {
    if (!ctx) return KZG_ERR_SHAPE;
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (bytes) KZG_HIP_CHECK(ctx, hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
    return KZG_OK;
}
extern "C" int kzg_dev_download(kzg_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes) {
    if (!ctx) return KZG_ERR_SHAPE;
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (bytes) KZG_HIP_CHECK(ctx, hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
    return KZG_OK;
}

extern "C" int kzg_prof_enable(kzg_ctx *ctx, int on) {
    if (!ctx) return KZG_ERR_SHAPE;
    Guard g(ctx);
    ctx->prof = on != 0;
    ctx->prof_only_accum = on == 2;
    return KZG_OK;
}
extern "C" int kzg_prof_reset(kzg_ctx *ctx) {
    if (!ctx) return KZG_ERR_SHAPE;
    Guard g(ctx);
    prof_collect(ctx);
    ctx->prof_map.clear();
    return KZG_OK;
}
extern "C" int kzg_prof_get(kzg_ctx *ctx, const char *kernel, uint64_t *launches, double *total_ms) {
    if (!ctx || !kernel) return KZG_ERR_SHAPE;
    if (!strcmp(kernel, "point_set_cache")) {  // create_witness_batched's opening-point-set cache: launches = hits, total_ms = misses
        uint64_t h = 0, m = 0;
        point_set_stats(ctx, &h, &m);
        if (launches) *launches = h;
        if (total_ms) *total_ms = (double)m;
        return KZG_OK;
    }
    Guard g(ctx);
    prof_collect(ctx);
    auto it = ctx->prof_map.find(kernel);
    if (launches) *launches = it == ctx->prof_map.end() ? 0 : it->second.launches;
    if (total_ms) *total_ms = it == ctx->prof_map.end() ? 0.0 : it->second.total_ms;
    return KZG_OK;
}
extern "C" int kzg_prof_names(kzg_ctx *ctx, char *buf, size_t buflen) {
    if (!ctx || !buf || !buflen) return KZG_ERR_SHAPE;
    Guard g(ctx);
    prof_collect(ctx);
    std::string s;
    for (auto &kv : ctx->prof_map) {
        if (!s.empty()) s += ",";
        s += kv.first;
    }
    snprintf(buf, buflen, "%s", s.c_str());
    return KZG_OK;
}

// The chip's v_mad_i64_i32 issue rate, measured on THIS device now (bench.py's roofline peak: boxes of one pool differ by
// several percent and the clock a box sustains under this load is not its nominal one).  Eight independent accumulator chains per
// lane, 8 waves per SIMD, ~30 ms: the same loop as tools/mad_issue.hip / tools/microbench.hip.
__global__ __launch_bounds__(256) void k_mad_issue_rate(uint32_t *out, int iters, uint32_t seed) {
    int32_t a = (int32_t)(seed + threadIdx.x), b = (int32_t)(seed * 3 + blockIdx.x);
    uint64_t c0 = a, c1 = b, c2 = a ^ b, c3 = a + b, c4 = a * 3, c5 = b * 5, c6 = a * 7, c7 = b * 9;
    for (int i = 0; i < iters; i++) {
#define KZG_M(c) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b) : "vcc");
        KZG_M(c0) KZG_M(c1) KZG_M(c2) KZG_M(c3) KZG_M(c4) KZG_M(c5) KZG_M(c6) KZG_M(c7)
#undef KZG_M
    }
    uint64_t s = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}

extern "C" int kzg_measure_mad_issue_rate(kzg_ctx *ctx, int waves_per_simd, double *tera_lane_mads_per_s) {
    if (!ctx || !tera_lane_mads_per_s || waves_per_simd < 1 || waves_per_simd > 8) return KZG_ERR_SHAPE;
    Guard g(ctx);
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const int blocks = ctx->num_cus * waves_per_simd, iters = 25000 * waves_per_simd;
    KZG_TRY(lane_reserve(ctx, 0, (size_t)blocks * 256 * 4 + 4096));
    uint32_t *out = (uint32_t *)lane_alloc(ctx, 0, (size_t)blocks * 256 * 4);
    if (!out) return fail(ctx, KZG_ERR_ALLOC, "workspace");
    hipStream_t st = ctx->lanes[0].stream;
    hipEvent_t e0, e1;
    KZG_HIP_CHECK(ctx, hipEventCreate(&e0));
    KZG_HIP_CHECK(ctx, hipEventCreate(&e1));
    hipLaunchKernelGGL(k_mad_issue_rate, dim3(blocks), dim3(256), 0, st, out, iters / 4, 7u);  // warm-up
    hipEventRecord(e0, st);
    hipLaunchKernelGGL(k_mad_issue_rate, dim3(blocks), dim3(256), 0, st, out, iters, 7u);
    hipEventRecord(e1, st);
    hipError_t e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (e != hipSuccess || ms <= 0.f) return fail(ctx, KZG_ERR_HIP, "mad issue-rate measurement failed");
    *tera_lane_mads_per_s = (double)blocks * 256.0 * (double)iters * 8.0 / ((double)ms * 1e-3) / 1e12;
    return KZG_OK;
}

// ---------------------------------------------------------------------------------------------
// MSM
// ---------------------------------------------------------------------------------------------
static int plan_pipeline(kzg_ctx *ctx, int want, int *nl_out, int *nas_out);
static void set_lane_mode(kzg_ctx *ctx, int lane, bool pipelined, bool deep);

// ---- leased lanes: the reference's blocking prover calls from many host threads (CtxGate, common.h) -------------------------
kzg::Lease::~Lease() {
    if (!ctx || lane < 0) return;
    set_lane_mode(ctx, lane, false, false);  // exclusive callers (lane 0) find the lone-MSM shape
    {
        std::lock_guard<std::mutex> lk(ctx->mu.m);
        ctx->mu.lane_busy &= ~(1u << lane);
        ctx->mu.shared_active--;
    }
    ctx->mu.cv.notify_all();
}

// the pool concurrent callers lease from: `streams` lanes + the accumulation streams, planned like a batch of that depth
static int plan_for_callers(kzg_ctx *ctx) {
    int nl = 1, nas = 0;
    KZG_TRY(plan_pipeline(ctx, ctx->opt_streams, &nl, &nas));
    ctx->pipe_lanes = nl;
    ctx->pipe_accum = nas;
    ctx->pipe_planned = true;
    return KZG_OK;
}

int kzg::lease_lane(kzg_ctx *ctx, Lease *ls) {
    CtxGate &g = ctx->mu;
    std::unique_lock<std::mutex> lk(g.m);
    for (;;) {
        g.cv.wait(lk, [&] {
            if (g.exclusive || g.excl_waiting) return false;
            if (g.shared_active == 0 || !ctx->pipe_planned) return true;
            return (g.lane_busy & ((1u << ctx->pipe_lanes) - 1u)) != ((1u << ctx->pipe_lanes) - 1u);
        });
        if (g.shared_active == 0 || ctx->pipe_planned) break;
        // a second caller and no plan yet: take the context exclusively once (waits for the first caller), plan, try again
        lk.unlock();
        int rc;
        {
            Guard ex(ctx);
            rc = hipSetDevice(ctx->device) == hipSuccess ? KZG_OK : fail(ctx, KZG_ERR_HIP, "hipSetDevice");
            if (rc == KZG_OK && !ctx->pipe_planned) rc = plan_for_callers(ctx);
        }
        if (rc != KZG_OK) return rc;
        lk.lock();
    }
    const int others = g.shared_active;
    int lane = 0;
    if (ctx->pipe_planned)
        while (lane < ctx->pipe_lanes && (g.lane_busy >> lane & 1u)) lane++;
    // (unplanned: only reached with no other caller active, lane 0)
    g.lane_busy |= 1u << lane;
    g.shared_active++;
    ls->ctx = ctx;
    ls->lane = lane;
    // a lone caller gets the latency shape (full accumulation grid on its own stream, quad-lane tail kernels); with others in
    // flight the call is one stage of a pipeline: batch-sized grid on a FIFO accumulation stream, lane-time tail
    const bool pipelined = others > 0 && ctx->pipe_planned;
    set_lane_mode(ctx, lane, pipelined, others >= 3);
    if (pipelined && ctx->pipe_accum > 0) {
        ls->pipelined = true;
        ls->slot = ctx->accum_rr.fetch_add(1);
    }
    return KZG_OK;
}

// how many of the `planned` accumulation streams MSMs of this size are spread over (small ones: all; common.h opt_accum_streams_small)
static int accum_streams_for(kzg_ctx *ctx, int planned, const kzg_srs *srs, size_t n) {
    if (planned <= 0) return 1;
    if (ctx->msm_small((size_t)srs->W * n)) return planned;
    return planned < ctx->opt_accum_streams ? planned : (ctx->opt_accum_streams > 0 ? ctx->opt_accum_streams : 1);
}

int kzg::lease_msm(kzg_ctx *ctx, const Lease &ls, const kzg_srs *srs, size_t offset, const void *d_sc, size_t n, int sfmt, MsmPoint **res) {
    if (ls.pipelined) {
        hipStream_t accum = ctx->accum_streams[ls.slot % (uint32_t)accum_streams_for(ctx, ctx->pipe_accum, srs, n)];
        return msm_run(ctx, ls.lane, srs, offset, d_sc, n, sfmt, res, accum, ctx->sorted_events[ls.lane], ctx->accum_events[ls.lane]);
    }
    return msm_run(ctx, ls.lane, srs, offset, d_sc, n, sfmt, res);
}

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
// Item b runs on lane b % nl; every bucket-accumulation kernel goes to one of the dedicated FIFO streams.
// Which of the context's streams sit on hardware queues of their own?  The runtime multiplexes streams onto its queue pool
// (GPU_MAX_HW_QUEUES, default 4, minus whatever the rest of the process uses; the assignment is not a plain round robin), and
// two streams on one queue run their kernels strictly one after the other.  Measured, once per context: a 0.3 ms spin kernel
// goes to one stream and a time-stamp kernel to every stream not yet classified -- a stamp taken after the spin ended waited
// behind it, i.e. shares its queue.  The streams are then re-ordered so that the first `probed_queues` of them (lanes first,
// then accumulation streams) are pairwise on different queues.
__global__ void k_probe_spin(unsigned long long ticks, unsigned long long *out) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
    }
    *out = wall_clock64();
}
__global__ void k_probe_mark(unsigned long long *out) { *out = wall_clock64(); }

static int probe_queues(kzg_ctx *ctx, int nl, int nas) {
    std::vector<hipStream_t *> ss;
    for (int l = 0; l < nl; l++) ss.push_back(&ctx->lanes[l].stream);
    for (int i = 0; i < nas; i++) ss.push_back(&ctx->accum_streams[i]);
    const size_t K = ss.size();
    std::vector<int> cls(K, -1);
    int ncls = 0;
    // Streams of the process' pool that an earlier context has measured keep their classes (no spin kernels on streams another
    // context may be busy on: its work would delay the marks and read as "shares a queue").
    std::unique_lock<std::mutex> plk(ctx->pool->mu);
    bool known = !ctx->pool->cls.empty();
    for (size_t t = 0; t < K && known; t++)
        if (!ctx->pool->cls.count(*ss[t])) known = false;
    if (known) {
        std::map<int, int> remap;
        for (size_t t = 0; t < K; t++) {
            const int c = ctx->pool->cls[*ss[t]];
            if (!remap.count(c)) remap[c] = ncls++;
            cls[t] = remap[c];
        }
    }
    unsigned long long *d = nullptr;
    if (!known) KZG_HIP_CHECK(ctx, hipMalloc((void **)&d, K * sizeof(unsigned long long)));
    std::vector<unsigned long long> h(K);
    for (size_t s0 = 0; s0 < K && !known; s0++) {
        if (cls[s0] != -1) continue;
        cls[s0] = ncls;
        hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(1), 0, *ss[s0], 30000ull, d + s0);  // wall_clock64: 100 MHz
        for (size_t t = s0 + 1; t < K; t++)
            if (cls[t] == -1) hipLaunchKernelGGL(k_probe_mark, dim3(1), dim3(1), 0, *ss[t], d + t);
        for (size_t t = s0; t < K; t++) KZG_HIP_CHECK(ctx, hipStreamSynchronize(*ss[t]));
        KZG_HIP_CHECK(ctx, hipMemcpy(h.data(), d, K * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (size_t t = s0 + 1; t < K; t++)
            if (cls[t] == -1 && h[t] >= h[s0]) cls[t] = ncls;
        ncls++;
    }
    if (!known) {
        hipFree(d);
        ctx->pool->cls.clear();
        for (size_t t = 0; t < K; t++) ctx->pool->cls[*ss[t]] = cls[t];
    }
    plk.unlock();
    // one representative per queue first, the sharers after them
    std::vector<hipStream_t> order;
    std::vector<bool> seen(ncls, false);
    for (size_t t = 0; t < K; t++)
        if (!seen[cls[t]]) {
            seen[cls[t]] = true;
            order.push_back(*ss[t]);
        }
    std::vector<bool> seen2(ncls, false);
    for (size_t t = 0; t < K; t++) {
        if (seen2[cls[t]]) order.push_back(*ss[t]);
        seen2[cls[t]] = true;
    }
    ctx->probed_queues = ncls;
    ctx->probed_order = order;
    if (getenv("KZG_DEBUG")) fprintf(stderr, "kzg: probe_queues: %zu streams on %d hardware queues\n", K, ncls);
    return KZG_OK;
}

struct BatchPipe {
    int nl = 1, nas = 0;
    uint8_t *d_out = nullptr;
    bool out_dev = false;
};

static void set_lane_mode(kzg_ctx *ctx, int lane, bool pipelined, bool deep) {
    MsmMode &m = ctx->lanes[lane].mode;
    m.accum_blocks = pipelined ? ctx->accum_blocks_batch() : ctx->accum_blocks_single();
    m.sort_threads = pipelined ? ctx->opt_sort_threads_batch : ctx->opt_sort_threads;
    m.tail_quads = pipelined ? false : ctx->opt_tail_quads != 0;
    m.tail_wide = pipelined && deep;
}

// Lanes and accumulation streams for a pipeline of up to `want` MSMs in flight (exclusive callers only: streams are created,
// probed and re-ordered here).  Every stream must map to a hardware queue of its own (streams that share a queue serialise: a
// lane's tail kernels would wait behind another lane's accumulation).  How many queues the process really has is MEASURED once
// per context (probe_queues), so the plan does not depend on what the host exported before HIP initialised; with fewer queues
// than lanes + accumulation streams the pipeline is narrowed to fit (measured on 4 queues: 3 lanes + 1 accumulation stream
// 392/s, 2 + 2: 330/s, 4 + 0: 383/s, against 405/s with 18 queues; profiles/r02_hw_queues.txt).
static int plan_pipeline(kzg_ctx *ctx, int want, int *nl_out, int *nas_out) {
    // (small MSMs are spread over more accumulation streams than large ones: the plan holds the larger number)
    int nl = want, nas = want > 1 ? ctx->opt_accum_streams : 0;
    if (nas > 0 && ctx->opt_accum_blocks_small > 0 && ctx->opt_accum_streams_small > nas) nas = ctx->opt_accum_streams_small;
    const int want_nl = nl, want_nas = nas;
    int found_queues = 0;
    if (want > 1) {
        KZG_TRY(ensure_lanes(ctx, want));
        for (int i = 0; i < nas; i++)
            if (!ctx->accum_streams[i]) KZG_HIP_CHECK(ctx, pool_accum(ctx->pool, i, &ctx->accum_streams[i]));
        int queues = ctx->opt_hw_queues;
        if (queues <= 0) {
            if (ctx->probed_queues == 0 || ctx->probed_lanes < want || ctx->probed_accum < nas) {
                KZG_TRY(probe_queues(ctx, want, nas));
                ctx->probed_lanes = want;
                ctx->probed_accum = nas;
            }
            queues = ctx->probed_queues;
        }
        found_queues = queues;
        if (nl + nas > queues) {
            if (queues >= 4) {  // measured (profiles/r02_hw_queues.txt): Q = 4: 3 + 1 best; Q = 8: 6 + 2; Q = 12: 10 + 2; "many + 1" loses 10 %
                nas = nas ? (queues >= 6 && nas >= 2 ? 2 : 1) : 0;
                nl = queues - nas;
            } else {
                nas = 0;
                nl = queues > 0 ? queues : 1;
            }
            if (nl > want) nl = want;
        }
        // hand the probed streams out so that the ones this plan uses are on different queues: lanes first, accumulation
        // streams next; the others stay parked in the remaining probed slots
        if (ctx->opt_hw_queues <= 0 && (int)ctx->probed_order.size() == ctx->probed_lanes + ctx->probed_accum &&
            nl <= ctx->probed_lanes && nas <= ctx->probed_accum) {
            for (auto &l : ctx->lanes) KZG_HIP_CHECK(ctx, hipStreamSynchronize(l.stream));
            size_t r = 0;
            for (int l = 0; l < nl; l++) ctx->lanes[l].stream = ctx->probed_order[r++];
            for (int i = 0; i < nas; i++) ctx->accum_streams[i] = ctx->probed_order[r++];
            for (int l = nl; l < ctx->probed_lanes; l++) ctx->lanes[l].stream = ctx->probed_order[r++];
            for (int i = nas; i < ctx->probed_accum; i++) ctx->accum_streams[i] = ctx->probed_order[r++];
            ctx->pipe_planned = false;  // the lease plan (below) re-derives itself from the new order
        }
    }
    if (getenv("KZG_DEBUG")) fprintf(stderr, "kzg: pipeline plan: %d lanes + %d accumulation streams\n", nl, nas);
    if (want > 1) {
        ctx->plan_want_lanes = want_nl, ctx->plan_want_accum = want_nas, ctx->plan_lanes = nl, ctx->plan_accum = nas, ctx->plan_queues = found_queues;
        if ((nl < want_nl || nas < want_nas) && !ctx->plan_warned) {
            // the silent cliff of round 4 (VERDICT weak #13): say it where the host's operator will see it, once per context
            ctx->plan_warned = true;
            fprintf(stderr, "kzg: device %d: this context's %d + %d streams found only %d hardware queues of their own (other streams of the "
                            "process hold the rest: another context, an RCCL communicator, the host's); the batched pipeline is narrowed to %d "
                            "lanes + %d accumulation streams -- expect 10-25 %% less batched throughput from THIS context.  Create contexts before "
                            "communicators, give the process more queues (kzg_init_hw_queues / GPU_MAX_HW_QUEUES before the first HIP call) or set "
                            "option \"streams\" explicitly; kzg_ctx_info reports the plan.\n",
                    ctx->device, want_nl, want_nas, found_queues, nl, nas);
        }
    }
    KZG_TRY(ensure_lanes(ctx, nl));
    while (nas && (int)ctx->sorted_events.size() < 2 * nl) {  // [0, nl): the lanes' own; [nl, 2 nl): their second MSM in flight (batch_msm)
        hipEvent_t e1 = nullptr, e2 = nullptr;
        KZG_HIP_CHECK(ctx, hipEventCreateWithFlags(&e1, hipEventDisableTiming));
        KZG_HIP_CHECK(ctx, hipEventCreateWithFlags(&e2, hipEventDisableTiming));
        ctx->sorted_events.push_back(e1);
        ctx->accum_events.push_back(e2);
    }
    ctx->planned_accum = nas;
    *nl_out = nl;
    *nas_out = nas;
    return KZG_OK;
}

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
