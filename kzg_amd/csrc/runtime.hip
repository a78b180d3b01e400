// runtime.hip -- what the entry points of capi.hip run ON: the context (kzg_ctx_create / destroy / options / info), the device's
// shared stream pool and the hardware-queue probe, the lanes (arena, pinned staging, leasing for concurrent blocking callers, the
// pipeline plan), device memory helpers and profiling (HIP events on the stream each kernel is launched on).  Split out of capi.hip in
// round 6; the functions other translation units use are declared in common.h.
#include <dlfcn.h>

#include <algorithm>

#include "common.h"

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

}  // namespace kzg

using namespace kzg;

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
extern "C" const char *kzg_version(void) { return "kzg-mi355x 0.1.0 (gfx950)"; }

// The pipelined paths want one hardware queue per stream (13 lanes + 4 accumulation streams, one exchange stream per device group, and ~6 for an RCCL communicator); the HIP runtime sizes its queue
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
// why every context plans 13 + 4 (option "streams"; round 5 planned 14 + 4 -- 474.2 against 471.2 commitments/s at 16: no loss; round 6 gave the
// 14th lane's queue to the device group's own exchange stream: with 14 + 4 + that stream + RCCL's six the 25th stream shared a queue and the group path
// fell from 465 to 337 commitments/s, with 13 lanes it is back at 475 and a plain context measures 471-472 against 473-474, profiles/r06_group_exchange_stream.txt) -- a
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

int kzg::ensure_lanes(kzg_ctx *ctx, int want) {  // exclusive callers only
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
int kzg::accum_streams_for(kzg_ctx *ctx, int planned, const kzg_srs *srs, size_t n) {
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

void kzg::set_lane_mode(kzg_ctx *ctx, int lane, bool pipelined, bool deep) {
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
int kzg::plan_pipeline(kzg_ctx *ctx, int want, int *nl_out, int *nas_out) {
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

