// mgpu.hip -- the multi-GPU commit path behind the C ABI: a group of GPUs (kzg_mctx), an SRS sharded contiguously over it
// (kzg_msrs), per-GPU partial MSMs and the RCCL exchange of the 144-byte partial points.
//
// Replaces the multi_exp call of KZGProver::commit / create_witness (src/coeff_form.rs:61,78) when the SRS does not (or
// should not) live on one GPU: rank r holds gs[lo_r, hi_r) resident, reduces coeffs[lo_r, hi_r) to one Jacobian point on its
// GPU (the whole bucket pipeline of msm.hip, locally), and ONE ncclAllGather moves world x batch x 144 bytes over xGMI; every
// rank then adds the `world` partials of each polynomial (k_sum_groups) and converts to affine once.  Exchanging buckets
// instead of reduced partials would move ~2^16 x 144 B per rank and commitment and be link-bound; this way the collective is
// latency-bound and independent of the polynomial size.
//
// RCCL is dlopen'ed (librccl.so.1) on first use so that single-GPU hosts need not have it and so that a process which already
// loaded a copy (PyTorch ships one under the same SONAME) shares that copy.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <functional>
#include <memory>
#include <system_error>
#include <thread>

#include "common.h"
#ifdef KZG_TEST_HOOKS
#include "../../include/kzg_mi355x_test.h"
#include "test_transport.h"
#endif

namespace kzg {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional: the time-out path (a dead peer) falls back to leaking the communicator
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string path;      // the file the adopted RCCL was loaded from
    std::string hip_path;  // the HIP runtime it is bound to
    int version = 0;
    double load_ms = 0;    // dlopen + symbol resolution of the adopted copy (a 570 MB file on a cold box)
    double uid_ms = -1;    // the last ncclGetUniqueId (creates the bootstrap root: the first thing that touches the network stack)
};

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static bool debug_on() {
    static const bool on = getenv("KZG_DEBUG") != nullptr;
    return on;
}
#define KZG_DBG(...)                              \
    do {                                          \
        if (debug_on()) {                         \
            fprintf(stderr, "[kzg mgpu] " __VA_ARGS__); \
            fputc('\n', stderr);                  \
        }                                         \
    } while (0)

// RCCL's communicator calls (ncclGetUniqueId, ncclCommInitRank / ncclCommInitAll, ncclCommDestroy) have no deadline of their own:
// the bootstrap is TCP over whatever interface RCCL picks, and on a host whose non-loopback interface swallows packets a
// world-1 ncclCommInitRank was seen to return only after five minutes (round 4's driver box).  They therefore run on a helper
// thread and the caller waits with a deadline.  A call that does not come back is ABANDONED: its thread is detached (it owns its
// state through a shared_ptr), and since it may hold RCCL's internal locks for ever, every later communicator call of this
// process fails at once (g_rccl_wedged) instead of queueing up behind it.
static std::atomic<bool> g_rccl_wedged{false};
static std::mutex g_wedge_mu;
static std::string g_wedge_msg;

struct BoundedState {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    ncclResult_t res = ncclSuccess;
    std::vector<ncclComm_t> comms;  // what a formation produced (copied out by the caller when it was still waiting)
    ncclUniqueId uid;
};

// true = f returned within the deadline (*res, *ms set); false = abandoned.  timeout_ms <= 0: inline, unbounded.
static bool run_bounded(int64_t timeout_ms, const std::shared_ptr<BoundedState> &st, std::function<ncclResult_t(BoundedState &)> f,
                        ncclResult_t *res, double *ms) {
    const double t0 = now_ms();
    if (timeout_ms <= 0) {
        *res = f(*st);
        *ms = now_ms() - t0;
        return true;
    }
    std::thread th;
    try {
        th = std::thread([st, f]() {
            ncclResult_t r = f(*st);
            std::lock_guard<std::mutex> lk(st->mu);
            st->res = r;
            st->done = true;
            st->cv.notify_all();
        });
    } catch (const std::system_error &) {  // no thread to be had: the call runs here, unbounded (no exception crosses the C ABI)
        *res = f(*st);
        *ms = now_ms() - t0;
        return true;
    }
    std::unique_lock<std::mutex> lk(st->mu);
    const bool ok = st->cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return st->done; });
    lk.unlock();
    *ms = now_ms() - t0;
    if (ok) {
        th.join();
        *res = st->res;
        return true;
    }
    th.detach();
    return false;
}

static int64_t default_comm_timeout_ms() {
    const char *e = getenv("KZG_COMM_TIMEOUT_MS");  // process-wide default (a group formed inside kzg_mctx_create* has no option call before it)
    if (e && *e) return atoll(e) < 0 ? 0 : atoll(e);
    return 60000;
}

// the shared object that holds `addr`
static std::string object_of(const void *addr) {
    Dl_info di;
    if (addr && dladdr(addr, &di) && di.dli_fname) return di.dli_fname;
    return "?";
}

static std::mutex g_rccl_mu;
static Rccl *g_rccl = nullptr;

// One candidate: resolve the entry points and find out which HIP runtime the copy is bound to.
static Rccl *rccl_try(void *h, std::string *err) {
    Rccl *r = new Rccl();
    r->handle = h;
    bool ok = true;
    auto sym = [&](const char *n) {
        void *p = dlsym(h, n);
        if (!p) {
            ok = false;
            *err = std::string("RCCL symbol missing: ") + n;
        }
        return p;
    };
    r->GetUniqueId = (decltype(r->GetUniqueId))sym("ncclGetUniqueId");
    r->CommInitRank = (decltype(r->CommInitRank))sym("ncclCommInitRank");
    r->CommInitAll = (decltype(r->CommInitAll))sym("ncclCommInitAll");
    r->CommDestroy = (decltype(r->CommDestroy))sym("ncclCommDestroy");
    r->AllGather = (decltype(r->AllGather))sym("ncclAllGather");
    r->GroupStart = (decltype(r->GroupStart))sym("ncclGroupStart");
    r->GroupEnd = (decltype(r->GroupEnd))sym("ncclGroupEnd");
    r->GetErrorString = (decltype(r->GetErrorString))sym("ncclGetErrorString");
    r->GetVersion = (decltype(r->GetVersion))sym("ncclGetVersion");
    if (ok) r->CommAbort = (decltype(r->CommAbort))dlsym(h, "ncclCommAbort");
    if (!ok) {
        delete r;
        return nullptr;
    }
    r->path = object_of((const void *)r->AllGather);
    r->GetVersion(&r->version);
    r->hip_path = object_of(dlsym(h, "hipStreamSynchronize"));  // resolved through this copy's own dependency chain
    return r;
}

// Which RCCL?  A stream and device pointers are handed across, so the copy must be bound to the SAME HIP runtime as this
// library (PyTorch wheels ship their own librccl / libamdhip64 next to /opt/rocm's: two runtimes in one process must not be
// mixed).  Candidates in order: the copy the process already holds (RTLD_NOLOAD: e.g. PyTorch's), librccl.so.1 from the library
// path, /opt/rocm's.  The first one on this library's runtime is adopted (kzg_mctx_info reports it); if every loadable copy is
// on another runtime the group refuses to form.
static Rccl *rccl_load(std::string *err) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl) return g_rccl;
    const double t_load0 = now_ms();
    const std::string mine = object_of((const void *)&hipStreamSynchronize);
#ifdef KZG_TEST_HOOKS
    if (getenv("KZG_TEST_SHM_TRANSPORT")) {  // tests: world > 1 on ONE GPU, which RCCL refuses (test_transport.h)
        Rccl *r = new Rccl();
        r->GetUniqueId = shmt::get_unique_id;
        r->CommInitRank = shmt::comm_init_rank;
        r->CommInitAll = shmt::comm_init_all;
        r->CommDestroy = shmt::comm_destroy;
        r->CommAbort = shmt::comm_destroy;
        r->AllGather = shmt::all_gather;
        r->GroupStart = shmt::group_start;
        r->GroupEnd = shmt::group_end;
        r->GetVersion = shmt::get_version;
        r->GetErrorString = shmt::error_string;
        r->path = "test-shm-transport";
        r->hip_path = mine;
        g_rccl = r;
        return r;
    }
#endif
    const char *names[] = {nullptr, "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    std::string load_err = "not found", mismatches;
    bool any = false;
    for (int i = 0; i < 4; i++) {
        void *h = i == 0 ? dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD) : dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
#ifdef KZG_TEST_HOOKS
        if (getenv("KZG_TEST_NO_RCCL")) {  // tests: the load-failure path on a host that does have RCCL
            if (h) dlclose(h);
            h = dlopen("librccl-not-installed.so.1", RTLD_NOW | RTLD_LOCAL);
        }
#endif
        if (!h) {
            const char *e = dlerror();  // read once: glibc clears the message on the first read
            if (e) load_err = e;
            continue;
        }
        any = true;
        std::string serr;
        Rccl *r = rccl_try(h, &serr);
        if (!r) {
            load_err = serr;
            continue;
        }
        if (r->hip_path != "?" && mine != "?" && r->hip_path != mine) {
            mismatches += (mismatches.empty() ? "" : "; ") + r->path + " -> " + r->hip_path;
            delete r;
            continue;
        }
        r->load_ms = now_ms() - t_load0;
        KZG_DBG("RCCL loaded in %.1f ms: %s version %d, HIP runtime %s", r->load_ms, r->path.c_str(), r->version, r->hip_path.c_str());
        g_rccl = r;
        return r;
    }
    if (any && !mismatches.empty())
        *err = "every loadable RCCL is bound to another HIP runtime than this library (" + mine + "): " + mismatches +
               " -- refusing to share streams across two runtimes";
    else
        *err = std::string("cannot load RCCL (librccl.so.1): ") + load_err;
    return nullptr;
}

constexpr size_t PARTIAL_BYTES = 144;  // KZG_G1_JACOBIAN_MONT_144: no field inversion per partial

}  // namespace kzg

using namespace kzg;

struct kzg_mctx {
    int world = 1;
    bool per_process = false;        // one process per GPU (ncclCommInitRank) vs one process driving all (ncclCommInitAll)
    std::vector<int> devices;        // local GPUs
    std::vector<int> ranks;          // their global ranks
    std::vector<kzg_ctx *> ctxs;
    std::vector<ncclComm_t> comms;   // one per local GPU once the communicator exists
    ncclUniqueId uid;                // per-process mode: kept until the communicator is created
    bool always_gather = false;
    // A peer that never arrives (a dead process, a hung GPU) must not hang the survivors inside the exchange: the wait behind the
    // all-gather is a poll with this deadline (option "gather_timeout_ms", 0 = wait for ever).  When it expires the communicators
    // are aborted (ncclCommAbort), the call returns KZG_ERR_INTERNAL and the group is DEAD: every later call on it fails at once;
    // the host destroys it and forms a new one.
    int64_t gather_timeout_ms = 60000;
    // ... and neither must communicator FORMATION (ncclCommInitRank / ncclCommInitAll): option "comm_timeout_ms" (0 = wait for
    // ever; default 60 s or KZG_COMM_TIMEOUT_MS).  On expiry the call returns KZG_ERR_INTERNAL with the phase timings in
    // kzg_mctx_last_error, the group is dead and the process forms no further communicators (run_bounded above).
    int64_t comm_timeout_ms = default_comm_timeout_ms();
    double t_init_ms = -1;           // ncclCommInit* (wall), -1 = no communicator was ever formed
    double t_first_ms = -1;          // the first exchange: enqueue -> complete (RCCL loads its kernels' code object here)
    double t_destroy_ms = -1;
    bool dead = false;
    std::vector<void *> d_stat;      // per local GPU: world x STATUS_BYTES for the status-only agreement (allocated with the group)
    int inject_alloc_fail = 0;       // KZG_TEST_HOOKS: the next growth of the exchange buffers fails on local GPU 0
    int inject_stall_ms = 0;         // KZG_TEST_HOOKS: the next exchange sits behind a spin kernel of this length (a late peer)
    std::mutex mu;
    std::string err;
    // grow-only exchange buffers per local GPU: partials of this GPU, partials of every rank
    std::vector<void *> d_part, d_gath;
    std::vector<size_t> cap_points;
    size_t agreed_points = 64;       // the batch every rank is known to hold buffers for (grown only by a successful agreement)
    size_t agreed_quot = 0;          // ... and the quotient length
    std::vector<void *> d_quot;      // create_witness: the quotient polynomial on each GPU
    std::vector<size_t> cap_quot;
    std::vector<void *> h_status;    // pinned: the status words of all ranks after the exchange, per local GPU
    // The exchange has a stream of ITS OWN per local GPU -- not a lane of the device's shared stream pool (runtime.hip, StreamPool): the
    // group's deadlines decide "exchange done / stream stuck" from this stream alone, so a plain Engine of the same device that is busy
    // on pool lane 0 cannot make a healthy exchange read as NotReady (and get the communicators aborted), and a collective that never
    // leaves its stream cannot hang another prover's hipStreamSynchronize.  x_ready orders the exchange behind the local
    // phase on the context's lane 0; x_done, recorded right behind the collective and its status download, is what the wait polls.
    std::vector<hipStream_t> xstream;
    std::vector<hipEvent_t> x_ready, x_done;
    int inject_fail = 0;             // KZG_TEST_HOOKS: the next local phase fails with this code on local GPU 0
    int nlocal() const { return (int)devices.size(); }
    // one persistent host thread per local GPU (groups of several GPUs in one process): a sharded call hands each of them its
    // GPU's share and waits -- no thread creation on the path of a 2 ms operation
    struct Worker {
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        std::function<int()> job;
        bool has_job = false, done = false, quit = false;
        int rc = 0;
    };
    std::vector<Worker *> workers;
};

static void worker_main(kzg_mctx::Worker *w) {
    std::unique_lock<std::mutex> lk(w->mu);
    for (;;) {
        w->cv.wait(lk, [&] { return w->has_job || w->quit; });
        if (w->quit) return;
        std::function<int()> job = std::move(w->job);
        w->has_job = false;
        lk.unlock();
        int rc = job();
        lk.lock();
        w->rc = rc;
        w->done = true;
        w->cv.notify_all();
    }
}

static void workers_start(kzg_mctx *m) {
    if (m->nlocal() < 2 || !m->workers.empty()) return;
    for (int i = 0; i < m->nlocal(); i++) {
        kzg_mctx::Worker *w = new kzg_mctx::Worker();
        w->th = std::thread(worker_main, w);
        m->workers.push_back(w);
    }
}

static void workers_stop(kzg_mctx *m) {
    for (kzg_mctx::Worker *w : m->workers) {
        {
            std::lock_guard<std::mutex> lk(w->mu);
            w->quit = true;
        }
        w->cv.notify_all();
        w->th.join();
        delete w;
    }
    m->workers.clear();
}

struct kzg_msrs {
    size_t n = 0;
    std::vector<kzg_srs *> shards;   // one per local GPU
    std::vector<size_t> first, len;
};

static std::mutex &merr_mu();
static std::string &create_err() {  // why this thread's last kzg_mctx_create* / kzg_mctx_unique_id failed (no group to ask)
    static thread_local std::string e;
    return e;
}
extern "C" const char *kzg_mctx_create_error(void) { return create_err().c_str(); }
static int mfail(kzg_mctx *m, int code, const std::string &msg) {
    std::lock_guard<std::mutex> lk(merr_mu());
    m->err = msg;
    return code;
}

static int mfail_ctx(kzg_mctx *m, int i, int code) {
    const char *e = kzg_last_error(m->ctxs[i]);
    return mfail(m, code, "GPU " + std::to_string(m->devices[i]) + ": " + (e ? e : ""));
}

#define KZG_NCCL(m, r, expr)                                                                                       \
    do {                                                                                                           \
        ncclResult_t _e = (expr);                                                                                  \
        if (_e != ncclSuccess) return mfail((m), KZG_ERR_HIP, std::string(#expr) + ": " + (r)->GetErrorString(_e)); \
    } while (0)

extern "C" int kzg_shard_range(size_t n, int rank, int world, size_t *lo, size_t *hi) {
    if (world <= 0 || rank < 0 || rank >= world) return KZG_ERR_SHAPE;
    size_t base = n / (size_t)world, extra = n % (size_t)world, r = (size_t)rank;
    size_t l = r * base + (r < extra ? r : extra);
    if (lo) *lo = l;
    if (hi) *hi = l + base + (r < extra ? 1 : 0);
    return KZG_OK;
}

static int mctx_buffers_grow(kzg_mctx *m, size_t batch);
static int mctx_make_ctxs(kzg_mctx *m) {
    // sized first: kzg_mctx_destroy walks these per context when a later device fails (devices = [0, 99])
    m->d_part.assign(m->nlocal(), nullptr);
    m->d_gath.assign(m->nlocal(), nullptr);
    m->cap_points.assign(m->nlocal(), 0);
    m->d_quot.assign(m->nlocal(), nullptr);
    m->cap_quot.assign(m->nlocal(), 0);
    m->h_status.assign(m->nlocal(), nullptr);
    m->d_stat.assign(m->nlocal(), nullptr);
    for (int i = 0; i < m->nlocal(); i++) {
        kzg_ctx *c = nullptr;
        int rc = kzg_ctx_create(m->devices[i], &c);
        if (rc != KZG_OK) return rc;
        m->ctxs.push_back(c);
        hipStream_t xs = nullptr;
        hipEvent_t e1 = nullptr, e2 = nullptr;
        hipSetDevice(m->devices[i]);
        const bool made = hipStreamCreateWithFlags(&xs, hipStreamNonBlocking) == hipSuccess &&
                          hipEventCreateWithFlags(&e1, hipEventDisableTiming) == hipSuccess &&
                          hipEventCreateWithFlags(&e2, hipEventDisableTiming) == hipSuccess;
        m->xstream.push_back(xs);
        m->x_ready.push_back(e1);
        m->x_done.push_back(e2);
        if (!made) return mfail(m, KZG_ERR_HIP, "hipStreamCreate / hipEventCreate (the group's exchange stream)");
        // (every context plans 13 lanes + 4 accumulation streams from the process' shared pool, the group's exchange stream above is the
        // 18th stream: an RCCL communicator needs about six
        // of the pool's 24 hardware queues -- runtime.hip, StreamPool; kzg_mctx_set_option(m, "streams", ...) overrides)
    }
    // the exchange buffers of ordinary calls (up to 64 polynomials per call) and the status-only agreement buffer exist from the
    // start: a group that formed can always exchange statuses, whatever fails later
    int rc = mctx_buffers_grow(m, 64);
    if (rc != KZG_OK) return rc;
    workers_start(m);
    return KZG_OK;
}

static std::string phase_report(const kzg_mctx *m, const Rccl *r) {
    char b[256];
    snprintf(b, sizeof b, "load=%.1f uid=%.1f init=%.1f first_exchange=%.1f destroy=%.1f", r ? r->load_ms : -1.0, r ? r->uid_ms : -1.0,
             m->t_init_ms, m->t_first_ms, m->t_destroy_ms);
    return b;
}

static int rccl_wedged_fail(kzg_mctx *m) {
    std::string msg;
    {
        std::lock_guard<std::mutex> lk(g_wedge_mu);
        msg = g_wedge_msg;
    }
    if (m) {
        m->dead = true;
        return mfail(m, KZG_ERR_INTERNAL, "an earlier RCCL communicator call of this process never returned (" + msg +
                                              "): no further communicators are formed in this process");
    }
    return KZG_ERR_INTERNAL;
}

static void rccl_mark_wedged(const std::string &msg) {
    std::lock_guard<std::mutex> lk(g_wedge_mu);
    g_wedge_msg = msg;
    g_rccl_wedged = true;
}

// the communicator is created when the first collective needs it (a group of one GPU never does unless asked to)
static int mctx_comm(kzg_mctx *m, Rccl **out) {
    std::string err;
    Rccl *r = rccl_load(&err);
    if (!r) return mfail(m, KZG_ERR_INTERNAL, err);
    *out = r;
    if (!m->comms.empty()) return KZG_OK;
    if (m->dead) return mfail(m, KZG_ERR_INTERNAL, "this device group is dead (its communicator could not be formed or an exchange timed out): destroy it and form a new one");
    if (g_rccl_wedged) return rccl_wedged_fail(m);
    auto st = std::make_shared<BoundedState>();
    st->comms.assign(m->nlocal(), nullptr);
    st->uid = m->uid;
    const bool per_process = m->per_process;
    const int world = m->world, rank0 = m->ranks[0];
    const std::vector<int> devices = m->devices;
    int64_t stall_ms = 0;
#ifdef KZG_TEST_HOOKS
    if (const char *e = getenv("KZG_TEST_FORMATION_STALL_MS")) stall_ms = atoll(e);  // tests: a formation that takes this long
#endif
    const char *what = per_process ? "ncclCommInitRank" : "ncclCommInitAll";
    ncclResult_t e = ncclSuccess;
    KZG_DBG("%s: world %d, %d local GPU(s), deadline %lld ms ...", what, world, m->nlocal(), (long long)m->comm_timeout_ms);
    const bool back = run_bounded(
        m->comm_timeout_ms, st,
        [r, per_process, world, rank0, devices, stall_ms](BoundedState &s) -> ncclResult_t {
            if (stall_ms > 0) std::this_thread::sleep_for(std::chrono::milliseconds(stall_ms));
            if (per_process) {
                if (hipSetDevice(devices[0]) != hipSuccess) return ncclUnhandledCudaError;
                return r->CommInitRank(&s.comms[0], world, s.uid, rank0);
            }
            return r->CommInitAll(s.comms.data(), (int)devices.size(), devices.data());
        },
        &e, &m->t_init_ms);
    KZG_DBG("%s %s after %.1f ms", what, back ? (e == ncclSuccess ? "returned" : "FAILED") : "ABANDONED", m->t_init_ms);
    if (!back) {
        m->dead = true;
        const std::string msg = std::string(what) + " did not return within " + std::to_string(m->comm_timeout_ms) + " ms (" +
                                phase_report(m, r) + " ms; RCCL " + r->path + ")";
        rccl_mark_wedged(msg);
        return mfail(m, KZG_ERR_INTERNAL,
                     msg + ": communicator formation abandoned, this group is dead.  RCCL bootstraps over TCP on the interface it "
                           "picks: on one node set NCCL_SOCKET_IFNAME=lo (and NCCL_RAS_ENABLE=0, NCCL_IB_DISABLE=1) before the first "
                           "RCCL call; NCCL_DEBUG=INFO NCCL_DEBUG_FILE=<file> shows where it waits");
    }
    if (e != ncclSuccess) {
        m->dead = true;  // every rank fails the same way or the peers' own deadlines expire: the group cannot be used
        return mfail(m, KZG_ERR_HIP, std::string(what) + ": " + r->GetErrorString(e) + " (" + phase_report(m, r) + " ms)");
    }
    m->comms = st->comms;
    hipSetDevice(m->devices[0]);
    return KZG_OK;
}

extern "C" int kzg_mctx_create(const int *devices, int n, kzg_mctx **out) {
    if (!out || !devices || n < 1 || n > 64) return KZG_ERR_SHAPE;
    bool distinct = true;
#ifdef KZG_TEST_HOOKS
    distinct = !getenv("KZG_TEST_SHM_TRANSPORT");  // tests: several ranks of a one-process group on one GPU (test_transport.h)
#endif
    for (int i = 0; i < n && distinct; i++)
        for (int j = 0; j < i; j++)
            if (devices[i] == devices[j]) return KZG_ERR_SHAPE;
    kzg_mctx *m = new kzg_mctx();
    m->world = n;
    for (int i = 0; i < n; i++) {
        m->devices.push_back(devices[i]);
        m->ranks.push_back(i);
    }
    int rc = mctx_make_ctxs(m);
    if (rc == KZG_OK && n > 1) {
        Rccl *r = nullptr;
        rc = mctx_comm(m, &r);
    }
    if (rc != KZG_OK) {
        {
            std::lock_guard<std::mutex> lk(merr_mu());
            create_err() = m->err.empty() && !m->ctxs.empty() ? std::string(kzg_last_error(m->ctxs.back())) : m->err;
        }
        kzg_mctx_destroy(m);
        return rc;
    }
    create_err().clear();
    *out = m;
    return KZG_OK;
}

extern "C" int kzg_mctx_unique_id(void *id_out) {
    if (!id_out) return KZG_ERR_SHAPE;
    std::string err;
    create_err().clear();
    Rccl *r = rccl_load(&err);
    if (!r) {
        create_err() = err;
        return KZG_ERR_INTERNAL;
    }
    if (g_rccl_wedged) {
        std::lock_guard<std::mutex> lk(g_wedge_mu);
        create_err() = "an earlier RCCL communicator call of this process never returned (" + g_wedge_msg + ")";
        return KZG_ERR_INTERNAL;
    }
    auto st = std::make_shared<BoundedState>();
    ncclResult_t e = ncclSuccess;
    double ms = 0;
    const int64_t deadline = default_comm_timeout_ms();
    if (!run_bounded(deadline, st, [r](BoundedState &s) { return r->GetUniqueId(&s.uid); }, &e, &ms)) {
        create_err() = "ncclGetUniqueId did not return within " + std::to_string(deadline) + " ms (RCCL " + r->path + ", load=" +
                       std::to_string(r->load_ms) + " ms): abandoned; on one node set NCCL_SOCKET_IFNAME=lo before the first RCCL call";
        rccl_mark_wedged(create_err());
        KZG_DBG("ncclGetUniqueId ABANDONED after %.1f ms", ms);
        return KZG_ERR_INTERNAL;
    }
    r->uid_ms = ms;
    KZG_DBG("ncclGetUniqueId: %.1f ms", ms);
    if (e != ncclSuccess) {
        create_err() = std::string("ncclGetUniqueId: ") + r->GetErrorString(e);
        return KZG_ERR_HIP;
    }
    static_assert(sizeof(ncclUniqueId) == KZG_UNIQUE_ID_BYTES, "ncclUniqueId size");
    memcpy(id_out, &st->uid, sizeof st->uid);
    return KZG_OK;
}

extern "C" int kzg_mctx_create_rank(int device, int rank, int world, const void *unique_id, kzg_mctx **out) {
    if (!out || world < 1 || rank < 0 || rank >= world || (!unique_id && world > 1)) return KZG_ERR_SHAPE;
    kzg_mctx *m = new kzg_mctx();
    m->world = world;
    m->per_process = true;
    m->devices.push_back(device);
    m->ranks.push_back(rank);
    if (unique_id) memcpy(&m->uid, unique_id, sizeof m->uid);
    else memset(&m->uid, 0, sizeof m->uid);
    int rc = mctx_make_ctxs(m);
    if (rc == KZG_OK && world > 1) {
        Rccl *r = nullptr;
        rc = mctx_comm(m, &r);  // collective: every rank is inside kzg_mctx_create_rank
    }
    if (rc != KZG_OK) {
        {
            std::lock_guard<std::mutex> lk(merr_mu());
            create_err() = m->err.empty() && !m->ctxs.empty() ? std::string(kzg_last_error(m->ctxs.back())) : m->err;
        }
        kzg_mctx_destroy(m);
        return rc;
    }
    create_err().clear();
    *out = m;
    return KZG_OK;
}

// true when local GPU i's exchange stream went idle within `ms` (polled: never blocks on a stream an aborted collective still
// holds).  Collectives run on that stream only, so a dead group's context, shards and buffers are safe to release exactly when it is
// idle -- what other contexts of the device keep on the shared pool lanes is finite work and no concern of this group.
static bool exchange_idle_within(kzg_mctx *m, int i, int64_t ms) {
    if (i >= (int)m->xstream.size() || !m->xstream[i]) return true;
    const double t0 = now_ms();
    for (;;) {
        if (hipStreamQuery(m->xstream[i]) != hipErrorNotReady) return true;
        if (now_ms() - t0 > (double)ms) return false;
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
}

extern "C" void kzg_mctx_destroy(kzg_mctx *m) {
    if (!m) return;
    workers_stop(m);
    for (int i = 0; i < (int)m->ctxs.size(); i++) {
        hipSetDevice(m->devices[i]);
        bool leak = false;
        if (m->ctxs[i]) {
            // A live group drains its streams.  A DEAD one (exchange or formation timed out) may have a stream that an aborted
            // collective never leaves: synchronising on it -- or destroying it, or freeing the buffers its kernels address -- would
            // turn "destroy it and form a new one" into the hang the deadline was there to prevent.  Such a context is left behind.
            if (!m->dead) {
                kzg_sync(m->ctxs[i]);
                if (i < (int)m->xstream.size() && m->xstream[i]) hipStreamSynchronize(m->xstream[i]);
            } else {
                leak = g_rccl_wedged ? true : !exchange_idle_within(m, i, 2000);  // (wedged: a call abandoned inside RCCL / HIP may hold the streams' locks)
            }
        }
        if (i < (int)m->comms.size() && m->comms[i] && g_rccl && !m->dead && !g_rccl_wedged) {
            Rccl *r = g_rccl;
            ncclComm_t c = m->comms[i];
            auto st = std::make_shared<BoundedState>();
            ncclResult_t e = ncclSuccess;
            double ms = 0;
            const int dev = m->devices[i];
            if (!run_bounded(m->comm_timeout_ms, st, [r, c, dev](BoundedState &) { hipSetDevice(dev); return r->CommDestroy(c); }, &e, &ms))
                rccl_mark_wedged("ncclCommDestroy did not return within " + std::to_string(m->comm_timeout_ms) + " ms");
            m->t_destroy_ms = ms;
            KZG_DBG("ncclCommDestroy: %.1f ms", ms);
        }
        if (leak) {
            fprintf(stderr, "kzg: device group on GPU %d destroyed while a stream is still held by an aborted collective: its context and "
                            "exchange buffers are left behind (not freed)\n", m->devices[i]);
            continue;
        }
        if (m->d_part[i]) hipFree(m->d_part[i]);
        if (m->d_gath[i]) hipFree(m->d_gath[i]);
        if (m->d_quot[i]) hipFree(m->d_quot[i]);
        if (m->h_status[i]) hipHostFree(m->h_status[i]);
        if (i < (int)m->d_stat.size() && m->d_stat[i]) hipFree(m->d_stat[i]);
        if (i < (int)m->xstream.size()) {
            if (m->x_ready[i]) hipEventDestroy(m->x_ready[i]);
            if (m->x_done[i]) hipEventDestroy(m->x_done[i]);
            if (m->xstream[i]) hipStreamDestroy(m->xstream[i]);
        }
        if (m->ctxs[i]) kzg_ctx_destroy(m->ctxs[i]);
    }
    delete m;
}

static std::mutex g_merr_mu;  // m->err is written under it (mfail) and copied out under it
static std::mutex &merr_mu() { return g_merr_mu; }

extern "C" const char *kzg_mctx_last_error(kzg_mctx *m) {
    // copied into a per-thread buffer, as kzg_last_error does: another thread failing on the group cannot invalidate the pointer
    if (!m) return "null group";
    static thread_local std::string tl;
    std::lock_guard<std::mutex> lk(g_merr_mu);
    tl = m->err;
    return tl.c_str();
}

// which RCCL the group runs on: "rccl=<file> version=<n> hip=<runtime file>" (loads RCCL if that has not happened yet)
extern "C" int kzg_mctx_info(kzg_mctx *m, char *buf, size_t buflen) {
    if (!m || !buf || !buflen) return KZG_ERR_SHAPE;
    std::string err;
    Rccl *r = rccl_load(&err);
    if (!r) {
        std::lock_guard<std::mutex> lk(g_merr_mu);
        m->err = err;
        return KZG_ERR_INTERNAL;
    }
    // formation_ms: what forming this group's communicator cost so far (RCCL load + unique id + ncclCommInit*), -1 before a
    // communicator exists; the phases follow (ms; -1 = has not happened)
    std::lock_guard<std::mutex> lk(m->mu);
    const double form = m->t_init_ms < 0 ? -1.0 : r->load_ms + (r->uid_ms > 0 ? r->uid_ms : 0.0) + m->t_init_ms;
    snprintf(buf, buflen, "rccl=%s version=%d hip=%s world=%d local=%d mode=%s formation_ms=%.1f phases_ms=[%s] comm_timeout_ms=%lld "
                          "gather_timeout_ms=%lld dead=%d", r->path.c_str(), r->version, r->hip_path.c_str(), m->world, m->nlocal(),
             m->per_process ? "process-per-gpu" : "one-process", form, phase_report(m, r).c_str(), (long long)m->comm_timeout_ms,
             (long long)m->gather_timeout_ms, m->dead ? 1 : 0);
    for (int i = 0; i < m->nlocal(); i++) {  // each local context's pipeline plan (narrowed_from != none: the queue pool is short)
        char ci[256];
        const size_t used = strlen(buf);
        if (kzg_ctx_info(m->ctxs[i], ci, sizeof ci) == KZG_OK && used + strlen(ci) + 16 < buflen) snprintf(buf + used, buflen - used, " ctx%d=[%s]", i, ci);
    }
    return KZG_OK;
}
extern "C" int kzg_mctx_world(const kzg_mctx *m) { return m ? m->world : 0; }
extern "C" int kzg_mctx_local_count(const kzg_mctx *m) { return m ? m->nlocal() : 0; }
extern "C" int kzg_mctx_rank(const kzg_mctx *m, int i) { return (m && i >= 0 && i < m->nlocal()) ? m->ranks[i] : -1; }
extern "C" kzg_ctx *kzg_mctx_ctx(kzg_mctx *m, int i) { return (m && i >= 0 && i < m->nlocal()) ? m->ctxs[i] : nullptr; }

extern "C" int kzg_mctx_set_option(kzg_mctx *m, const char *key, int64_t value) {
    if (!m || !key) return KZG_ERR_SHAPE;
    std::lock_guard<std::mutex> lk(m->mu);
    if (std::string(key) == "always_gather") {
        m->always_gather = value != 0;
        return KZG_OK;
    }
    if (std::string(key) == "gather_timeout_ms") {
        if (value < 0) return mfail(m, KZG_ERR_SHAPE, "gather_timeout_ms must be >= 0 (0 = wait for ever)");
        m->gather_timeout_ms = value;
        return KZG_OK;
    }
    if (std::string(key) == "comm_timeout_ms") {
        if (value < 0) return mfail(m, KZG_ERR_SHAPE, "comm_timeout_ms must be >= 0 (0 = wait for ever)");
        m->comm_timeout_ms = value;
        return KZG_OK;
    }
    for (int i = 0; i < m->nlocal(); i++) {
        int rc = kzg_ctx_set_option(m->ctxs[i], key, value);
        if (rc != KZG_OK) return mfail_ctx(m, i, rc);
    }
    return KZG_OK;
}

// ---------------------------------------------------------------------------------------------
// sharded SRS
// ---------------------------------------------------------------------------------------------
static bool exchange_idle_within(kzg_mctx *m, int i, int64_t ms);
template <class F>
static int msrs_build(kzg_mctx *m, size_t n, kzg_msrs **out, F make_shard) {
    kzg_msrs *s = new kzg_msrs();
    s->n = n;
    const int L = m->nlocal();
    s->shards.assign(L, nullptr);
    s->first.assign(L, 0);
    s->len.assign(L, 0);
    std::vector<int> rcs(L, KZG_OK);
    auto work = [&](int i) {
        size_t lo = 0, hi = 0;
        kzg_shard_range(n, m->ranks[i], m->world, &lo, &hi);
        s->first[i] = lo;
        s->len[i] = hi - lo;
        rcs[i] = make_shard(i, lo, hi - lo, &s->shards[i]);
    };
    if (L == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int i = 0; i < L; i++) th.emplace_back(work, i);
        for (auto &t : th) t.join();
    }
    for (int i = 0; i < L; i++)
        if (rcs[i] != KZG_OK) {
            int rc = mfail_ctx(m, i, rcs[i]);
            kzg_msrs_free(m, s);
            return rc;
        }
    *out = s;
    return KZG_OK;
}

extern "C" int kzg_srs_setup_g1_sharded(kzg_mctx *m, const void *sec, int sfmt, size_t n, kzg_msrs **out) {
    if (!m || !sec || !out) return KZG_ERR_SHAPE;
    std::lock_guard<std::mutex> lk(m->mu);
    return msrs_build(m, n, out, [&](int i, size_t lo, size_t len, kzg_srs **shard) {
        return kzg_srs_setup_g1_shard(m->ctxs[i], sec, sfmt, lo, len, shard);
    });
}

extern "C" int kzg_srs_upload_g1_sharded(kzg_mctx *m, const void *pts, size_t n, int pfmt, kzg_msrs **out) {
    if (!m || !out || (!pts && n)) return KZG_ERR_SHAPE;
    std::lock_guard<std::mutex> lk(m->mu);
    size_t psz = point_format_bytes(pfmt);
    if (!psz) return mfail(m, KZG_ERR_SHAPE, "unknown G1 point format");
    return msrs_build(m, n, out, [&](int i, size_t lo, size_t len, kzg_srs **shard) {
        return kzg_srs_upload_g1(m->ctxs[i], (const uint8_t *)pts + lo * psz, len, pfmt, shard);
    });
}

extern "C" size_t kzg_msrs_len(const kzg_msrs *s) { return s ? s->n : 0; }

extern "C" const kzg_srs *kzg_msrs_shard(const kzg_msrs *s, int i, size_t *first) {
    if (!s || i < 0 || i >= (int)s->shards.size()) return nullptr;
    if (first) *first = s->first[i];
    return s->shards[i];
}

extern "C" void kzg_msrs_free(kzg_mctx *m, kzg_msrs *s) {
    if (!s) return;
    for (size_t i = 0; i < s->shards.size(); i++) {
        if (!s->shards[i]) continue;
        kzg_ctx *c = m && i < m->ctxs.size() ? m->ctxs[i] : nullptr;
        // hipFree synchronises the device: on a DEAD group whose stream an aborted collective still holds that is the hang the
        // deadlines exist to prevent -- the shard's memory is left behind with the context (kzg_mctx_destroy)
        if (m && m->dead && c && (g_rccl_wedged || !exchange_idle_within(m, (int)i, 0))) {
            fprintf(stderr, "kzg: SRS shard of a dead device group on GPU %d is left behind (its stream is still held by an aborted "
                            "collective; freeing would wait for it)\n", m->devices[i]);
            continue;
        }
        kzg_srs_free(c, s->shards[i]);
    }
    delete s;
}

// ---------------------------------------------------------------------------------------------
// sharded commit
// ---------------------------------------------------------------------------------------------
// What a rank sends: `batch` 144-byte partials followed by one status slot of the same size (word 0 = the kzg_status of its
// local phase; a whole point slot, so that the gathered records stay a whole number of points apart).  Every rank ALWAYS enters the exchange -- with its failure code when its local phase failed -- and every rank reads
// every status afterwards, so a failing rank can neither leave the others waiting inside ncclAllGather nor be the only one to
// report the failure: all ranks return the same error.
constexpr size_t STATUS_BYTES = 16, STATUS_ALL_OFF = 256;  // pinned staging: own status at 0, every rank's from STATUS_ALL_OFF
static size_t record_bytes(size_t batch) { return (batch + 1) * PARTIAL_BYTES; }

static int mctx_buffers_grow(kzg_mctx *m, size_t batch) {
    for (int i = 0; i < m->nlocal(); i++) {
        if (m->cap_points[i] >= batch && m->d_stat[i]) continue;
        if (hipSetDevice(m->devices[i]) != hipSuccess) return mfail(m, KZG_ERR_HIP, "hipSetDevice");
        kzg_sync(m->ctxs[i]);
        if (!m->h_status[i] && hipHostMalloc(&m->h_status[i], STATUS_ALL_OFF + STATUS_BYTES * (size_t)m->world, hipHostMallocDefault) != hipSuccess)
            return mfail(m, KZG_ERR_ALLOC, "hipHostMalloc(status words)");
        if (!m->d_stat[i] && hipMalloc(&m->d_stat[i], STATUS_BYTES * ((size_t)m->world + 1)) != hipSuccess)
            return mfail(m, KZG_ERR_ALLOC, "hipMalloc(status agreement buffer)");
        if (m->cap_points[i] >= batch) continue;
#ifdef KZG_TEST_HOOKS
        if (m->inject_alloc_fail && i == 0) {
            m->inject_alloc_fail = 0;
            return mfail(m, KZG_ERR_ALLOC, "hipMalloc(partial-point exchange buffers): injected failure (test hook)");
        }
#endif
        size_t cap = batch < 64 ? 64 : batch;
        void *np = nullptr, *ng = nullptr;
        if (hipMalloc(&np, record_bytes(cap)) != hipSuccess || hipMalloc(&ng, record_bytes(cap) * (size_t)m->world) != hipSuccess) {
            if (np) hipFree(np);  // the smaller buffers stay usable
            return mfail(m, KZG_ERR_ALLOC, "hipMalloc(partial-point exchange buffers)");
        }
        if (m->d_part[i]) hipFree(m->d_part[i]);
        if (m->d_gath[i]) hipFree(m->d_gath[i]);
        m->d_part[i] = np;
        m->d_gath[i] = ng;
        m->cap_points[i] = cap;
    }
    return KZG_OK;
}

// Wait for local GPU i's exchange stream, with the group's deadline.  KZG_OK, or the group is aborted and dead.
static int mctx_wait(kzg_mctx *m, Rccl *r, int i, const char *what) {
    hipSetDevice(m->devices[i]);
    // completion = the event recorded on the group's exchange stream right behind the collective and its status download; nothing
    // but the exchange ever runs on that stream
    const double t_w0 = now_ms();
    if (hipEventRecord(m->x_done[i], m->xstream[i]) != hipSuccess) return mfail(m, KZG_ERR_HIP, std::string(what) + ": hipEventRecord");
    // The local phase in front of the exchange (x_ready: recorded on the context's lane 0, a stream of the device's shared pool) is
    // this rank's own, finite work -- possibly queued behind another context's kernels on that lane: it is waited for WITHOUT a
    // deadline.  The deadline below is the collective's alone.
    if (hipEventSynchronize(m->x_ready[i]) != hipSuccess) return mfail(m, KZG_ERR_HIP, std::string(what) + ": the local phase failed");
    const double t_w1 = now_ms();
    if (m->gather_timeout_ms <= 0) {
        if (hipEventSynchronize(m->x_done[i]) != hipSuccess) return mfail(m, KZG_ERR_HIP, std::string(what) + " failed");
        return KZG_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        hipError_t e = hipEventQuery(m->x_done[i]);
        if (e == hipSuccess) {
            KZG_DBG("%s: local phase %.2f ms, collective %.2f ms", what, t_w1 - t_w0, now_ms() - t_w1);
            return KZG_OK;
        }
        if (e != hipErrorNotReady) return mfail(m, KZG_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
        const int64_t us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
        if (us > m->gather_timeout_ms * 1000) break;
        if (us > 2000) std::this_thread::sleep_for(std::chrono::microseconds(100));  // a healthy exchange is over long before
    }
    // a peer never arrived: abort the communicators so that the collective kernels leave the streams, and retire the group.
    // ncclCommAbort itself waits for the stream's kernels (measured: behind a 12 s spin kernel it returned after 12 s), so it runs
    // bounded like every other communicator call; abort + drain together get 5 s, a stream that still hangs is left behind.
    m->dead = true;
    const double t_ab = now_ms();
    bool abort_back = true;
    KZG_DBG("%s: deadline of %lld ms passed, aborting the communicators", what, (long long)m->gather_timeout_ms);
    if (r && r->CommAbort && !g_rccl_wedged) {
        auto st = std::make_shared<BoundedState>();
        st->comms = m->comms;
        ncclResult_t e = ncclSuccess;
        double ms = 0;
        Rccl *rr = r;
        abort_back = run_bounded(5000, st, [rr](BoundedState &s) {
                for (auto &c : s.comms)
                    if (c) rr->CommAbort(c);
                return ncclSuccess;
            }, &e, &ms);
        if (!abort_back) rccl_mark_wedged("ncclCommAbort did not return within 5000 ms");
        KZG_DBG("ncclCommAbort %s after %.1f ms", abort_back ? "returned" : "ABANDONED", ms);
    }
    m->comms.clear();
    // (an abandoned ncclCommAbort may sit inside the HIP runtime holding the stream's lock: no further call on those streams)
    for (int j = 0; j < m->nlocal() && abort_back; j++) {
        hipSetDevice(m->devices[j]);
        while (hipStreamQuery(m->xstream[j]) == hipErrorNotReady && now_ms() - t_ab < 5000.0)
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    KZG_DBG("%s: returning the time-out after %.1f ms of abort + drain", what, now_ms() - t_ab);
    return mfail(m, KZG_ERR_INTERNAL, std::string(what) + " did not complete within " + std::to_string(m->gather_timeout_ms) +
                                          " ms: a peer is dead or stalled.  The communicators were aborted and this group is dead "
                                          "(every further call fails; destroy it and form a new one)");
}

#ifdef KZG_TEST_HOOKS
__global__ void k_test_stall(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
    }
}
#endif

// Every rank reports `code` (its own resource status before the data exchange); all ranks return the first failing rank's code.
// Uses only what exists since the group was formed (d_stat, h_status, the communicator), so a rank that could not grow its
// exchange buffers can still tell the others -- nobody is left waiting inside the data all-gather of a call one rank abandoned.
static int mctx_agree(kzg_mctx *m, int code, const char *what) {
    const bool exchange = m->world > 1 || m->always_gather;
    if (!exchange || !m->per_process) return code;  // every rank is in this process: the failure is known to all of them
    Rccl *r = nullptr;
    int crc = mctx_comm(m, &r);
    if (crc != KZG_OK) return crc;  // no communicator: nothing can be exchanged (the peers' wait has its deadline)
    hipStream_t st = m->xstream[0];  // the group's own exchange stream: the agreement depends on nothing the lanes hold
    hipSetDevice(m->devices[0]);
    int32_t *hs = (int32_t *)m->h_status[0];
    hs[0] = code;
    hs[1] = m->ranks[0];
    hs[2] = hs[3] = 0;
    uint8_t *mine = (uint8_t *)m->d_stat[0], *all = mine + STATUS_BYTES;
    int32_t *hall = (int32_t *)((uint8_t *)m->h_status[0] + STATUS_ALL_OFF);
    if (hipMemcpyAsync(mine, hs, STATUS_BYTES, hipMemcpyHostToDevice, st) != hipSuccess) return mfail(m, KZG_ERR_HIP, "status upload");
    const bool first = m->t_first_ms < 0;
    const double t_x0 = now_ms();
    ncclResult_t e = r->AllGather(mine, all, STATUS_BYTES, ncclUint8, m->comms[0], st);
    if (e != ncclSuccess) return mfail(m, KZG_ERR_HIP, std::string("ncclAllGather(status): ") + r->GetErrorString(e));
    if (hipMemcpyAsync(hall, all, STATUS_BYTES * (size_t)m->world, hipMemcpyDeviceToHost, st) != hipSuccess)
        return mfail(m, KZG_ERR_HIP, "status download");
    KZG_TRY(mctx_wait(m, r, 0, "status agreement"));
    if (first) {
        m->t_first_ms = now_ms() - t_x0;
        KZG_DBG("first exchange (status agreement): %.1f ms", m->t_first_ms);
    }
    for (int rk = 0; rk < m->world; rk++)
        if (hall[4 * rk] != KZG_OK) {
            if (rk == m->ranks[0]) return code;  // this rank's own failure: its message is already in place
            return mfail(m, hall[4 * rk], "rank " + std::to_string(rk) + " could not prepare " + what + " (status " +
                                              std::to_string(hall[4 * rk]) + "); every rank returns this error");
        }
    return KZG_OK;
}

// Buffers for a call with `batch` partials per rank.  Whether the ranks must agree is decided by what they last AGREED on
// (agreed_points: the same on every rank -- same calls, same outcomes), not by a rank's own capacity: after a growth that failed on
// one rank only, the others hold larger buffers than that rank, and on the next call all of them must still enter the agreement the
// failed rank enters (tests/test_gpu_mgpu_world.py runs exactly that at world 2..8).  Ordinary calls pay nothing.
static int mctx_buffers(kzg_mctx *m, size_t batch) {
    if (m->dead) return mfail(m, KZG_ERR_INTERNAL, "this device group is dead (an earlier exchange timed out and was aborted): destroy it and form a new one");
    if (batch <= m->agreed_points) return KZG_OK;
    int rc = mctx_buffers_grow(m, batch);  // (a no-op on a rank that already grew in a round another rank failed)
    rc = mctx_agree(m, rc, "its exchange buffers");
    if (rc == KZG_OK) m->agreed_points = batch;
    return rc;
}

// run f(i) for every local GPU (on the group's persistent worker threads when there are several); rcs[i] = its status
template <class F>
static void for_each_local(kzg_mctx *m, std::vector<int> &rcs, F f) {
    const int L = m->nlocal();
    rcs.assign(L, KZG_OK);
    if (L == 1 || m->workers.empty()) {
        for (int i = 0; i < L; i++) rcs[i] = f(i);
    } else {
        for (int i = 0; i < L; i++) {
            kzg_mctx::Worker *w = m->workers[i];
            std::lock_guard<std::mutex> lk(w->mu);
            w->job = [&f, i]() { return f(i); };
            w->has_job = true;
            w->done = false;
            w->cv.notify_all();
        }
        for (int i = 0; i < L; i++) {
            kzg_mctx::Worker *w = m->workers[i];
            std::unique_lock<std::mutex> lk(w->mu);
            w->cv.wait(lk, [&] { return w->done; });
            rcs[i] = w->rc;
        }
    }
#ifdef KZG_TEST_HOOKS
    if (m->inject_fail) {
        rcs[0] = fail(m->ctxs[0], m->inject_fail, "injected local failure (test hook)");
        m->inject_fail = 0;
    }
#endif
}

// Stage 2 of every sharded operation: d_part[i] holds `batch` Jacobian partials on every local GPU whose local phase succeeded
// (local_rc[i] == KZG_OK).  One all-gather of partials + status, then the group's first local GPU adds the `world` partials of
// each polynomial and writes `batch` points in ofmt to `out` (host).  Returns the first failing rank's status on every rank.
static int mctx_combine(kzg_mctx *m, size_t batch, const std::vector<int> &local_rc, void *out, int ofmt) {
    const bool exchange = m->world > 1 || m->always_gather;
    int first_bad = -1;
    for (int i = 0; i < m->nlocal(); i++)
        if (local_rc[i] != KZG_OK && first_bad < 0) first_bad = i;
    if (!exchange || !m->per_process) {
        // every rank is in this process: a local failure is known to all of them already -- no collective to keep in step
        if (first_bad >= 0) return mfail_ctx(m, first_bad, local_rc[first_bad]);
    }
    const void *src = m->d_part[0];
    size_t count = 1, gstride = 1, istride = batch;
    if (exchange) {
        Rccl *r = nullptr;
        KZG_TRY(mctx_comm(m, &r));  // no communicator, nothing to enter (the peers' waits have their deadline)
        const size_t rec = record_bytes(batch);
        int upload_rc = KZG_OK;
        for (int i = 0; i < m->nlocal(); i++) {  // the status block rides behind the partials, in stream order
            hipSetDevice(m->devices[i]);
            hipStream_t st = m->ctxs[i]->lanes[0].stream;
            int32_t *hs = (int32_t *)m->h_status[i];
            hs[0] = local_rc[i];
            hs[1] = m->ranks[i];
            hs[2] = hs[3] = 0;
            uint8_t *slot = (uint8_t *)m->d_part[i] + batch * PARTIAL_BYTES;
            // the slot says "failed" until this rank's real status has landed in it: if the upload itself fails the rank still
            // enters the all-gather below and its peers read a failure, not stale zeros
            if (hipMemsetAsync(slot, 0xff, STATUS_BYTES, st) != hipSuccess ||
                hipMemcpyAsync(slot, hs, STATUS_BYTES, hipMemcpyHostToDevice, st) != hipSuccess)
                upload_rc = mfail(m, KZG_ERR_HIP, "status upload");
            // the exchange stream takes over behind everything lane 0 holds for this call: the partials and the status slot
            if (hipEventRecord(m->x_ready[i], st) != hipSuccess || hipStreamWaitEvent(m->xstream[i], m->x_ready[i], 0) != hipSuccess)
                upload_rc = mfail(m, KZG_ERR_HIP, "ordering the exchange stream behind the local phase");
#ifdef KZG_TEST_HOOKS
            if (m->inject_stall_ms && i == 0) {  // a late peer: the exchange sits behind a spin kernel
                hipLaunchKernelGGL(k_test_stall, dim3(1), dim3(1), 0, m->xstream[i], (unsigned long long)m->inject_stall_ms * 100000ull);  // wall_clock64: 100 MHz
                m->inject_stall_ms = 0;
            }
#endif
        }
        const bool first = m->t_first_ms < 0;
        if (first)  // the first exchange is timed on its own (RCCL loads its kernels here): the local phases drain first
            for (int i = 0; i < m->nlocal(); i++) kzg_sync(m->ctxs[i]);
        const double t_x0 = now_ms();
        KZG_NCCL(m, r, r->GroupStart());
        for (int i = 0; i < m->nlocal(); i++) {
            ncclResult_t e = r->AllGather(m->d_part[i], m->d_gath[i], rec, ncclUint8, m->comms[i], m->xstream[i]);
            if (e != ncclSuccess) {
                r->GroupEnd();
                return mfail(m, KZG_ERR_HIP, std::string("ncclAllGather: ") + r->GetErrorString(e));
            }
        }
        KZG_NCCL(m, r, r->GroupEnd());
        // every local GPU's copy must be complete before its buffers are re-used by the next call; the statuses of all ranks
        // come back with local GPU 0's
        hipSetDevice(m->devices[0]);
        int32_t *hall = (int32_t *)((uint8_t *)m->h_status[0] + STATUS_ALL_OFF);
        if (hipMemcpy2DAsync(hall, STATUS_BYTES, (const uint8_t *)m->d_gath[0] + batch * PARTIAL_BYTES, rec, STATUS_BYTES,
                             (size_t)m->world, hipMemcpyDeviceToHost, m->xstream[0]) != hipSuccess)
            return mfail(m, KZG_ERR_HIP, "status download");
        for (int i = 0; i < m->nlocal(); i++) KZG_TRY(mctx_wait(m, r, i, "the all-gather of the partial points"));
        if (first) {
            m->t_first_ms = now_ms() - t_x0;
            KZG_DBG("first exchange (all-gather of the partials): %.1f ms", m->t_first_ms);
        }
        if (upload_rc != KZG_OK) return upload_rc;
        for (int rk = 0; rk < m->world; rk++)
            if (hall[4 * rk] != KZG_OK) {
                if (first_bad >= 0 && m->ranks[first_bad] == rk) return mfail_ctx(m, first_bad, local_rc[first_bad]);
                // (a slot still holding the 0xff fill reads as -1 = KZG_ERR_HIP: that rank's status upload failed)
                return mfail(m, hall[4 * rk], "rank " + std::to_string(rk) + " failed in its local phase (status " +
                                                  std::to_string(hall[4 * rk]) + "); every rank returns this error");
            }
        src = m->d_gath[0];  // [world] records of [batch partials, status slot]; the sum below runs on the same stream
        count = (size_t)m->world;
        gstride = 1;
        istride = batch + 1;
    }
    int rc = g1_sum_batch_strided(m->ctxs[0], src, count, batch, gstride, istride, KZG_G1_JACOBIAN_MONT_144, KZG_IN_DEVICE, out, ofmt,
                                  POINTS_TRUSTED);  // the group's own partial sums
    if (rc != KZG_OK) return mfail_ctx(m, 0, rc);
    return KZG_OK;
}

// the part of [first, first + len) below n
static size_t clip_len(size_t first, size_t len, size_t n) { return n <= first ? 0 : (n - first < len ? n - first : len); }

extern "C" int kzg_commit_coeff_sharded_batch(kzg_mctx *m, const kzg_msrs *srs, const void *coeffs, size_t n, size_t batch,
                                              int sfmt, int flags, void *out, int ofmt) {
    if (!m || !srs || !out || (!coeffs && n && batch)) return KZG_ERR_SHAPE;
    std::lock_guard<std::mutex> lk(m->mu);
    if ((int)srs->shards.size() != m->nlocal()) return mfail(m, KZG_ERR_SHAPE, "SRS belongs to another group");
    if (n > srs->n) return mfail(m, KZG_ERR_SHAPE, "polynomial longer than the SRS (reference: slice index panic)");
    if (!point_format_bytes(ofmt)) return mfail(m, KZG_ERR_SHAPE, "unknown G1 output format");
    if (sfmt != KZG_FR_MONT_LE_32 && sfmt != KZG_FR_CANONICAL_LE_32) return mfail(m, KZG_ERR_SHAPE, "unknown scalar format");
    if (flags & KZG_OUT_DEVICE) return mfail(m, KZG_ERR_SHAPE, "sharded commit writes its result to host memory");
    if (batch == 0) return KZG_OK;
    if (batch > (1u << 20)) return mfail(m, KZG_ERR_SHAPE, "batch <= 2^20");
    KZG_TRY(mctx_buffers(m, batch));
    const bool in_dev = (flags & KZG_IN_DEVICE) != 0;
    std::vector<int> rcs;
    for_each_local(m, rcs, [&](int i) {
        const size_t len = clip_len(srs->first[i], srs->len[i], n);  // this GPU's terms of an n-coefficient polynomial
        const void *sc;
        size_t stride;
        if (in_dev) {
            sc = ((const void *const *)coeffs)[i];
            stride = len * 32;
        } else {
            sc = (const uint8_t *)coeffs + srs->first[i] * 32;
            stride = n * 32;
        }
        return msm_batch_strided(m->ctxs[i], srs->shards[i], 0, sc, len, batch, stride, sfmt,
                                 (in_dev ? KZG_IN_DEVICE : 0) | KZG_OUT_DEVICE, m->d_part[i], KZG_G1_JACOBIAN_MONT_144);
    });
    return mctx_combine(m, batch, rcs, out, ofmt);
}

extern "C" int kzg_commit_coeff_sharded(kzg_mctx *m, const kzg_msrs *srs, const void *coeffs, size_t n, int sfmt, int flags,
                                        void *out, int ofmt) {
    return kzg_commit_coeff_sharded_batch(m, srs, coeffs, n, 1, sfmt, flags, out, ofmt);
}

// the coefficient vector local GPU i reads: the host's (every GPU stages it) or, with KZG_IN_DEVICE, GPU i's resident copy
static const void *whole_poly(const void *coeffs, int flags, int i) {
    return (flags & KZG_IN_DEVICE) ? ((const void *const *)coeffs)[i] : coeffs;
}

// the replicated quotient polynomial of create_witness (both forms): n scalars on every local GPU; the ranks agree when it grows
static int mctx_quotient_buffers(kzg_mctx *m, size_t n) {
    if (n <= m->agreed_quot) return KZG_OK;  // (what the ranks agreed on, not this rank's own capacity: see mctx_buffers)
    int rc = KZG_OK;
    for (int i = 0; i < m->nlocal() && rc == KZG_OK; i++) {
        if (m->cap_quot[i] >= n) continue;
        hipSetDevice(m->devices[i]);
        kzg_sync(m->ctxs[i]);
        if (m->d_quot[i]) hipFree(m->d_quot[i]);
        m->d_quot[i] = nullptr;
        m->cap_quot[i] = 0;
        if (hipMalloc(&m->d_quot[i], n * 32) != hipSuccess) rc = mfail(m, KZG_ERR_ALLOC, "hipMalloc(quotient)");
        else m->cap_quot[i] = n;
    }
    rc = mctx_agree(m, rc, "its quotient buffer");
    m->agreed_quot = rc == KZG_OK ? n : 0;  // after a failure (the failing rank has no buffer left) the next call agrees again
    return rc;
}

extern "C" int kzg_witness_coeff_sharded(kzg_mctx *m, const kzg_msrs *srs, const void *coeffs, size_t n, const void *x,
                                         const void *y, int sfmt, int flags, void *out, int ofmt) {
    // KZGProver::create_witness (src/coeff_form.rs:66-81): q = (p - y)/(X - x) has n - 1 coefficients; rank r reduces
    // q[lo_r, hi_r) against its shard.  The O(n) quotient scan is replicated on every GPU.
    if (!m || !srs || !coeffs || !x || !y || !out || n == 0) return KZG_ERR_SHAPE;
    std::lock_guard<std::mutex> lk(m->mu);
    if ((int)srs->shards.size() != m->nlocal()) return mfail(m, KZG_ERR_SHAPE, "SRS belongs to another group");
    if (n - 1 > srs->n) return mfail(m, KZG_ERR_SHAPE, "quotient longer than the SRS (reference: slice index panic)");
    if (!point_format_bytes(ofmt)) return mfail(m, KZG_ERR_SHAPE, "unknown G1 output format");
    if (flags & KZG_OUT_DEVICE) return mfail(m, KZG_ERR_SHAPE, "sharded create_witness writes its result to host memory");
    KZG_TRY(mctx_buffers(m, 1));
    KZG_TRY(mctx_quotient_buffers(m, n));
    const int in_dev = flags & KZG_IN_DEVICE;
    std::vector<int> off_poly(m->nlocal(), 0), rcs;
    for_each_local(m, rcs, [&](int i) {
        int q = kzg_quotient_linear(m->ctxs[i], whole_poly(coeffs, flags, i), n, x, y, sfmt, in_dev | KZG_OUT_DEVICE, m->d_quot[i]);
        if (q == KZG_ERR_POINT_NOT_ON_POLY) {
            off_poly[i] = 1;  // the reference fails after the division (every rank sees the same remainder)
            q = KZG_OK;
        }
        if (q != KZG_OK) return q;
        const size_t len = clip_len(srs->first[i], srs->len[i], n - 1);
        return msm_batch_strided(m->ctxs[i], srs->shards[i], 0, (const uint8_t *)m->d_quot[i] + srs->first[i] * 32, len, 1,
                                 len * 32, sfmt, KZG_IN_DEVICE | KZG_OUT_DEVICE, m->d_part[i], KZG_G1_JACOBIAN_MONT_144);
    });
    KZG_TRY(mctx_combine(m, 1, rcs, out, ofmt));
    for (int i = 0; i < m->nlocal(); i++)
        if (off_poly[i]) return mfail(m, KZG_ERR_POINT_NOT_ON_POLY, "point not on polynomial!");
    return KZG_OK;
}

extern "C" int kzg_witness_coeff_batched_sharded(kzg_mctx *m, const kzg_msrs *srs, const void *coeffs, size_t n, const void *xs,
                                                 const void *ys, size_t k, int sfmt, int flags, void *out_w, int ofmt,
                                                 void *out_r, size_t *out_r_len) {
    // KZGProver::create_witness_batched (src/coeff_form.rs:83-111) over the group: every rank computes the interpolant I and the
    // quotient (p - I)/Z on its GPU (replicated: NTTs do not shard) and reduces its own slice of the quotient
    // against its SRS shard; one exchange of 144-byte partials.  The reference's two failures (duplicate opening points -> the
    // invert().unwrap() panic, a point off the polynomial) come out of the replicated part, identically on every rank.
    if (!m || !srs || !coeffs || !xs || !ys || !out_w || !out_r || !out_r_len || n == 0) return KZG_ERR_SHAPE;
    std::lock_guard<std::mutex> lk(m->mu);
    if ((int)srs->shards.size() != m->nlocal()) return mfail(m, KZG_ERR_SHAPE, "SRS belongs to another group");
    if (!point_format_bytes(ofmt)) return mfail(m, KZG_ERR_SHAPE, "unknown G1 output format");
    if (flags & KZG_OUT_DEVICE) return mfail(m, KZG_ERR_SHAPE, "sharded create_witness_batched writes its result to host memory");
    if (k == 0 || k > 16384) return mfail(m, KZG_ERR_SHAPE, "1 <= opening points <= 16384");
    KZG_TRY(mctx_buffers(m, 1));
    const size_t rl = k == 1 ? 2 : k;
    std::vector<std::vector<uint8_t>> rbuf(m->nlocal(), std::vector<uint8_t>(rl * 32));
    std::vector<size_t> rlen(m->nlocal(), 0);
    std::vector<int> soft(m->nlocal(), KZG_OK), rcs;  // the reference's own errors: identical on every rank, reported after the exchange
    for_each_local(m, rcs, [&](int i) {
        WitnessSink sink{srs->shards[i], srs->first[i], srs->len[i], srs->n, m->d_part[i]};
        int rc = witness_coeff_batched_run(m->ctxs[i], sink, whole_poly(coeffs, flags, i), n, xs, ys, k, sfmt, flags & KZG_IN_DEVICE,
                                           nullptr, ofmt, rbuf[i].data(), &rlen[i]);
        if (rc == KZG_ERR_POINT_NOT_ON_POLY) {  // found after the partial was written, like the reference (division, then the test)
            soft[i] = rc;
            rc = KZG_OK;
        }
        return rc;
    });
    KZG_TRY(mctx_combine(m, 1, rcs, out_w, ofmt));
    for (int i = 0; i < m->nlocal(); i++)
        if (soft[i]) return mfail_ctx(m, i, soft[i]);
    memcpy(out_r, rbuf[0].data(), rlen[0] * 32);
    *out_r_len = rlen[0];
    return KZG_OK;
}

extern "C" int kzg_witness_eval_sharded(kzg_mctx *m, const kzg_msrs *lagrange, const void *evals, size_t d, size_t index, int sfmt,
                                        int flags, void *out, int ofmt) {
    // KZGProverEvalForm::create_witness (src/eval_form.rs:124-140) over the group: every rank computes div_by_omega_i (:58-84) of
    // (evals - evals[index]) on its GPU -- the closed form of poly.hip, replicated (an O(d) pass) -- and reduces
    // q[lo_r, hi_r) against its shard of the Lagrange-basis SRS; one exchange of 144-byte partials.
    if (!m || !lagrange || !evals || !out) return KZG_ERR_SHAPE;
    std::lock_guard<std::mutex> lk(m->mu);
    if ((int)lagrange->shards.size() != m->nlocal()) return mfail(m, KZG_ERR_SHAPE, "SRS belongs to another group");
    if (d == 0 || (d & (d - 1))) return mfail(m, KZG_ERR_SHAPE, "evaluation domain size must be a power of two");
    if (index >= d) return mfail(m, KZG_ERR_SHAPE, "evaluation index out of range (reference: index panic)");
    if (d > lagrange->n) return mfail(m, KZG_ERR_SHAPE, "evaluations longer than the Lagrange SRS (reference: slice panic)");
    if (!point_format_bytes(ofmt)) return mfail(m, KZG_ERR_SHAPE, "unknown G1 output format");
    if (sfmt != KZG_FR_MONT_LE_32 && sfmt != KZG_FR_CANONICAL_LE_32) return mfail(m, KZG_ERR_SHAPE, "unknown scalar format");
    if (flags & KZG_OUT_DEVICE) return mfail(m, KZG_ERR_SHAPE, "sharded create_witness writes its result to host memory");
    KZG_TRY(mctx_buffers(m, 1));
    KZG_TRY(mctx_quotient_buffers(m, d));
    const int in_dev = flags & KZG_IN_DEVICE;
    std::vector<int> rcs;
    for_each_local(m, rcs, [&](int i) {
        int q = kzg_quotient_eval(m->ctxs[i], whole_poly(evals, flags, i), d, index, sfmt, in_dev | KZG_OUT_DEVICE, m->d_quot[i]);
        if (q != KZG_OK) return q;
        const size_t len = clip_len(lagrange->first[i], lagrange->len[i], d);
        return msm_batch_strided(m->ctxs[i], lagrange->shards[i], 0, (const uint8_t *)m->d_quot[i] + lagrange->first[i] * 32, len, 1,
                                 len * 32, sfmt, KZG_IN_DEVICE | KZG_OUT_DEVICE, m->d_part[i], KZG_G1_JACOBIAN_MONT_144);
    });
    return mctx_combine(m, 1, rcs, out, ofmt);
}

#ifdef KZG_TEST_HOOKS
extern "C" int kzg_test_mctx_inject_alloc_failure(kzg_mctx *m) {
    if (!m) return KZG_ERR_SHAPE;
    std::lock_guard<std::mutex> lk(m->mu);
    m->inject_alloc_fail = 1;
    return KZG_OK;
}
extern "C" int kzg_test_mctx_inject_stall(kzg_mctx *m, int ms) {
    if (!m || ms < 0 || ms > 20000) return KZG_ERR_SHAPE;
    std::lock_guard<std::mutex> lk(m->mu);
    m->inject_stall_ms = ms;
    return KZG_OK;
}
extern "C" int kzg_test_mctx_inject_failure(kzg_mctx *m, int code) {
    if (!m) return KZG_ERR_SHAPE;
    std::lock_guard<std::mutex> lk(m->mu);
    m->inject_fail = code;
    return KZG_OK;
}
#endif
