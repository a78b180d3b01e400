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

#include <functional>
#include <thread>

#include "common.h"
#ifdef KZG_TEST_HOOKS
#include "../../include/kzg_mi355x_test.h"
#endif

namespace kzg {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string path;      // the file the adopted RCCL was loaded from
    std::string hip_path;  // the HIP runtime it is bound to
    int version = 0;
};

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
    const std::string mine = object_of((const void *)&hipStreamSynchronize);
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
    std::mutex mu;
    std::string err;
    // grow-only exchange buffers per local GPU: partials of this GPU, partials of every rank
    std::vector<void *> d_part, d_gath;
    std::vector<size_t> cap_points;
    std::vector<void *> d_quot;      // create_witness: the quotient polynomial on each GPU
    std::vector<size_t> cap_quot;
    std::vector<void *> h_status;    // pinned: the status words of all ranks after the exchange, per local GPU
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

static int mctx_make_ctxs(kzg_mctx *m) {
    // sized first: kzg_mctx_destroy walks these per context when a later device fails (devices = [0, 99])
    m->d_part.assign(m->nlocal(), nullptr);
    m->d_gath.assign(m->nlocal(), nullptr);
    m->cap_points.assign(m->nlocal(), 0);
    m->d_quot.assign(m->nlocal(), nullptr);
    m->cap_quot.assign(m->nlocal(), 0);
    m->h_status.assign(m->nlocal(), nullptr);
    for (int i = 0; i < m->nlocal(); i++) {
        kzg_ctx *c = nullptr;
        int rc = kzg_ctx_create(m->devices[i], &c);
        if (rc != KZG_OK) return rc;
        m->ctxs.push_back(c);
    }
    workers_start(m);
    return KZG_OK;
}

// the communicator is created when the first collective needs it (a group of one GPU never does unless asked to)
static int mctx_comm(kzg_mctx *m, Rccl **out) {
    std::string err;
    Rccl *r = rccl_load(&err);
    if (!r) return mfail(m, KZG_ERR_INTERNAL, err);
    *out = r;
    if (!m->comms.empty()) return KZG_OK;
    m->comms.assign(m->nlocal(), nullptr);
    if (m->per_process) {
        if (hipSetDevice(m->devices[0]) != hipSuccess) return mfail(m, KZG_ERR_NO_DEVICE, "hipSetDevice");
        ncclResult_t e = r->CommInitRank(&m->comms[0], m->world, m->uid, m->ranks[0]);
        if (e != ncclSuccess) {
            m->comms.clear();
            return mfail(m, KZG_ERR_HIP, std::string("ncclCommInitRank: ") + r->GetErrorString(e));
        }
    } else {
        ncclResult_t e = r->CommInitAll(m->comms.data(), m->nlocal(), m->devices.data());
        if (e != ncclSuccess) {
            m->comms.clear();
            return mfail(m, KZG_ERR_HIP, std::string("ncclCommInitAll: ") + r->GetErrorString(e));
        }
    }
    return KZG_OK;
}

extern "C" int kzg_mctx_create(const int *devices, int n, kzg_mctx **out) {
    if (!out || !devices || n < 1 || n > 64) return KZG_ERR_SHAPE;
    for (int i = 0; i < n; i++)
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
        kzg_mctx_destroy(m);
        return rc;
    }
    *out = m;
    return KZG_OK;
}

extern "C" int kzg_mctx_unique_id(void *id_out) {
    if (!id_out) return KZG_ERR_SHAPE;
    std::string err;
    Rccl *r = rccl_load(&err);
    if (!r) return KZG_ERR_INTERNAL;
    ncclUniqueId id;
    if (r->GetUniqueId(&id) != ncclSuccess) return KZG_ERR_HIP;
    static_assert(sizeof(ncclUniqueId) == KZG_UNIQUE_ID_BYTES, "ncclUniqueId size");
    memcpy(id_out, &id, sizeof id);
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
        kzg_mctx_destroy(m);
        return rc;
    }
    *out = m;
    return KZG_OK;
}

extern "C" void kzg_mctx_destroy(kzg_mctx *m) {
    if (!m) return;
    workers_stop(m);
    for (int i = 0; i < (int)m->ctxs.size(); i++) {
        hipSetDevice(m->devices[i]);
        if (m->ctxs[i]) kzg_sync(m->ctxs[i]);
        if (i < (int)m->comms.size() && m->comms[i] && g_rccl) g_rccl->CommDestroy(m->comms[i]);
        if (m->d_part[i]) hipFree(m->d_part[i]);
        if (m->d_gath[i]) hipFree(m->d_gath[i]);
        if (m->d_quot[i]) hipFree(m->d_quot[i]);
        if (m->h_status[i]) hipHostFree(m->h_status[i]);
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
    snprintf(buf, buflen, "rccl=%s version=%d hip=%s world=%d local=%d mode=%s", r->path.c_str(), r->version, r->hip_path.c_str(),
             m->world, m->nlocal(), m->per_process ? "process-per-gpu" : "one-process");
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
    for (int i = 0; i < m->nlocal(); i++) {
        int rc = kzg_ctx_set_option(m->ctxs[i], key, value);
        if (rc != KZG_OK) return mfail_ctx(m, i, rc);
    }
    return KZG_OK;
}

// ---------------------------------------------------------------------------------------------
// sharded SRS
// ---------------------------------------------------------------------------------------------
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
    for (size_t i = 0; i < s->shards.size(); i++)
        if (s->shards[i]) kzg_srs_free(m && i < m->ctxs.size() ? m->ctxs[i] : nullptr, s->shards[i]);
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

static int mctx_buffers(kzg_mctx *m, size_t batch) {
    for (int i = 0; i < m->nlocal(); i++) {
        if (m->cap_points[i] >= batch) continue;
        if (hipSetDevice(m->devices[i]) != hipSuccess) return mfail(m, KZG_ERR_HIP, "hipSetDevice");
        kzg_sync(m->ctxs[i]);
        if (m->d_part[i]) hipFree(m->d_part[i]);
        if (m->d_gath[i]) hipFree(m->d_gath[i]);
        m->d_part[i] = m->d_gath[i] = nullptr;
        m->cap_points[i] = 0;
        size_t cap = batch < 64 ? 64 : batch;
        if (hipMalloc(&m->d_part[i], record_bytes(cap)) != hipSuccess ||
            hipMalloc(&m->d_gath[i], record_bytes(cap) * (size_t)m->world) != hipSuccess)
            return mfail(m, KZG_ERR_ALLOC, "hipMalloc(partial-point exchange buffers)");
        if (!m->h_status[i] && hipHostMalloc(&m->h_status[i], STATUS_ALL_OFF + STATUS_BYTES * (size_t)m->world, hipHostMallocDefault) != hipSuccess)
            return mfail(m, KZG_ERR_ALLOC, "hipHostMalloc(status words)");
        m->cap_points[i] = cap;
    }
    return KZG_OK;
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
        KZG_TRY(mctx_comm(m, &r));
        const size_t rec = record_bytes(batch);
        for (int i = 0; i < m->nlocal(); i++) {  // the status block rides behind the partials, in stream order
            hipSetDevice(m->devices[i]);
            int32_t *hs = (int32_t *)m->h_status[i];
            hs[0] = local_rc[i];
            hs[1] = m->ranks[i];
            hs[2] = hs[3] = 0;
            if (hipMemcpyAsync((uint8_t *)m->d_part[i] + batch * PARTIAL_BYTES, hs, STATUS_BYTES, hipMemcpyHostToDevice,
                               m->ctxs[i]->lanes[0].stream) != hipSuccess)
                return mfail(m, KZG_ERR_HIP, "status upload");
        }
        KZG_NCCL(m, r, r->GroupStart());
        for (int i = 0; i < m->nlocal(); i++) {
            ncclResult_t e = r->AllGather(m->d_part[i], m->d_gath[i], rec, ncclUint8, m->comms[i], m->ctxs[i]->lanes[0].stream);
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
                             (size_t)m->world, hipMemcpyDeviceToHost, m->ctxs[0]->lanes[0].stream) != hipSuccess)
            return mfail(m, KZG_ERR_HIP, "status download");
        for (int i = 0; i < m->nlocal(); i++) {
            hipSetDevice(m->devices[i]);
            if (hipStreamSynchronize(m->ctxs[i]->lanes[0].stream) != hipSuccess) return mfail(m, KZG_ERR_HIP, "all-gather failed");
        }
        for (int rk = 0; rk < m->world; rk++)
            if (hall[4 * rk] != KZG_OK) {
                if (first_bad >= 0 && m->ranks[first_bad] == rk) return mfail_ctx(m, first_bad, local_rc[first_bad]);
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
    for (int i = 0; i < m->nlocal(); i++) {
        if (m->cap_quot[i] >= n) continue;
        hipSetDevice(m->devices[i]);
        kzg_sync(m->ctxs[i]);
        if (m->d_quot[i]) hipFree(m->d_quot[i]);
        m->d_quot[i] = nullptr;
        m->cap_quot[i] = 0;
        if (hipMalloc(&m->d_quot[i], n * 32) != hipSuccess) return mfail(m, KZG_ERR_ALLOC, "hipMalloc(quotient)");
        m->cap_quot[i] = n;
    }
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
    if (k == 0 || k > 4096) return mfail(m, KZG_ERR_SHAPE, "1 <= opening points <= 4096");
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

#ifdef KZG_TEST_HOOKS
extern "C" int kzg_test_mctx_inject_failure(kzg_mctx *m, int code) {
    if (!m) return KZG_ERR_SHAPE;
    std::lock_guard<std::mutex> lk(m->mu);
    m->inject_fail = code;
    return KZG_OK;
}
#endif
