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

#include <thread>

#include "common.h"

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
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

static std::mutex g_rccl_mu;
static Rccl *g_rccl = nullptr;

static Rccl *rccl_load(std::string *err) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl) return g_rccl;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);  // a copy this process already holds (e.g. PyTorch's)
    const char *names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (int i = 0; !h && i < 3; i++) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        *err = std::string("cannot load RCCL (librccl.so.1): ") + (dlerror() ? dlerror() : "not found");
        return nullptr;
    }
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
    if (!ok) {
        delete r;
        return nullptr;
    }
    g_rccl = r;
    return r;
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
    int nlocal() const { return (int)devices.size(); }
};

struct kzg_msrs {
    size_t n = 0;
    std::vector<kzg_srs *> shards;   // one per local GPU
    std::vector<size_t> first, len;
};

static int mfail(kzg_mctx *m, int code, const std::string &msg) {
    m->err = msg;
    return code;
}

static int mfail_ctx(kzg_mctx *m, int i, int code) {
    const char *e = kzg_last_error(m->ctxs[i]);
    m->err = "GPU " + std::to_string(m->devices[i]) + ": " + (e ? e : "");
    return code;
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
    for (int i = 0; i < m->nlocal(); i++) {
        kzg_ctx *c = nullptr;
        int rc = kzg_ctx_create(m->devices[i], &c);
        if (rc != KZG_OK) return rc;
        m->ctxs.push_back(c);
    }
    m->d_part.assign(m->nlocal(), nullptr);
    m->d_gath.assign(m->nlocal(), nullptr);
    m->cap_points.assign(m->nlocal(), 0);
    m->d_quot.assign(m->nlocal(), nullptr);
    m->cap_quot.assign(m->nlocal(), 0);
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
    for (int i = 0; i < (int)m->ctxs.size(); i++) {
        hipSetDevice(m->devices[i]);
        if (m->ctxs[i]) kzg_sync(m->ctxs[i]);
        if (i < (int)m->comms.size() && m->comms[i] && g_rccl) g_rccl->CommDestroy(m->comms[i]);
        if (m->d_part[i]) hipFree(m->d_part[i]);
        if (m->d_gath[i]) hipFree(m->d_gath[i]);
        if (m->d_quot[i]) hipFree(m->d_quot[i]);
        if (m->ctxs[i]) kzg_ctx_destroy(m->ctxs[i]);
    }
    delete m;
}

extern "C" const char *kzg_mctx_last_error(kzg_mctx *m) { return m ? m->err.c_str() : "null group"; }
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
        if (hipMalloc(&m->d_part[i], cap * PARTIAL_BYTES) != hipSuccess ||
            hipMalloc(&m->d_gath[i], cap * PARTIAL_BYTES * (size_t)m->world) != hipSuccess)
            return mfail(m, KZG_ERR_ALLOC, "hipMalloc(partial-point exchange buffers)");
        m->cap_points[i] = cap;
    }
    return KZG_OK;
}

// Stage 2 of every sharded operation: d_part[i] holds `batch` Jacobian partials on every local GPU.  One all-gather, then the
// group's first local GPU adds the `world` partials of each polynomial and writes `batch` points in ofmt to `out` (host).
static int mctx_combine(kzg_mctx *m, size_t batch, void *out, int ofmt) {
    const void *src = m->d_part[0];
    size_t count = 1;
    if (m->world > 1 || m->always_gather) {
        Rccl *r = nullptr;
        KZG_TRY(mctx_comm(m, &r));
        KZG_NCCL(m, r, r->GroupStart());
        for (int i = 0; i < m->nlocal(); i++) {
            ncclResult_t e = r->AllGather(m->d_part[i], m->d_gath[i], batch * PARTIAL_BYTES, ncclUint8, m->comms[i],
                                          m->ctxs[i]->lanes[0].stream);
            if (e != ncclSuccess) {
                r->GroupEnd();
                return mfail(m, KZG_ERR_HIP, std::string("ncclAllGather: ") + r->GetErrorString(e));
            }
        }
        KZG_NCCL(m, r, r->GroupEnd());
        // every local GPU's copy must be complete before its buffers are re-used by the next call
        for (int i = 1; i < m->nlocal(); i++) {
            hipSetDevice(m->devices[i]);
            if (hipStreamSynchronize(m->ctxs[i]->lanes[0].stream) != hipSuccess) return mfail(m, KZG_ERR_HIP, "all-gather failed");
        }
        src = m->d_gath[0];  // [world][batch] partials; the sum below runs on the same stream, after the collective
        count = (size_t)m->world;
    }
    int rc = g1_sum_batch_strided(m->ctxs[0], src, count, batch, 1, batch, KZG_G1_JACOBIAN_MONT_144, KZG_IN_DEVICE, out, ofmt,
                                  POINTS_TRUSTED);  // the group's own partial sums
    if (rc != KZG_OK) return mfail_ctx(m, 0, rc);
    return KZG_OK;
}

// run f(i) for every local GPU (one host thread each when there are several), return the first failure
template <class F>
static int for_each_local(kzg_mctx *m, F f) {
    const int L = m->nlocal();
    std::vector<int> rcs(L, KZG_OK);
    if (L == 1) {
        rcs[0] = f(0);
    } else {
        std::vector<std::thread> th;
        for (int i = 0; i < L; i++) th.emplace_back([&, i]() { rcs[i] = f(i); });
        for (auto &t : th) t.join();
    }
    for (int i = 0; i < L; i++)
        if (rcs[i] != KZG_OK) return mfail_ctx(m, i, rcs[i]);
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
    KZG_TRY(for_each_local(m, [&](int i) {
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
    }));
    return mctx_combine(m, batch, out, ofmt);
}

extern "C" int kzg_commit_coeff_sharded(kzg_mctx *m, const kzg_msrs *srs, const void *coeffs, size_t n, int sfmt, int flags,
                                        void *out, int ofmt) {
    return kzg_commit_coeff_sharded_batch(m, srs, coeffs, n, 1, sfmt, flags, out, ofmt);
}

extern "C" int kzg_witness_coeff_sharded(kzg_mctx *m, const kzg_msrs *srs, const void *coeffs, size_t n, const void *x,
                                         const void *y, int sfmt, void *out, int ofmt) {
    // KZGProver::create_witness (src/coeff_form.rs:66-81): q = (p - y)/(X - x) has n - 1 coefficients; rank r reduces
    // q[lo_r, hi_r) against its shard.  The O(n) quotient scan is replicated on every GPU.
    if (!m || !srs || !coeffs || !x || !y || !out || n == 0) return KZG_ERR_SHAPE;
    std::lock_guard<std::mutex> lk(m->mu);
    if ((int)srs->shards.size() != m->nlocal()) return mfail(m, KZG_ERR_SHAPE, "SRS belongs to another group");
    if (n - 1 > srs->n) return mfail(m, KZG_ERR_SHAPE, "quotient longer than the SRS (reference: slice index panic)");
    if (!point_format_bytes(ofmt)) return mfail(m, KZG_ERR_SHAPE, "unknown G1 output format");
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
    std::vector<int> off_poly(m->nlocal(), 0);
    int rc = for_each_local(m, [&](int i) {
        int q = kzg_quotient_linear(m->ctxs[i], coeffs, n, x, y, sfmt, KZG_OUT_DEVICE, m->d_quot[i]);
        if (q == KZG_ERR_POINT_NOT_ON_POLY) {
            off_poly[i] = 1;  // the reference fails after the division; the collective below must still be entered by all
            q = KZG_OK;
        }
        if (q != KZG_OK) return q;
        const size_t len = clip_len(srs->first[i], srs->len[i], n - 1);
        return msm_batch_strided(m->ctxs[i], srs->shards[i], 0, (const uint8_t *)m->d_quot[i] + srs->first[i] * 32, len, 1,
                                 len * 32, sfmt, KZG_IN_DEVICE | KZG_OUT_DEVICE, m->d_part[i], KZG_G1_JACOBIAN_MONT_144);
    });
    if (rc != KZG_OK) return rc;
    KZG_TRY(mctx_combine(m, 1, out, ofmt));
    for (int i = 0; i < m->nlocal(); i++)
        if (off_poly[i]) return mfail(m, KZG_ERR_POINT_NOT_ON_POLY, "point not on polynomial!");
    return KZG_OK;
}
