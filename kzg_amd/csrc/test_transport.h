// test_transport.h -- KZG_TEST_HOOKS builds only (kzg_amd/libkzg_mi355x_hooks.so, loaded by tests/): a stand-in for the eight RCCL
// entry points mgpu.hip uses, selected by KZG_TEST_SHM_TRANSPORT=1, that exchanges through host memory instead of xGMI.
//
// Why it exists: RCCL refuses two ranks on one GPU ("Duplicate GPU detected"), and the test box has one GPU -- so with RCCL alone the
// group code can only ever run at world size 1 there.  With this transport `world` processes (or one process with `world` contexts)
// share GPU 0 and everything ABOVE the transport runs for real at world 2..8: kzg_shard_range, the per-rank SRS shards, per-rank
// partial MSMs over different slices, the [world][batch + 1] record layout, the sum of partials from different ranks, the status
// agreement with one rank failing, the persistent per-GPU worker threads and the grouped all-gather of the one-process mode.
// RCCL itself (communicator, ncclAllGather on the lane's stream, the polled deadline and abort) is exercised at world 1 by
// tests/test_gpu_mgpu.py.  The product library contains none of this.
//
// Semantics: ncclAllGather is blocking here (the stream is drained, the record goes through a POSIX shared-memory segment named by
// the unique id, a counting barrier, then every rank's record is copied back to the device); GroupStart / GroupEnd defer the calls
// of a group so that one process can drive several ranks.  A peer that never arrives makes the barrier fail after 30 s.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

namespace kzg {
namespace shmt {

constexpr size_t SLOT_BYTES = 4u << 20;  // largest record: (batch + 1) x 144 B  =>  batch <= 29126
constexpr int MAX_WORLD = 64;

struct Seg {
    std::atomic<uint32_t> arrived, gen, users;
    uint32_t pad;
    uint64_t bytes[MAX_WORLD];  // what each rank brought to the collective in flight: ranks that disagree are told so
    uint8_t data[1];            // world x SLOT_BYTES
};

struct Comm {
    int rank = 0, world = 1;
    Seg *seg = nullptr;
    size_t map_bytes = 0;  // 0: heap (every rank in this process)
};

struct Op {
    Comm *c;
    const void *send;
    void *recv;
    size_t bytes;
    hipStream_t st;
};

static thread_local std::vector<Op> t_ops;
static thread_local int t_depth = 0;

static size_t seg_bytes(int world) { return sizeof(Seg) + (size_t)world * SLOT_BYTES; }

static ncclResult_t barrier(Comm *c) {
    if (!c->map_bytes) return ncclSuccess;  // one process drives every rank: nothing to wait for
    Seg *s = c->seg;
    const uint32_t g = s->gen.load();
    if (s->arrived.fetch_add(1) + 1 == (uint32_t)c->world) {
        s->arrived.store(0);
        s->gen.fetch_add(1);
        return ncclSuccess;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (s->gen.load() == g) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) return ncclSystemError;
        usleep(20);
    }
    return ncclSuccess;
}

static ncclResult_t get_unique_id(ncclUniqueId *id) {
    memset(id, 0, sizeof *id);
    std::random_device rd;
    snprintf(id->internal, sizeof id->internal, "/kzg-test-%d-%08x%08x", (int)getpid(), (unsigned)rd(), (unsigned)rd());
    return ncclSuccess;
}

static ncclResult_t comm_init_rank(ncclComm_t *out, int world, ncclUniqueId id, int rank) {
    if (world < 1 || world > MAX_WORLD || rank < 0 || rank >= world || id.internal[0] != '/') return ncclInvalidArgument;
    id.internal[sizeof id.internal - 1] = 0;
    int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    const size_t bytes = seg_bytes(world);
    if (ftruncate(fd, (off_t)bytes) != 0) {  // every rank sets the same size; new pages read as zero
        close(fd);
        return ncclSystemError;
    }
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    Comm *c = new Comm();
    c->rank = rank;
    c->world = world;
    c->seg = (Seg *)p;
    c->map_bytes = bytes;
    ncclResult_t e = barrier(c);           // like ncclCommInitRank: returns when every rank is here
    if (rank == 0) shm_unlink(id.internal);  // the mappings keep the segment alive
    if (e != ncclSuccess) {
        munmap(p, bytes);
        delete c;
        return e;
    }
    *out = (ncclComm_t)c;
    return ncclSuccess;
}

static ncclResult_t comm_init_all(ncclComm_t *out, int n, const int *) {
    if (n < 1 || n > MAX_WORLD) return ncclInvalidArgument;
    Seg *s = (Seg *)calloc(1, seg_bytes(n));
    if (!s) return ncclSystemError;
    s->users.store((uint32_t)n);
    for (int i = 0; i < n; i++) {
        Comm *c = new Comm();
        c->rank = i;
        c->world = n;
        c->seg = s;
        out[i] = (ncclComm_t)c;
    }
    return ncclSuccess;
}

static ncclResult_t comm_destroy(ncclComm_t h) {
    Comm *c = (Comm *)h;
    if (!c) return ncclSuccess;
    if (c->map_bytes) munmap(c->seg, c->map_bytes);
    else if (c->seg->users.fetch_sub(1) == 1) free(c->seg);
    delete c;
    return ncclSuccess;
}

static ncclResult_t flush() {
    std::vector<Op> ops;
    ops.swap(t_ops);
    if (ops.empty()) return ncclSuccess;
    Comm *c0 = ops[0].c;
    // one process driving every rank must present all of them in one group; one process per rank presents exactly one
    if (c0->map_bytes ? ops.size() != 1 : (int)ops.size() != c0->world) return ncclInvalidUsage;
    for (const Op &o : ops) {
        if (o.bytes > SLOT_BYTES || o.c->seg != c0->seg) return ncclInvalidArgument;
        o.c->seg->bytes[o.c->rank] = o.bytes;
        if (hipMemcpyAsync(o.c->seg->data + (size_t)o.c->rank * SLOT_BYTES, o.send, o.bytes, hipMemcpyDeviceToHost, o.st) != hipSuccess ||
            hipStreamSynchronize(o.st) != hipSuccess)
            return ncclUnhandledCudaError;
    }
    ncclResult_t e = barrier(c0);  // every rank's record is in the segment
    if (e != ncclSuccess) return e;
    bool same = true;  // (real RCCL would hang or corrupt: here the ranks learn that they are not in the same collective)
    for (int r = 0; r < c0->world; r++) same = same && c0->seg->bytes[r] == ops[0].bytes;
    if (!same) {
        barrier(c0);
        return ncclInvalidUsage;
    }
    for (const Op &o : ops) {
        for (int r = 0; r < o.c->world; r++)
            if (hipMemcpyAsync((uint8_t *)o.recv + (size_t)r * o.bytes, o.c->seg->data + (size_t)r * SLOT_BYTES, o.bytes,
                               hipMemcpyHostToDevice, o.st) != hipSuccess)
                return ncclUnhandledCudaError;
        if (hipStreamSynchronize(o.st) != hipSuccess) return ncclUnhandledCudaError;
    }
    return barrier(c0);  // nobody overwrites its slot before every rank has read it
}

static ncclResult_t all_gather(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclComm_t h, hipStream_t st) {
    if (dt != ncclUint8 || !h) return ncclInvalidArgument;
    t_ops.push_back(Op{(Comm *)h, send, recv, count, st});
    return t_depth ? ncclSuccess : flush();
}

static ncclResult_t group_start() {
    t_depth++;
    return ncclSuccess;
}

static ncclResult_t group_end() {
    if (t_depth <= 0) return ncclInvalidUsage;
    return --t_depth ? ncclSuccess : flush();
}

static ncclResult_t get_version(int *v) {
    *v = 0;
    return ncclSuccess;
}

static const char *error_string(ncclResult_t e) {
    switch (e) {
    case ncclSuccess: return "no error";
    case ncclSystemError: return "test transport: shared-memory segment or barrier failed (a peer never arrived?)";
    case ncclInvalidArgument: return "test transport: invalid argument (record larger than its slot?)";
    case ncclInvalidUsage: return "test transport: the ranks are not in the same collective (different sizes), or a group does not hold one all-gather per rank of this process";
    default: return "test transport: HIP error";
    }
}

}  // namespace shmt
}  // namespace kzg
