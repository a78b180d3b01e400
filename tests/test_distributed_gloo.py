"""N>1 protocol on CPU: world_size-2 torch.distributed (gloo), contiguous SRS shards (kzg_shard_range), one partial point
per rank and polynomial plus the status slot, all_gather laid out [world][batch + 1], status agreement, per-polynomial sums --
tests/protocol_model.py with the oracle standing in for the per-rank GPU operations (tests may use the oracle; the product never
does).  The product's own exchange (kzg_amd/csrc/mgpu.hip, RCCL) is exercised on the GPU by tests/test_gpu_mgpu.py; this test
pins the partition rule, the gathered layout, the unique-id hand-off and "a failing rank fails every rank, nobody hangs"."""
import os
import random
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kzg_amd.distributed import broadcast_unique_id, shard_range
from tests.protocol_model import GroupDead, ProtocolModel, RankFailed
from oracle import c_oracle as C
from oracle import kzg_model as M


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, tau, seed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = random.Random(seed)
    coeffs = [rng.randrange(M.R) for _ in range(n)]
    lo, hi = shard_range(n, rank, world)
    full = C.setup_g1(tau, n)
    shard = full[96 * lo: 96 * hi]

    def local_msm(polys, batch):
        return b"".join(C.msm_g1(shard, sc) for sc in polys)

    def local_sum(raw, nranks, batch, stride):        # [world][batch + 1][96], as ncclAllGather leaves it
        res = []
        for b in range(batch):
            acc = bytes(96)
            for w in range(nranks):                   # = k_sum_groups with gstride 1, istride batch + 1
                acc = C.g1_add(acc, raw[96 * (w * stride + b): 96 * (w * stride + b + 1)])
            res.append(acc)
        return res

    # the 128-byte communicator id travels from rank 0 to everyone over the process group
    uid = broadcast_unique_id(dist, rank, lambda: bytes(range(128)))
    assert uid == bytes(range(128))
    committer = ProtocolModel(dist, rank, world, local_msm, local_sum)
    # a batch of two polynomials: p and 3*p (distinguishes the [world][batch] layout from its transpose)
    got = committer.commit_batch([coeffs[lo:hi], [3 * c % M.R for c in coeffs[lo:hi]]], 2)
    # status agreement: rank 1's local phase fails (allocation failure, code -3); BOTH ranks leave the exchange with that error
    class Boom(Exception):
        status = -3

    def failing_msm(polys, batch):
        if rank == 1:
            raise Boom()
        return local_msm(polys, batch)

    try:
        ProtocolModel(dist, rank, world, failing_msm, local_sum).commit_batch([coeffs[lo:hi]], 1)
        agreed = None
    except RankFailed as e:
        agreed = (e.rank, e.status)
    # and the group is still usable afterwards
    again = committer.commit_batch([coeffs[lo:hi]], 1)
    q.put((rank, (got, agreed, again)))
    dist.barrier()
    dist.destroy_process_group()


def _deadline_worker(rank, world, port, q):
    """rank 1 never enters the second exchange (a dead peer); rank 0's wait has a deadline"""
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pm = ProtocolModel(dist, rank, world, lambda polys, batch: bytes(96 * batch), lambda raw, w, b, s: [raw[:96]] * b, gather_timeout_s=1.5)
    first = pm.commit_batch([[]], 1)       # a healthy exchange under the same deadline
    out = ("ok", first == [bytes(96)])
    if rank == 0:
        t0 = time.time()
        try:
            pm.commit_batch([[]], 1)
            out += ("completed",)
        except GroupDead:
            out += ("dead after %.1f s" % (time.time() - t0),)
        try:
            pm.commit_batch([[]], 1)           # the group stays dead: no second wait
            out += ("completed",)
        except GroupDead as e:
            out += (str(e),)
    else:
        time.sleep(4.0)                        # alive but never arrives
    q.put((rank, out))
    q.close()
    q.join_thread()
    os._exit(0)                                # no further collective on a retired group


def test_exchange_deadline_world2_gloo():
    """The time-out rule of mgpu.hip (mctx_wait / option gather_timeout_ms) at world size 2: a peer that never enters the exchange
    turns into an error on the survivor within the deadline, and the survivor's group is dead afterwards -- no hang."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_deadline_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
    assert results[0][:2] == ("ok", True) and results[1][:2] == ("ok", True)
    assert results[0][2].startswith("dead after") and float(results[0][2].split()[2]) < 3.5
    assert results[0][3] == "this device group is dead"


def test_shard_range_partitions():
    for n in (0, 1, 7, 1000, 1 << 20, (1 << 24) + 5):
        for world in (1, 2, 3, 8):
            rs = [shard_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in rs) - min(h - l for l, h in rs) <= 1


def test_sharded_commit_world2_gloo():
    world, n, tau, seed = 2, 257, 0x1234ABCD, 5
    C.build()   # before the workers start: two processes must not race to build the oracle library on a fresh checkout
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, tau, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    # (cold start: two torch imports and the first oracle build can take minutes on a fresh checkout)
    results = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    rng = random.Random(seed)
    coeffs = [rng.randrange(M.R) for _ in range(n)]
    ptau = C.poly_eval(coeffs, tau)
    want = [C.g1_mul(C.g1_generator(), ptau), C.g1_mul(C.g1_generator(), 3 * ptau % M.R)]   # [p(tau)]G, [3p(tau)]G
    assert results[0][0] == results[1][0] == want
    assert results[0][1] == results[1][1] == (1, -3)
    assert results[0][2] == results[1][2] == want[:1]


def _growth_worker(rank, world, port, q):
    """rank 1's allocation fails once when the exchange buffers must grow (batch 70 > the 64 held from the start)"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fails = [1 if rank == 1 else 0]

    class NoMemory(Exception):
        status = -3

    def grow(batch):
        if fails[0]:
            fails[0] -= 1
            raise NoMemory()

    pm = ProtocolModel(dist, rank, world, lambda polys, batch: bytes([rank + 1]) * (96 * batch),
                       lambda raw, w, b, stride: [bytes(raw[96 * (r * stride + i)] for r in range(w)) for i in range(b)], grow=grow)
    out = [pm.commit_batch(None, 3)]                    # ordinary call: no agreement
    try:
        pm.commit_batch(None, 70)
        out.append("no error")
    except RankFailed as e:
        out.append((e.rank, e.status))
    caps = (pm.cap, pm.agreed)                          # the ranks now differ in capacity, not in what they agreed on
    out.append(pm.commit_batch(None, 70)[:2])           # every rank enters the agreement again (a no-op growth where it already grew)
    out.append(pm.commit_batch(None, 66)[:1])           # below what was agreed: straight to the exchange on every rank
    q.put((rank, (out, caps, pm.collectives)))
    dist.barrier()
    dist.destroy_process_group()


def test_growth_failure_on_one_rank_keeps_the_collective_sequence_world3_gloo():
    """mctx_buffers' rule at world size 3: whether to run the status-only agreement is decided by what the ranks last agreed on, so
    the ranks enter the same collectives in the same order even after a growth that failed on one of them only."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_growth_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want_row = bytes([1, 2, 3])                          # slot r * (batch + 1) + i holds rank r's fill byte
    for r in range(world):
        out, caps, coll = results[r]
        assert out[0] == [want_row] * 3 and out[1] == (1, -3) and out[2] == [want_row] * 2 and out[3] == [want_row]
        assert caps == ((64, 64) if r == 1 else (70, 64))
        assert coll == results[0][2]                     # the same collectives, in the same order, on every rank
    assert [c[0] for c in results[0][2]] == ["gather", "agree", "agree", "gather", "gather"]
