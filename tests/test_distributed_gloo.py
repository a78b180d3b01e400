"""N>1 data path on CPU: world_size-2 torch.distributed (gloo), contiguous SRS shards, one partial point
per rank, all_gather of 96-byte partials, local sum -- kzg_amd.distributed.ShardedCommitter with the
oracle standing in for the per-rank GPU operations (tests may use the oracle; the product never does)."""
import os
import random
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kzg_amd.distributed import ShardedCommitter, shard_range
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
        return torch.frombuffer(bytearray(b"".join(C.msm_g1(shard, sc) for sc in polys)), dtype=torch.uint8)

    def local_sum(grouped, batch):
        raw = bytes(grouped.numpy().tobytes())
        per = len(raw) // batch
        res = []
        for b in range(batch):
            acc = bytes(96)
            for i in range(0, per, 96):
                acc = C.g1_add(acc, raw[b * per + i: b * per + i + 96])
            res.append(acc)
        return res

    committer = ShardedCommitter(dist, rank, world, local_msm, local_sum)
    # a batch of two polynomials: p and 3*p (tests the [world][batch] -> [batch][world] regrouping)
    got = committer.commit_batch([coeffs[lo:hi], [3 * c % M.R for c in coeffs[lo:hi]]], 2)
    q.put((rank, got))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    for n in (0, 1, 7, 1000, 1 << 20):
        for world in (1, 2, 3, 8):
            rs = [shard_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in rs) - min(h - l for l, h in rs) <= 1


def test_sharded_commit_world2_gloo():
    world, n, tau, seed = 2, 257, 0x1234ABCD, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, tau, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = random.Random(seed)
    coeffs = [rng.randrange(M.R) for _ in range(n)]
    ptau = C.poly_eval(coeffs, tau)
    want = [C.g1_mul(C.g1_generator(), ptau), C.g1_mul(C.g1_generator(), 3 * ptau % M.R)]   # [p(tau)]G, [3p(tau)]G
    assert results[0] == results[1] == want
