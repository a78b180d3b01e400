"""N>1 protocol on CPU: world_size-2 torch.distributed (gloo), contiguous SRS shards (kzg_shard_range), one partial point
per rank and polynomial, all_gather laid out [world][batch], per-polynomial sums -- kzg_amd.distributed.ProtocolModel with the
oracle standing in for the per-rank GPU operations (tests may use the oracle; the product never does).  The product's own
exchange (kzg_amd/csrc/mgpu.hip, RCCL) is exercised on the GPU by tests/test_gpu_mgpu.py; this test pins the partition rule,
the gathered layout and the unique-id hand-off it relies on."""
import os
import random
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kzg_amd.distributed import ProtocolModel, broadcast_unique_id, shard_range
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

    def local_sum(gathered, nranks, batch):
        raw = bytes(gathered.numpy().tobytes())      # [world][batch][96], as ncclAllGather leaves it
        res = []
        for b in range(batch):
            acc = bytes(96)
            for w in range(nranks):                   # = k_sum_groups with gstride 1, istride batch
                acc = C.g1_add(acc, raw[96 * (w * batch + b): 96 * (w * batch + b + 1)])
            res.append(acc)
        return res

    # the 128-byte communicator id travels from rank 0 to everyone over the process group
    uid = broadcast_unique_id(dist, rank, lambda: bytes(range(128)))
    assert uid == bytes(range(128))
    committer = ProtocolModel(dist, rank, world, local_msm, local_sum)
    # a batch of two polynomials: p and 3*p (distinguishes the [world][batch] layout from its transpose)
    got = committer.commit_batch([coeffs[lo:hi], [3 * c % M.R for c in coeffs[lo:hi]]], 2)
    q.put((rank, got))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    for n in (0, 1, 7, 1000, 1 << 20, (1 << 24) + 5):
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
