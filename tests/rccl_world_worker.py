"""One rank of a REAL device group: one process per GPU, the product library, RCCL over xGMI -- run by tests/test_gpu_rccl_world.py
on boxes with at least two GPUs.  The RCCL unique id travels over the bench's TCP star (tools/benchlib/control.py; no torch).  Every
rank runs the same calls and prints one JSON object; the parent compares the ranks with each other and with the oracle.

  python tests/rccl_world_worker.py <rank> <world> <port> <seed>
"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import kzg_amd  # noqa: E402
from kzg_amd import _lib as L  # noqa: E402
from kzg_amd.api import DeviceGroup  # noqa: E402
from kzg_amd.distributed import shard_range  # noqa: E402
from tools.benchlib.control import TcpStar  # noqa: E402

R = kzg_amd.api.R_MODULUS
TAU = 0x0BADC0FFEE123457
N = (1 << 18) + 37     # ragged shards


def horner(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R
    return acc


def main():
    rank, world, port, seed = (int(v) for v in sys.argv[1:5])
    real_stdout = os.dup(1)
    os.dup2(2, 1)        # RCCL's banner goes to stderr; stdout carries the JSON object only
    star = TcpStar(rank, world, "127.0.0.1", port, timeout_s=120)
    # Forming the communicator is the ENVIRONMENT's part (RCCL's bootstrap over the node's fabric): when it does not happen within the
    # library's deadlines every rank reports why and the parent SKIPS -- a wrong result below is what fails the test.
    environment = ("did not return within", "did not complete within", "never returned", "cannot load RCCL", "communicator formation abandoned")
    rng = random.Random(seed)
    p = [rng.randrange(R) for _ in range(N)]
    try:
        uid = None
        if rank == 0:
            try:
                uid = DeviceGroup.unique_id()
            except kzg_amd.EngineError as e:      # the others must not be left waiting for the id
                uid = {"error": str(e)}
        uid = star.all_gather(uid)[0]
        if isinstance(uid, dict):
            raise kzg_amd.EngineError(uid["error"])
        group = DeviceGroup.for_rank(rank, rank, world, uid)          # device = rank: one process per GPU
        if world == 1:
            group.set_option("always_gather", 1)      # a group of one still forms its communicator and runs the all-gather
        out = {"rank": rank, "world": group.world, "info": group.info(), "torch_imported": "torch" in sys.modules}
        srs = group.setup(TAU, N)
        shard, first = srs.shard(0)
        lo, hi = shard_range(N, rank, world)
        assert (first, len(shard)) == (lo, hi - lo)
        out["commit"] = group.commit(srs, p).hex()                    # the first exchange: the communicator is formed here at the latest
    except kzg_amd.EngineError as e:
        if not any(t in str(e) for t in environment):
            raise
        os.write(real_stdout, (json.dumps({"rank": rank, "environment": str(e)[:1500]}) + "\n").encode())
        return
    out["formation"] = group.formation()
    batch = 4
    bp = [[rng.randrange(R) for _ in range(N)] for _ in range(batch - 2)] + [[0] * N, [R - 1] * N]
    flat = kzg_amd.pack_scalars([c for q in bp for c in q])
    out["batch"] = [b.hex() for b in group.commit_batch(srs, flat, N, batch)]
    b = group.engine(0).alloc_scalars((hi - lo) * batch)
    b.upload(kzg_amd.pack_scalars([c for q in bp for c in q[lo:hi]]))
    out["batch_device"] = [v.hex() for v in group.commit_batch(srs, [b], N, batch)]
    b.free()
    x = rng.randrange(R)
    y = horner(p, x)
    out["witness"] = group.create_witness(srs, p, (x, y)).hex()
    try:
        group.create_witness(srs, p, (x, (y + 1) % R))
        out["witness_off_poly"] = "no error"
    except kzg_amd.PointNotOnPolynomial:
        out["witness_off_poly"] = "PointNotOnPolynomial"
    xs = [rng.randrange(R) for _ in range(5)]
    w, r = group.create_witness_batched(srs, p, [(v, horner(p, v)) for v in xs])
    out["witness_batched"] = [w.hex(), [hex(c) for c in r]]
    out["info_after"] = group.info()
    star.all_gather(None)
    srs.free()
    group.close()
    star.close()
    os.write(real_stdout, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
