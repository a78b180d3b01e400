"""One rank (or the one driving process) of a device group of world size W on ONE GPU -- run by tests/test_gpu_mgpu_world.py.

RCCL refuses two ranks on one GPU, so the group runs over the test transport of the hooks build (kzg_amd/csrc/test_transport.h,
KZG_TEST_SHM_TRANSPORT=1): everything above the transport is the product code of mgpu.hip at world > 1.  Every rank runs the same
scenario on inputs derived from the seed and prints one JSON object; the parent compares the ranks with each other and with the
oracle.

  python tests/mgpu_world_worker.py rank <rank> <world> <unique id hex> <seed>
  python tests/mgpu_world_worker.py one  <world> <seed>
"""
import ctypes
import hashlib
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
assert os.environ.get("KZG_TEST_SHM_TRANSPORT"), "run by tests/test_gpu_mgpu_world.py"

from kzg_amd import _lib as L  # noqa: E402

L.load(os.path.join(ROOT, "kzg_amd", "libkzg_mi355x_hooks.so"))  # the hooks build as THE library of this process
import kzg_amd  # noqa: E402
from kzg_amd.distributed import shard_range  # noqa: E402

R = kzg_amd.api.R_MODULUS
TAU = 0x0BADC0FFEE123457
N = 5003          # ragged for every world size tested (2, 3, 8)
D = 1 << 11       # evaluation form


def horner(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R
    return acc


def scenario(group, rank_of_failure, seed):
    """rank_of_failure: the global rank that injects its failures (this process does so when it holds that rank)."""
    lib = group.lib
    for name, args in {"kzg_test_mctx_inject_failure": [ctypes.c_void_p, ctypes.c_int],
                       "kzg_test_mctx_inject_alloc_failure": [ctypes.c_void_p]}.items():
        getattr(lib, name).argtypes = args
        getattr(lib, name).restype = ctypes.c_int
    ranks = [group.rank(i) for i in range(group.local_count)]
    i_fail = rank_of_failure in ranks and (group.local_count == 1 or ranks.index(rank_of_failure) == 0)
    rng = random.Random(seed)
    out = {"world": group.world, "ranks": ranks, "info": group.info()}
    srs = group.setup(TAU, N)
    shards = []
    for i in range(group.local_count):
        shard, first = srs.shard(i)
        lo, hi = shard_range(N, group.rank(i), group.world)
        assert (first, len(shard)) == (lo, hi - lo)
        shards.append([first, len(shard), hashlib.sha256(shard.download()).hexdigest()])
    out["shards"] = shards
    polys = {m: [rng.randrange(R) for _ in range(m)] for m in (N, 777, 1, 0)}
    out["commit"] = {str(m): group.commit(srs, p).hex() for m, p in polys.items()}
    # a batch, host-resident and as device-resident per-GPU slices
    batch = 5
    bp = [[rng.randrange(R) for _ in range(N)] for _ in range(batch - 2)] + [[0] * N, [R - 1] * N]
    flat = kzg_amd.pack_scalars([c for p in bp for c in p])
    out["batch_host"] = [b.hex() for b in group.commit_batch(srs, flat, N, batch)]
    out["batch_compressed"] = [b.hex() for b in group.commit_batch(srs, flat, N, batch, ofmt=L.G1_ZCASH_COMPRESSED)]
    bufs = []
    for i in range(group.local_count):
        lo, hi = shard_range(N, group.rank(i), group.world)
        b = group.engine(i).alloc_scalars((hi - lo) * batch)
        b.upload(kzg_amd.pack_scalars([c for p in bp for c in p[lo:hi]]))
        bufs.append(b)
    out["batch_device"] = [b.hex() for b in group.commit_batch(srs, bufs, N, batch)]
    for b in bufs:
        b.free()
    # device-resident polynomials SHORTER than the SRS (4000 of 5003 coefficients): a rank holds [batch][its terms below 4000], the
    # last ranks hold fewer terms than their shard or none
    short, bufs = 4000, []
    for i in range(group.local_count):
        lo, hi = shard_range(N, group.rank(i), group.world)
        lo, hi = min(lo, short), min(hi, short)
        b = group.engine(i).alloc_scalars(max(hi - lo, 1) * batch)
        if hi > lo:
            b.upload(kzg_amd.pack_scalars([c for p in bp for c in p[lo:hi]]))
        bufs.append(b)
    out["batch_device_short"] = [b.hex() for b in group.commit_batch(srs, bufs, short, batch)]
    for b in bufs:
        b.free()
    # create_witness: on the polynomial, off it (the reference's error, after the exchange), a polynomial of one coefficient
    p = polys[N]
    x = rng.randrange(R)
    y = horner(p, x)
    out["witness"] = group.create_witness(srs, p, (x, y)).hex()
    try:
        group.create_witness(srs, p, (x, (y + 1) % R))
        out["witness_off_poly"] = "no error"
    except kzg_amd.PointNotOnPolynomial:
        out["witness_off_poly"] = "PointNotOnPolynomial"
    # create_witness_batched: 7 openings
    xs = [rng.randrange(R) for _ in range(7)]
    w, r = group.create_witness_batched(srs, p, [(v, horner(p, v)) for v in xs])
    out["witness_batched"] = [w.hex(), [hex(c) for c in r]]
    # device-resident whole polynomial on every local GPU
    whole = []
    for i in range(group.local_count):
        b = group.engine(i).alloc_scalars(N)
        b.upload(kzg_amd.pack_scalars(p))
        whole.append(b)
    out["witness_device"] = group.create_witness(srs, whole, (x, y)).hex()
    for b in whole:
        b.free()
    # evaluation form: the Lagrange-basis SRS sharded over the group
    lag_single = kzg_amd.setup_lagrange(group.engine(0), TAU, D)
    lag = group.upload(lag_single.download(), D)
    lag_single.free()
    from oracle import c_oracle as C
    pe = [rng.randrange(R) for _ in range(D)]
    evals = C.fft(pe)
    out["evals_sha"] = hashlib.sha256(kzg_amd.pack_scalars(evals)).hexdigest()
    out["witness_eval"] = {str(m): group.create_witness_eval(lag, evals, m).hex() for m in (0, 1, 777, D - 1)}
    out["commit_eval"] = group.commit(lag, evals).hex()
    lag.free()
    # --- failures of ONE rank: every rank must return that rank's error, and the group must stay in step afterwards ---
    small = polys[777]
    want_small = out["commit"]["777"]
    if i_fail:
        assert lib.kzg_test_mctx_inject_failure(group.handle, L.KZG_ERR_ALLOC) == 0
    try:
        group.commit(srs, small)
        out["local_failure"] = "no error"
    except Exception as e:  # noqa: BLE001
        out["local_failure"] = [type(e).__name__, getattr(e, "code", None), str(e)[:200]]
    out["after_local_failure"] = group.commit(srs, small).hex() == want_small
    # a resource failure BEFORE the exchange (growing the exchange buffers: 70 > the 64 partials held from the start)
    big = 70
    bflat = kzg_amd.pack_scalars([c for _ in range(big) for c in small])
    if i_fail:
        assert lib.kzg_test_mctx_inject_alloc_failure(group.handle) == 0
    try:
        group.commit_batch(srs, bflat, 777, big)
        out["alloc_failure"] = "no error"
    except Exception as e:  # noqa: BLE001
        out["alloc_failure"] = [type(e).__name__, getattr(e, "code", None), str(e)[:200]]
    # the ranks that did grow and the one that did not must still be in the same collective on the next calls
    got = group.commit_batch(srs, bflat, 777, big)
    out["after_alloc_failure"] = all(b.hex() == want_small for b in got) and len(got) == big
    out["after_alloc_failure_66"] = all(b.hex() == want_small for b in group.commit_batch(srs, bflat[:66 * 777 * 32], 777, 66))
    out["last_commit"] = group.commit(srs, polys[N]).hex()
    srs.free()
    # --- an SRS shorter than the group: ranks >= 3 hold EMPTY shards and contribute the identity to every exchange ---
    tiny = group.setup(TAU, 3)
    tp = [rng.randrange(R) for _ in range(3)]
    t = {"shards": [[tiny.shard(i)[1], len(tiny.shard(i)[0])] for i in range(group.local_count)],
         "commit": {str(m): group.commit(tiny, tp[:m]).hex() for m in (3, 2, 1, 0)}}
    tx = rng.randrange(R)
    t["witness"] = group.create_witness(tiny, tp, (tx, horner(tp, tx))).hex()
    for k in (1, 2):
        pts = [(v, horner(tp, v)) for v in [rng.randrange(R) for _ in range(k)]]
        w, r = group.create_witness_batched(tiny, tp, pts)
        t["batched_%d" % k] = [w.hex(), [hex(c) for c in r], [[hex(a), hex(b)] for a, b in pts]]
    out["tiny"] = t
    tiny.free()
    return out


def main():
    mode = sys.argv[1]
    if mode == "rank":
        rank, world, uid, seed = int(sys.argv[2]), int(sys.argv[3]), bytes.fromhex(sys.argv[4]), int(sys.argv[5])
        group = kzg_amd.DeviceGroup.for_rank(0, rank, world, uid)
    else:
        world, seed = int(sys.argv[2]), int(sys.argv[3])
        group = kzg_amd.DeviceGroup([0] * world)
    res = scenario(group, world - 1 if mode == "rank" else 0, seed)
    group.close()
    print("RESULT " + json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
