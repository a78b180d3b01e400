"""The multi-GPU commit path behind the C ABI (kzg_mctx / kzg_msrs, kzg_amd/csrc/mgpu.hip) on however many GPUs the box has
(one on the test box: a group of one, with the RCCL all-gather forced on so that the exchange code runs).  Parity: the group's
commitment equals the single-GPU commitment, the oracle's Pippenger and [p(tau)]G; the 8-way sharding of configs[4] is
exercised shard by shard in tests/test_gpu_fullsize.py."""
import ctypes
import random

import pytest

import kzg_amd
from kzg_amd import _lib as L
from kzg_amd.distributed import shard_range
from oracle import c_oracle as C
from oracle import kzg_model as M
from tests.gpu_common import engine, rand_scalars  # noqa: F401

pytestmark = pytest.mark.gpu

TAU = 0x0BADC0FFEE123457


@pytest.fixture(scope="module")
def group():
    lib = L.load()
    ndev = lib.kzg_device_count()
    assert ndev >= 1
    g = kzg_amd.DeviceGroup(list(range(min(ndev, 8))))
    yield g
    g.close()


@pytest.mark.parametrize("always_gather", [0, 1])
def test_group_commit_matches_single_gpu_and_oracle(engine, group, always_gather):
    group.set_option("always_gather", always_gather)   # 1: ncclAllGather runs even in a group of one
    rng = random.Random(11)
    n = 1000
    srs = group.setup(TAU, n)
    assert len(srs) == n
    blob = C.setup_g1(TAU, n)
    # every resident shard holds exactly its contiguous range of setup(tau, n).gs
    for i in range(group.local_count):
        shard, first = srs.shard(i)
        lo, hi = shard_range(n, group.rank(i), group.world)
        assert first == lo and len(shard) == hi - lo
        assert shard.download() == blob[96 * lo: 96 * hi]
    for m in (n, 777, 1, 0):        # polynomials shorter than the SRS only reach the first shards
        coeffs = rand_scalars(rng, m)
        got = group.commit(srs, coeffs)
        assert got == C.msm_g1(blob[: 96 * m], coeffs), m
        assert got == C.g1_mul(C.g1_generator(), C.poly_eval(coeffs, TAU) if m else 0), m
    with pytest.raises(kzg_amd.ReferencePanic):          # polynomial longer than the SRS: the slice index panic
        group.commit(srs, rand_scalars(rng, n + 1))
    srs.free()
    group.set_option("always_gather", 0)


def test_group_commit_batch_host_and_device_resident(engine, group):
    group.set_option("always_gather", 1)
    rng = random.Random(12)
    n, batch = 4096 + 37, 5
    srs = group.setup(TAU, n)
    polys = [rand_scalars(rng, n) for _ in range(batch - 2)] + [[0] * n, [M.R - 1] * n]
    want = [C.g1_mul(C.g1_generator(), C.poly_eval(p, TAU)) for p in polys]
    flat = kzg_amd.pack_scalars([c for p in polys for c in p])
    assert group.commit_batch(srs, flat, n, batch) == want
    # compressed output format through the group path
    got48 = group.commit_batch(srs, flat, n, batch, ofmt=L.G1_ZCASH_COMPRESSED)
    assert [M.g1_to_compressed(C.blob_to_point(w)) for w in want] == got48
    # device-resident slices: per local GPU a [batch][shard] array
    bufs = []
    for i in range(group.local_count):
        lo, hi = shard_range(n, group.rank(i), group.world)
        e = group.engine(i)
        b = e.alloc_scalars((hi - lo) * batch)
        b.upload(kzg_amd.pack_scalars([c for p in polys for c in p[lo:hi]]))
        bufs.append(b)
    assert group.commit_batch(srs, bufs, n, batch) == want
    for b in bufs:
        b.free()
    srs.free()
    group.set_option("always_gather", 0)


def test_group_upload_and_witness(engine, group):
    group.set_option("always_gather", 1)
    rng = random.Random(13)
    n = 600
    blob = C.setup_g1(TAU, n)
    srs = group.upload(blob, n)
    coeffs = rand_scalars(rng, n)
    assert group.commit(srs, coeffs) == C.msm_g1(blob, coeffs)
    x = rng.randrange(M.R)
    y = C.poly_eval(coeffs, x)
    w = group.create_witness(srs, coeffs, (x, y))
    ptau = C.poly_eval(coeffs, TAU)
    assert w == C.g1_mul(C.g1_generator(), (ptau - y) * M.fr_inv(TAU - x) % M.R)
    with pytest.raises(kzg_amd.PointNotOnPolynomial):
        group.create_witness(srs, coeffs, (x, (y + 1) % M.R))
    # degree-0 polynomial: empty quotient -> identity (src/coeff_form.rs:332-341 edge)
    assert group.create_witness(srs, [5], (3, 5)) == bytes(96)
    srs.free()
    group.set_option("always_gather", 0)


def test_group_per_process_mode_world1(engine):
    """kzg_mctx_unique_id + kzg_mctx_create_rank (the one-process-per-GPU entry): a world of one rank."""
    uid = kzg_amd.DeviceGroup.unique_id()
    assert len(uid) == 128 and any(uid)
    g = kzg_amd.DeviceGroup.for_rank(0, 0, 1, uid)
    assert g.world == 1 and g.local_count == 1 and g.rank(0) == 0
    g.set_option("always_gather", 1)     # ncclCommInitRank + ncclAllGather in a world of one
    rng = random.Random(14)
    n = 300
    srs = g.setup(TAU, n)
    coeffs = rand_scalars(rng, n)
    assert g.commit(srs, coeffs) == C.g1_mul(C.g1_generator(), C.poly_eval(coeffs, TAU))
    srs.free()
    g.close()


def test_group_rejects_bad_arguments():
    lib = L.load()
    h = ctypes.c_void_p()
    arr = (ctypes.c_int * 2)(0, 0)
    assert lib.kzg_mctx_create(arr, 2, ctypes.byref(h)) == L.KZG_ERR_SHAPE      # duplicate device
    assert lib.kzg_mctx_create(arr, 0, ctypes.byref(h)) == L.KZG_ERR_SHAPE
    assert lib.kzg_mctx_create_rank(0, 3, 2, None, ctypes.byref(h)) == L.KZG_ERR_SHAPE
    arr = (ctypes.c_int * 1)(99)
    assert lib.kzg_mctx_create(arr, 1, ctypes.byref(h)) == L.KZG_ERR_NO_DEVICE
