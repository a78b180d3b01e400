"""MSM parity: kzg_msm_g1 / kzg_commit_coeff through the C ABI vs the oracle (C restatement + known-tau)."""
import ctypes
import random

import pytest

import kzg_amd
from kzg_amd import _lib as L
from oracle import c_oracle as C
from oracle import kzg_model as M
from tests.gpu_common import rand_scalars

pytestmark = pytest.mark.gpu

TAU = 0x1234567_89abcdef


@pytest.fixture(scope="module")
def srs_small(engine):
    blob = C.setup_g1(TAU, 300)
    s = kzg_amd.Srs.upload(engine, blob, 300)
    yield s, blob
    s.free()


@pytest.mark.parametrize("n", [0, 1, 2, 3, 16, 255, 300])
def test_msm_small_sizes(engine, srs_small, n):
    srs, blob = srs_small
    rng = random.Random(100 + n)
    sc = rand_scalars(rng, n)
    got = engine.msm(srs, sc, n=n)
    assert got == C.msm_g1(blob[: 96 * n], sc)


def test_msm_edge_scalars(engine, srs_small):
    """zero scalars, r-1, 1, all-equal scalars (every term in one bucket), u64-valued scalars."""
    srs, blob = srs_small
    rng = random.Random(7)
    n = 200
    cases = {
        "zeros": [0] * n,
        "ones": [1] * n,
        "r_minus_1": [M.R - 1] * n,
        "all_equal": [rng.randrange(M.R)] * n,
        "mixed": [0, 1, M.R - 1, 2 ** 255 % M.R, (1 << 128) - 1] * (n // 5),
        "u64": rand_scalars(rng, n, "u64"),
        "half_boundary": [0x8000] * n,           # digit exactly 2^(c-1) for c = 16
        "carry_chain": [int("7fff" * 15, 16)] * n,
    }
    for name, sc in cases.items():
        assert engine.msm(srs, sc) == C.msm_g1(blob[: 96 * n], sc), name


def test_msm_offset_and_formats(engine, srs_small):
    srs, blob = srs_small
    rng = random.Random(8)
    n, off = 100, 37
    sc = rand_scalars(rng, n)
    want = C.msm_g1(blob[96 * off: 96 * (off + n)], sc)
    assert engine.msm(srs, sc, offset=off) == want
    P = C.blob_to_point(want)
    assert engine.msm(srs, sc, offset=off, ofmt=L.G1_ZCASH_COMPRESSED) == M.g1_to_compressed(P)
    assert engine.msm(srs, sc, offset=off, ofmt=L.G1_ZCASH_UNCOMPRESSED) == M.g1_to_uncompressed(P)
    jac = engine.msm(srs, sc, offset=off, ofmt=L.G1_JACOBIAN_MONT)
    rinv = pow(M.FQ_MONT_R, -1, M.Q)
    X, Y, Z = (int.from_bytes(jac[48 * i:48 * i + 48], "little") * rinv % M.Q for i in range(3))
    zi = pow(Z, -1, M.Q)
    assert (X * zi * zi % M.Q, Y * zi * zi * zi % M.Q) == P
    # identity result in every format
    z = [0] * n
    assert engine.msm(srs, z) == bytes(96)
    assert engine.msm(srs, z, ofmt=L.G1_ZCASH_COMPRESSED) == M.g1_to_compressed(None)
    assert engine.msm(srs, z, ofmt=L.G1_ZCASH_UNCOMPRESSED) == M.g1_to_uncompressed(None)
    # a lone host-bound result is converted on the host by default (emit.h compiled for the host); the GPU's k_emit_points
    # (option host_affine = 0; also what batches and device outputs use) must give the same bytes in every format
    # (Jacobian coordinates are not canonical -- the order of additions inside a bucket differs from run to run -- so that
    # format is compared as a point)
    def jac_point(b):
        X, Y, Z = (int.from_bytes(b[48 * i:48 * i + 48], "little") * rinv % M.Q for i in range(3))
        if Z == 0:
            return None
        zi = pow(Z, -1, M.Q)
        return (X * zi * zi % M.Q, Y * zi * zi * zi % M.Q)

    for fmt in (L.G1_AFFINE_MONT, L.G1_ZCASH_COMPRESSED, L.G1_ZCASH_UNCOMPRESSED, L.G1_JACOBIAN_MONT):
        norm = jac_point if fmt == L.G1_JACOBIAN_MONT else (lambda b: b)
        on_host = [norm(engine.msm(srs, v, offset=off, ofmt=fmt)) for v in (sc, z, [M.R - 1] * n)]
        engine.set_option("host_affine", 0)
        try:
            assert [norm(engine.msm(srs, v, offset=off, ofmt=fmt)) for v in (sc, z, [M.R - 1] * n)] == on_host, fmt
        finally:
            engine.set_option("host_affine", 1)
    assert jac_point(engine.msm(srs, sc, offset=off, ofmt=L.G1_JACOBIAN_MONT)) == P
    # montgomery-form scalars, device resident
    buf = engine.alloc_scalars(n, sfmt=L.FR_MONT)
    buf.upload(b"".join(M.fr_to_mont_le(s) for s in sc))
    assert engine.msm(srs, buf, offset=off) == want
    buf.free()
    with pytest.raises(kzg_amd.ReferencePanic):
        engine.msm(srs, sc, offset=250)  # 250 + 100 > 300: slice index panic in the reference


def test_srs_with_identity_and_repeated_points(engine):
    rng = random.Random(9)
    G = C.g1_generator()
    P = C.g1_mul(G, 777)
    nP = C.point_to_blob(M.g1_neg(C.blob_to_point(P)))
    pts = [P, P, nP, bytes(96), G, P, bytes(96), nP] * 8
    n = len(pts)
    blob = b"".join(pts)
    srs = kzg_amd.Srs.upload(engine, blob, n)
    for sc in ([5] * n, rand_scalars(rng, n), [1, 1, 2, 9, 0, M.R - 2, 3, 4] * 8):
        assert engine.msm(srs, sc) == C.msm_g1(blob, sc)
    srs.free()


@pytest.mark.parametrize("pfmt", [L.G1_ZCASH_UNCOMPRESSED, L.G1_ZCASH_COMPRESSED, L.G1_JACOBIAN_MONT])
def test_srs_upload_formats(engine, pfmt):
    rng = random.Random(10)
    n = 40
    blob = C.setup_g1(TAU, n)
    pts = [C.blob_to_point(blob[96 * i:96 * i + 96]) for i in range(n)]
    pts[5] = None
    if pfmt == L.G1_ZCASH_UNCOMPRESSED:
        raw = b"".join(M.g1_to_uncompressed(p) for p in pts)
    elif pfmt == L.G1_ZCASH_COMPRESSED:
        raw = b"".join(M.g1_to_compressed(p) for p in pts)
    else:
        # non-trivial Z: (x z^2, y z^3, z)
        out = []
        for p in pts:
            if p is None:
                out.append(bytes(144)); continue
            z = rng.randrange(1, M.Q)
            c = [p[0] * z * z % M.Q, p[1] * z * z * z % M.Q, z]
            out.append(b"".join((v * M.FQ_MONT_R % M.Q).to_bytes(48, "little") for v in c))
        raw = b"".join(out)
    srs = kzg_amd.Srs.upload(engine, raw, n, pfmt)
    want = b"".join(M.g1_to_affine_mont(p) for p in pts)
    assert srs.download() == want
    sc = rand_scalars(rng, n)
    assert engine.msm(srs, sc) == C.msm_g1(want, sc)
    srs.free()
    if pfmt != L.G1_JACOBIAN_MONT:
        # an x with x^3 + 4 a non-residue (compressed) / a y off the curve (uncompressed) must be rejected
        x = 5
        while pow((x ** 3 + 4) % M.Q, (M.Q - 1) // 2, M.Q) == 1:
            x += 1
        if pfmt == L.G1_ZCASH_COMPRESSED:
            bad_pt = bytearray(x.to_bytes(48, "big")); bad_pt[0] |= 0x80
        else:
            bad_pt = bytearray(M.g1_to_uncompressed(pts[0])); bad_pt[95] ^= 1
        bad = bytes(bad_pt) + raw[L.POINT_BYTES[pfmt]:]
        with pytest.raises(kzg_amd.EngineError):
            kzg_amd.Srs.upload(engine, bad, n, pfmt)
        # x >= q is not a canonical encoding
        big = bytearray((M.Q + 1).to_bytes(48, "big")); big[0] |= 0x80 if pfmt == L.G1_ZCASH_COMPRESSED else 0
        bad = bytes(big) + raw[48:]
        with pytest.raises(kzg_amd.EngineError):
            kzg_amd.Srs.upload(engine, bad, n, pfmt)


def test_setup_matches_reference_setup(engine):
    """setup(s, n) (src/lib.rs:38-55) on the GPU == the oracle's chain gs[i] = gs[i-1]*s."""
    params = kzg_amd.setup(engine, TAU, 130)
    assert params.gs.download() == C.setup_g1(TAU, 130)
    small = kzg_amd.setup(engine, TAU, 9)
    assert small.gs.download() == b"".join(M.g1_to_affine_mont(p) for p in M.setup_g1(TAU, 9))
    params.gs.free(); small.gs.free()


@pytest.mark.parametrize("log_n,kind", [(10, "full"), (12, "u64"), (14, "full")])
def test_commit_known_tau(engine, log_n, kind):
    """commit(p) == [p(tau)]G  (SURVEY 8c), config-1 style at 2^10 and larger."""
    n = 1 << log_n
    rng = random.Random(log_n)
    params = kzg_amd.setup(engine, TAU, n)
    coeffs = rand_scalars(rng, n, kind)
    prover = kzg_amd.KZGProver(params)
    poly = kzg_amd.Polynomial(coeffs)
    got = prover.commit(poly)
    want = C.g1_mul(C.g1_generator(), C.poly_eval(coeffs, TAU))
    assert got == want
    if log_n <= 12:
        assert got == C.msm_g1(params.gs.download(), coeffs)
    assert kzg_amd.KZGVerifier(params).verify_poly(got, poly)
    coeffs2 = list(coeffs); coeffs2[2] = (coeffs2[2] + 1) % M.R
    assert not kzg_amd.KZGVerifier(params).verify_poly(got, kzg_amd.Polynomial(coeffs2))
    with pytest.raises(kzg_amd.ReferencePanic):
        prover.commit(kzg_amd.Polynomial(coeffs + [1]))  # longer than the SRS
    params.gs.free()


def test_msm_batch_and_sum(engine, srs_small):
    srs, blob = srs_small
    rng = random.Random(21)
    n, batch = 128, 6
    vecs = [rand_scalars(rng, n) for _ in range(batch)]
    flat = [s for v in vecs for s in v]
    got = engine.msm_batch(srs, flat, n, batch)
    want = [C.msm_g1(blob[: 96 * n], v) for v in vecs]
    assert got == want
    total = engine.g1_sum(want)
    acc = bytes(96)
    for w in want:
        acc = C.g1_add(acc, w)
    assert total == acc
    assert engine.g1_sum([]) == bytes(96)
    assert engine.g1_sum([want[0], C.point_to_blob(M.g1_neg(C.blob_to_point(want[0])))]) == bytes(96)


def test_msm_batch_pipeline_paths(engine):
    """kzg_msm_g1_batch in its pipelined form (more MSMs than lanes, accumulation kernels on the dedicated FIFO streams) on
    adversarial scalar sets that drive the rare paths: all-equal scalars (one bucket per window: the overflow slices of
    k_fold_overflow and their last-arrival sums), all-zero, u64-valued, and uniform ones -- each vector against the known-tau
    identity; under every pipeline plan the queue planner can choose (18 / 4 / 3 / 1 hardware queues assumed, accumulation
    streams on and off) and with the latency-mode tail kernels on and off for single MSMs."""
    rng = random.Random(2024)
    n, lanes = 1 << 12, 4
    params = kzg_amd.setup(engine, TAU, n)
    srs = params.gs
    vecs = [rand_scalars(rng, n) for _ in range(5)]
    vecs += [[rng.randrange(M.R)] * n, [0] * n, rand_scalars(rng, n, "u64"), [M.R - 1] * n, [1] + [0] * (n - 1)]
    vecs += [rand_scalars(rng, n) for _ in range(3)]                     # 13 MSMs over 4 lanes: every lane re-used 3 times
    flat = [s for v in vecs for s in v]
    want = [C.g1_mul(C.g1_generator(), C.poly_eval(v, TAU)) for v in vecs]
    engine.set_option("streams", lanes)
    try:
        for accum_streams, hw_queues in ((2, 0), (1, 0), (0, 0), (2, 4), (2, 3), (2, 1), (1, 5)):
            engine.set_option("accum_streams", accum_streams)
            engine.set_option("hw_queues", hw_queues)
            assert engine.msm_batch(srs, flat, n, len(vecs)) == want, (accum_streams, hw_queues)
        for quads in (0, 1):
            engine.set_option("tail_quads", quads)
            assert [engine.msm(srs, v) for v in vecs[4:10]] == want[4:10], quads
    finally:
        engine.set_option("tail_quads", 1)
        engine.set_option("hw_queues", 0)
        engine.set_option("accum_streams", 2)
        engine.set_option("streams", 8)
    srs.free()


@pytest.mark.parametrize("single_pass", [0, 1])
def test_c17_sort_ragged_sizes(engine, single_pass):
    """The production window width (c = 17, chosen by option at a small size) over ragged term counts around the chunk sizes of
    the two-level sort (1024 scalars per level-1 chunk, 2048 records per level-2 chunk, one sort block per 2048 scalars), with
    uniform, all-equal (every record of a window in one bin), zero and r - 1 scalars; both sort implementations."""
    rng = random.Random(17 + single_pass)
    n = 6200
    engine.set_option("window_bits", 17)
    engine.set_option("sort_single_pass", single_pass)
    try:
        params = kzg_amd.setup(engine, TAU, n, g2_len=0)
        assert params.gs.window_info() == (17, 15)
        G = C.g1_generator()
        for k in (1, 2, 63, 64, 1023, 1024, 1025, 2047, 2048, 2049, 4097, 6200):
            for sc in (rand_scalars(rng, k), [rng.randrange(M.R)] * k, [M.R - 1] * k):
                assert engine.msm(params.gs, sc) == C.g1_mul(G, C.poly_eval(sc, TAU)), (k, single_pass)
        assert engine.msm(params.gs, [0] * 3000) == bytes(96)
        sub = rand_scalars(rng, 2100)
        off = 4000
        want = C.g1_mul(G, C.poly_eval(sub, TAU) * pow(TAU, off, M.R) % M.R)
        assert engine.msm(params.gs, sub, offset=off) == want
        params.gs.free()
    finally:
        engine.set_option("sort_single_pass", 0)
        engine.set_option("window_bits", 0)


def test_sharded_srs_partials_sum_to_full_commit(engine):
    """Multi-GPU data path on one GPU: 4 contiguous SRS shards (kzg_srs_setup_g1_shard), one partial MSM
    each, kzg_g1_sum of the partials == commit against the full SRS == [p(tau)]G."""
    from kzg_amd.distributed import shard_range
    rng = random.Random(77)
    n, world = 1 << 12, 4
    coeffs = rand_scalars(rng, n)
    full = kzg_amd.setup(engine, TAU, n)
    want = kzg_amd.KZGProver(full).commit(kzg_amd.Polynomial(coeffs))
    assert want == C.g1_mul(C.g1_generator(), C.poly_eval(coeffs, TAU))
    parts = []
    for r in range(world):
        lo, hi = shard_range(n, r, world)
        shard = kzg_amd.setup_shard(engine, TAU, lo, hi - lo)
        assert shard.download() == full.gs.download(lo, hi - lo)
        parts.append(engine.msm(shard, coeffs[lo:hi]))
        shard.free()
    assert engine.g1_sum(parts) == want
    full.gs.free()


def test_g1_sum_batch(engine, srs_small):
    srs, blob = srs_small
    pts = [blob[96 * i: 96 * (i + 1)] for i in range(12)]
    got = engine.g1_sum_batch(pts, 3, 4)
    for g in range(4):
        acc = bytes(96)
        for i in range(3):
            acc = C.g1_add(acc, pts[3 * g + i])
        assert got[g] == acc
    # 4 groups or fewer bound for the host are converted on the calling thread; 6 groups, and host_affine = 0, on the GPU
    six = engine.g1_sum_batch(pts, 2, 6)
    assert six == [C.g1_add(pts[2 * g], pts[2 * g + 1]) for g in range(6)]
    engine.set_option("host_affine", 0)
    try:
        assert engine.g1_sum_batch(pts, 3, 4) == got
    finally:
        engine.set_option("host_affine", 1)


def test_concurrent_callers(engine, srs_small):
    """The C ABI is thread-safe: many host threads on one kzg_ctx (serialised on its mutex) and on separate
    contexts (truly concurrent on the GPU) all get the right answers."""
    import threading
    srs, blob = srs_small
    rng = random.Random(55)
    jobs = [rand_scalars(rng, 64 + 8 * i) for i in range(8)]
    want = [C.msm_g1(blob[: 96 * len(j)], j) for j in jobs]
    got = [None] * len(jobs)
    errs = []

    def work(i, eng, s):
        try:
            for _ in range(3):
                got[i] = eng.msm(s, jobs[i])
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=work, args=(i, engine, srs)) for i in range(len(jobs))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs and got == want
    # separate contexts on the same GPU
    engines = [kzg_amd.Engine(0) for _ in range(3)]
    srss = [kzg_amd.Srs.upload(e, blob, 300) for e in engines]
    got = [None] * len(jobs)
    ths = [threading.Thread(target=work, args=(i, engines[i % 3], srss[i % 3])) for i in range(len(jobs))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs and got == want
    for e, s in zip(engines, srss):
        s.free()
        e.close()


@pytest.mark.parametrize("log_n", [12, 16, 17])
def test_small_msm_pipeline_shapes_agree(engine, log_n):
    """MSMs of up to 2^17 points inside a pipeline take a 160-block accumulation grid on four accumulation streams (round 4); the old
    shape (accum_blocks_small = 0: the batch grid on two streams), other grids, a narrowed stream plan and the lone call must all give
    the same bytes -- uniform, u64-valued, all-equal and sparse coefficients -- each equal to the oracle's [p(tau)]G."""
    import kzg_amd
    from kzg_amd import _lib as L
    n, batch, tau = 1 << log_n, 36, 0x5151ABCD
    params = kzg_amd.setup(engine, tau, n, g2_len=0)
    scal = engine.alloc_scalars(n * batch)
    views = []
    for b in range(batch):
        v = kzg_amd.DeviceBuffer.__new__(kzg_amd.DeviceBuffer)
        v.engine, v.n, v.sfmt, v.ptr = engine, n, scal.sfmt, ctypes.c_void_p(scal.ptr.value + 32 * n * b)
        views.append(v)
        if b % 4 == 0:
            v.fill_random(900 + b)
        elif b % 4 == 1:
            v.fill_random(900 + b, u64_valued=True)
        elif b % 4 == 2:
            v.upload(M.fr_to_le((0xABCDEF0123456789 << 100) % M.R) * n)                     # all equal: one bucket per window
        else:
            v.upload(b"".join(M.fr_to_le(7 if i % 997 == 0 else 0) for i in range(n)))      # sparse
    G = C.g1_generator()
    want = [C.g1_mul(G, C.poly_eval_bytes(v.download(), n, tau)) for v in views[:8]]
    try:
        ref = engine.msm_batch(params.gs, scal, n, batch)
        assert ref[:8] == want
        for opts in ({"accum_blocks_small": 0}, {"accum_blocks_small": 64}, {"accum_blocks_small": 256, "accum_streams_small": 3},
                     {"small_entries": 0}, {"accum_streams": 1, "accum_streams_small": 0}, {"defer_tail": 0}):
            for k, val in opts.items():
                engine.set_option(k, val)
            assert engine.msm_batch(params.gs, scal, n, batch) == ref, opts
            for k, val in {"accum_blocks_small": 160, "accum_streams_small": 4, "small_entries": 2 << 20, "accum_streams": 2, "defer_tail": 1}.items():
                engine.set_option(k, val)
        assert [engine.msm(params.gs, v) for v in views[:6]] == ref[:6]                     # lone calls
    finally:
        for k, val in {"accum_blocks_small": 160, "accum_streams_small": 4, "small_entries": 2 << 20, "accum_streams": 2, "defer_tail": 1}.items():
            engine.set_option(k, val)
        scal.free()
        params.gs.free()
