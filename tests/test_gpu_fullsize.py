"""Parity at BASELINE.json's full sizes through size-independent properties (known-tau identities, eval-form ==
coeff-form, linearity, round trips), plus adversarial scalar distributions that force the deep paths of the MSM
(multi-level bucket folding, single-bucket inputs, every window size).

Every expected value comes from the ORACLE: the coefficient bytes are downloaded and p(tau), p(x), I(tau) are computed by
oracle/kzg_oracle.c's Horner loop (`oeval`, ~0.1 s per 2^20 evaluation), then one oracle scalar multiplication.  No right-hand
side below touches a HIP kernel; the engine's own kzg_poly_eval is checked as a value under test (test_poly_eval_2_20)."""
import ctypes
import random

import pytest

import kzg_amd
from kzg_amd import _lib as L
from oracle import c_oracle as C
from oracle import kzg_model as M
from tests.gpu_common import rand_scalars

pytestmark = pytest.mark.gpu
R = M.R
TAU = 0x5EED5EED5EED5EED


@pytest.fixture(scope="module")
def big(engine):
    n = 1 << 20
    params = kzg_amd.setup(engine, TAU, n)
    lag = kzg_amd.setup_lagrange(engine, TAU, n)
    yield n, params, lag
    params.gs.free()
    lag.free()


_dl_cache = {}


def oeval(p, x):
    """p(x) by the oracle.  p: a DeviceBuffer (downloaded once, cached by address) or a list of ints."""
    if isinstance(p, kzg_amd.DeviceBuffer):
        key = (p.ptr.value, p.n)
        if key not in _dl_cache:
            if len(_dl_cache) > 4:
                _dl_cache.clear()
            _dl_cache[key] = p.download()
        return C.poly_eval_bytes(_dl_cache[key], p.n, x)
    return C.poly_eval(p, x)


def _fresh(buf):
    """forget a cached download (the buffer was rewritten or freed)"""
    _dl_cache.pop((buf.ptr.value, buf.n), None)


def _msm_dev(engine, srs, buf, n, offset=0):
    out = ctypes.create_string_buffer(96)
    rc = engine.lib.kzg_msm_g1(engine.ctx, srs.handle, offset, buf.ptr, n, buf.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
    assert rc == 0, engine.last_error()
    return out.raw


def test_config2_commit_2_20(engine, big):
    """configs[1]: degree-2^20 coeff-form commit == [p(tau)]G, full-width and u64-valued coefficients."""
    n, params, _ = big
    for seed, u64 in ((11, False), (12, True)):
        buf = engine.alloc_scalars(n).fill_random(seed, u64_valued=u64)
        got = _msm_dev(engine, params.gs, buf, n)
        assert got == C.g1_mul(C.g1_generator(), oeval(buf, TAU))
        _fresh(buf); buf.free()


def test_poly_eval_2_20(engine, big):
    """kzg_poly_eval (the Horner-scan kernels) as a value under test: equal to the oracle's evaluation at 2^20."""
    n, _, _ = big
    buf = engine.alloc_scalars(n).fill_random(15)
    for x in (TAU, 0, 1, R - 1, kzg_amd.splitmix_scalar(3, 3)):
        assert engine.poly_eval(buf, x) == oeval(buf, x)
    _fresh(buf)
    _fresh(buf); buf.free()


def test_sort_variants_and_batch_tail_agree_2_20(engine, big):
    """configs[1] at full size through every variant of the pipeline around the accumulation kernel: two-level sort (default)
    and single-pass sort, latency-mode tail on and off, a lone MSM and a deep batch (work-efficient tail) -- the same bytes from
    all of them, for full-width, u64-valued and all-equal coefficients (all-equal: every record of a window in ONE bin of the
    two-level sort and one bucket: the overflow slices), each equal to [p(tau)]G."""
    n, params, _ = big
    G = C.g1_generator()
    bufs = [engine.alloc_scalars(n).fill_random(31), engine.alloc_scalars(n).fill_random(32, u64_valued=True)]
    eq = engine.alloc_scalars(n)
    eq.upload(M.fr_to_le(0x1234567890ABCDEF1122334455667788990011223344556677889900AABBCCDD % R) * n)
    bufs.append(eq)
    want = [C.g1_mul(G, oeval(b, TAU)) for b in bufs]
    try:
        for single_pass in (0, 1):
            engine.set_option("sort_single_pass", single_pass)
            for quads in (1, 0):
                engine.set_option("tail_quads", quads)
                assert [_msm_dev(engine, params.gs, b, n) for b in bufs] == want, (single_pass, quads)
        engine.set_option("sort_single_pass", 0)
        engine.set_option("tail_quads", 1)
        # deep batch: 6 MSMs over 2 lanes (>= 2 per lane: the work-efficient tail), all against the first buffer's polynomial
        engine.set_option("streams", 2)
        rep = engine.alloc_scalars(6 * n)
        for j in range(6):
            hipcpy = bufs[j % 3].download()
            v = kzg_amd.DeviceBuffer.__new__(kzg_amd.DeviceBuffer)
            v.engine, v.n, v.sfmt, v.ptr = engine, n, rep.sfmt, ctypes.c_void_p(rep.ptr.value + 32 * n * j)
            v.upload(hipcpy)
        # (deferred tails, round 4: with more MSMs than lanes a lane's tail is enqueued after the sort of its next MSM -- two in flight
        # per lane, the arena's two halves; and the same batch with every tail right behind its accumulation)
        for defer in (1, 0):
            engine.set_option("defer_tail", defer)
            out = ctypes.create_string_buffer(96 * 6)
            rc = engine.lib.kzg_msm_g1_batch(engine.ctx, params.gs.handle, 0, rep.ptr, n, 6, rep.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
            assert rc == 0, engine.last_error()
            assert [out.raw[96 * j:96 * j + 96] for j in range(6)] == [want[j % 3] for j in range(6)], defer
        rep.free()
    finally:
        engine.set_option("sort_single_pass", 0)
        engine.set_option("tail_quads", 1)
        engine.set_option("defer_tail", 1)
        engine.set_option("streams", 8)
    for b in bufs:
        _fresh(b); b.free()


def test_config3_eval_form_equals_coeff_form_2_20(engine, big):
    """configs[2]: NTT then Lagrange-SRS MSM gives the same commitment as the monomial MSM; iNTT round trip."""
    n, params, lag = big
    buf = engine.alloc_scalars(n).fill_random(13)
    orig = buf.download()
    c1 = _msm_dev(engine, params.gs, buf, n)
    assert c1 == C.g1_mul(C.g1_generator(), C.poly_eval_bytes(orig, n, TAU))      # the oracle pins what the two forms agree on
    engine.ntt(buf, 20)
    _fresh(buf)
    c2 = _msm_dev(engine, lag, buf, n)
    assert c1 == c2
    ok = ctypes.c_int()
    rc = engine.lib.kzg_verify_poly_eval(engine.ctx, params.gs.handle, c1, L.G1_AFFINE_MONT, buf.ptr, n, buf.sfmt, L.IN_DEVICE, ctypes.byref(ok))
    assert rc == 0 and ok.value == 1
    engine.ntt(buf, 20, inverse=True)
    assert buf.download() == orig
    _fresh(buf); buf.free()


def test_config4_witnesses_2_20(engine, big):
    """configs[3]: create_witness (single opening) and create_witness_batched (k = 256) at degree 2^20."""
    n, params, lag = big
    buf = engine.alloc_scalars(n).fill_random(14)
    ptau = oeval(buf, TAU)
    x = kzg_amd.splitmix_scalar(99, 0)
    y = oeval(buf, x)
    out = ctypes.create_string_buffer(96)
    b32 = lambda v: (v % R).to_bytes(32, "little")  # noqa: E731
    rc = engine.lib.kzg_witness_coeff(engine.ctx, params.gs.handle, buf.ptr, n, b32(x), b32(y), buf.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
    assert rc == 0 and out.raw == C.g1_mul(C.g1_generator(), (ptau - y) * pow(TAU - x, -1, R) % R)
    rc = engine.lib.kzg_witness_coeff(engine.ctx, params.gs.handle, buf.ptr, n, b32(x), b32(y + 1), buf.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
    assert rc == L.KZG_ERR_POINT_NOT_ON_POLY
    k = 256
    xs = [kzg_amd.splitmix_scalar(7, i) for i in range(k)]
    ys = [oeval(buf, v) for v in xs]
    rbuf, rlen = ctypes.create_string_buffer(32 * k), ctypes.c_size_t()
    rc = engine.lib.kzg_witness_coeff_batched(engine.ctx, params.gs.handle, buf.ptr, n, kzg_amd.pack_scalars(xs), kzg_amd.pack_scalars(ys),
                                              k, buf.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT, rbuf, ctypes.byref(rlen))
    assert rc == 0 and rlen.value == k
    I = kzg_amd.unpack_scalars(rbuf.raw)
    assert all(oeval(I, xs[i]) == ys[i] for i in (0, 17, 255))
    Z = 1
    for v in xs:
        Z = Z * (TAU - v) % R
    assert out.raw == C.g1_mul(C.g1_generator(), (ptau - oeval(I, TAU)) * pow(Z, -1, R) % R)
    # eval-form witness at index m == coeff-form witness at w^m
    m = 54321
    xm = pow(kzg_amd.compute_omega(n)[2], m, R)
    ym = oeval(buf, xm)
    rc = engine.lib.kzg_witness_coeff(engine.ctx, params.gs.handle, buf.ptr, n, b32(xm), b32(ym), buf.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
    w_coeff = out.raw
    engine.ntt(buf, 20)
    _fresh(buf)
    rc2 = engine.lib.kzg_witness_eval(engine.ctx, lag.handle, buf.ptr, n, m, buf.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
    assert rc == 0 and rc2 == 0 and out.raw == w_coeff
    _fresh(buf); buf.free()


def test_commit_and_witness_2_23_default_window20(engine):
    """The size from which the engine picks 20-bit windows by itself: commit, a single-point witness (MSM of n - 1 terms, the
    error path) and a blocking call from a device buffer with an offset sub-range, each against the known-tau identity."""
    n = 1 << 23
    params = kzg_amd.setup(engine, TAU, n, g2_len=0)
    assert params.gs.window_info() == (20, 13)
    buf = engine.alloc_scalars(n).fill_random(23)
    ptau = oeval(buf, TAU)
    G = C.g1_generator()
    out = ctypes.create_string_buffer(96)
    rc = engine.lib.kzg_commit_coeff(engine.ctx, params.gs.handle, buf.ptr, n, buf.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
    assert rc == 0 and out.raw == C.g1_mul(G, ptau)
    x = kzg_amd.splitmix_scalar(23, 1)
    y = oeval(buf, x)
    b32 = lambda v: (v % R).to_bytes(32, "little")  # noqa: E731
    rc = engine.lib.kzg_witness_coeff(engine.ctx, params.gs.handle, buf.ptr, n, b32(x), b32(y), buf.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
    assert rc == 0 and out.raw == C.g1_mul(G, (ptau - y) * pow(TAU - x, -1, R) % R)
    rc = engine.lib.kzg_witness_coeff(engine.ctx, params.gs.handle, buf.ptr, n, b32(x), b32(y + 1), buf.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
    assert rc == L.KZG_ERR_POINT_NOT_ON_POLY
    off, m = 1234567, (1 << 22) + 89
    sub = _view(engine, buf, 0, m)
    rc = engine.lib.kzg_msm_g1(engine.ctx, params.gs.handle, off, sub.ptr, m, buf.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
    assert rc == 0 and out.raw == C.g1_mul(G, pow(TAU, off, R) * oeval(sub, TAU) % R)
    params.gs.free()
    _fresh(buf); buf.free()


def test_msm_linearity_and_offsets_2_20(engine, big):
    """MSM(a) + MSM(b) == MSM(a + b); MSM over [0, n) == MSM[0, h) + MSM[h, n)."""
    n, params, _ = big
    a = engine.alloc_scalars(n).fill_random(21)
    b = engine.alloc_scalars(n).fill_random(22)
    ca, cb = _msm_dev(engine, params.gs, a, n), _msm_dev(engine, params.gs, b, n)
    # a + b evaluated through the known-tau identity: p_a(tau) + p_b(tau)
    s = (oeval(a, TAU) + oeval(b, TAU)) % R
    assert engine.g1_sum([ca, cb]) == C.g1_mul(C.g1_generator(), s)
    h = 333_333
    lo = _msm_dev(engine, params.gs, a, h)
    hi_buf = kzg_amd.DeviceBuffer.__new__(kzg_amd.DeviceBuffer)
    hi_buf.engine, hi_buf.n, hi_buf.sfmt, hi_buf.ptr = engine, n - h, a.sfmt, ctypes.c_void_p(a.ptr.value + 32 * h)
    hi = _msm_dev(engine, params.gs, hi_buf, n - h, offset=h)
    assert engine.g1_sum([lo, hi]) == ca
    _fresh(a); _fresh(b); a.free(); b.free()


@pytest.mark.parametrize("case", ["all_equal", "two_values", "zeros_and_ones", "single_window", "max_digits"])
def test_adversarial_scalars_2_16(engine, case):
    """Distributions that put everything into a handful of buckets: exercises several fold rounds."""
    n = 1 << 16
    rng = random.Random(hash(case) & 0xFFFF)
    params = kzg_amd.setup(engine, TAU, n)
    if case == "all_equal":
        sc = [rng.randrange(R)] * n
    elif case == "two_values":
        v = [rng.randrange(R), R - 1]
        sc = [v[i & 1] for i in range(n)]
    elif case == "zeros_and_ones":
        sc = [i % 3 % 2 for i in range(n)]
    elif case == "single_window":
        sc = [0x1234 << 48] * n
    else:  # every signed digit at its extreme
        sc = [int("8000" * 15, 16) % R] * n
    got = engine.msm(params.gs, sc)
    ptau = 0
    # sum_i sc_i tau^i with few distinct values: group by value
    pw = 1
    acc = {}
    for i in range(n):
        acc[sc[i]] = (acc.get(sc[i], 0) + pw) % R
        pw = pw * TAU % R
    ptau = sum(v * s for v, s in acc.items()) % R
    assert got == C.g1_mul(C.g1_generator(), ptau)
    params.gs.free()


@pytest.mark.parametrize("c", [4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15])
def test_every_window_size(engine, c):
    """Every window width of the single-pass pipeline (c = 4 .. 15; 16 .. 20 in test_wide_windows), chosen by option, at a size
    where each has many partials per bucket; full-width, u64-valued and all-equal scalars (widths whose top window holds only
    a few bits -- 9, 11, 12, 14, 15 -- put every scalar into a handful of buckets there: the fold's overflow slices)."""
    n = 3000 + 97 * c
    rng = random.Random(c)
    engine.set_option("window_bits", c)
    try:
        params = kzg_amd.setup(engine, TAU, n)
        cc, W = params.gs.window_info()
        assert cc == c and W == -(-256 // c)
        for sc in (rand_scalars(rng, n), rand_scalars(rng, n, "u64"), [rng.randrange(R)] * n):
            assert engine.msm(params.gs, sc) == C.g1_mul(C.g1_generator(), C.poly_eval(sc, TAU)), c
        params.gs.free()
    finally:
        engine.set_option("window_bits", 0)


@pytest.mark.parametrize("wb,naf", [(20, 0), (17, 0), (17, 18)])
def test_heavy_bins_2_18(engine, wb, naf):
    """Level 2 of the two-level sorts cuts a bin that holds far more than its share into slices sorted by separate blocks: u64-valued
    scalars (at 20 bits the whole top window lands in the first 17 buckets = bin 0), all-equal scalars (13 / 15 buckets hold
    everything), bits and bytes (a handful of buckets of bin 0), a mix with random ones -- each against the known-tau identity, for
    the 20-bit and the 17-bit windows and the positional tables."""
    n = 1 << 18
    rng = random.Random(wb)
    engine.set_option("window_bits", wb)
    engine.set_option("naf_window", naf)
    try:
        params = kzg_amd.setup(engine, TAU, n, g2_len=0)
        assert params.gs.window_info() == ((18, 15) if naf else (wb, 13 if wb == 20 else 15))
        pw = [1]
        for _ in range(n - 1):
            pw.append(pw[-1] * TAU % R)
        G = C.g1_generator()
        eq = rng.randrange(R)
        cases = {
            "u64": rand_scalars(rng, n, "u64"),
            "all_equal": [eq] * n,
            "mixed": [eq if i % 3 == 0 else rng.getrandbits(64) if i % 3 == 1 else rng.randrange(R) for i in range(n)],
            "small": [rng.randrange(1, 1 << 12) for _ in range(n)],
            "bits": [rng.randrange(2) for _ in range(n)],
            "bytes": [rng.randrange(256) for _ in range(n)],
            "all_ones": [1] * n,
        }
        want = {name: C.g1_mul(G, sum(s * p for s, p in zip(sc, pw)) % R) for name, sc in cases.items()}
        for mode in (1, 2):             # slices always / never (every bin sorted by one block)
            engine.set_option("heavy_bins", mode)
            for name, sc in cases.items():
                assert engine.msm(params.gs, sc) == want[name], (name, mode)
        # adaptive (the default; setting the option clears the history): the slice kernels are enqueued for the next 64 MSMs after
        # one that met an oversized bin
        engine.set_option("heavy_bins", 0)
        uniform = rand_scalars(rng, n)
        want_u = C.g1_mul(G, sum(s * p for s, p in zip(uniform, pw)) % R)
        engine.prof_reset(); engine.prof_enable(True)
        launched = []
        for sc, w in ((uniform, want_u), (cases["bits"], want["bits"]), (cases["all_ones"], want["all_ones"]), (uniform, want_u)):
            assert engine.msm(params.gs, sc) == w
            launched.append(engine.prof_get("k_heavy_place")[0])
        if naf:     # the short top digit of a NAF makes an oversized bin out of uniform scalars too
            assert launched == [0, 1, 2, 3]
        else:       # uniform: nothing; bits: planned with nothing seen yet; then in slices, uniform ones included (idle)
            assert launched == [0, 0, 1, 2]
        engine.set_option("heavy_bins", 0)
        assert engine.msm(params.gs, uniform) == want_u
        assert engine.prof_get("k_heavy_place")[0] == launched[-1]
        engine.prof_enable(False)
        params.gs.free()
    finally:
        engine.prof_enable(False)
        engine.set_option("heavy_bins", 0)
        engine.set_option("window_bits", 0)
        engine.set_option("naf_window", 0)


def test_heavy_bins_2_22_wide_records(engine):
    """The same with the 8-byte level-1 records (17-bit windows above 2^21 points: the table index no longer fits 25 bits): bits,
    all-ones and u64-valued coefficients at 2^22, slices always / adaptive, from device buffers, against the known-tau identity."""
    import numpy as np
    n = 1 << 22
    params = kzg_amd.setup(engine, TAU, n, g2_len=0)
    assert params.gs.window_info() == (17, 15)
    rng = np.random.default_rng(22)

    def blob(vals):
        a = np.zeros((n, 4), dtype="<u8")
        a[:, 0] = vals
        return a.tobytes()

    cases = {"bits": blob(rng.integers(0, 2, size=n, dtype=np.uint64)), "all_ones": blob(np.ones(n, dtype=np.uint64)),
             "u64": blob(rng.integers(0, 1 << 63, size=n, dtype=np.uint64))}
    buf = engine.alloc_scalars(n)
    G = C.g1_generator()
    try:
        for mode in (1, 0):
            engine.set_option("heavy_bins", mode)
            for name, b in cases.items():
                buf.upload(b)
                want = C.g1_mul(G, C.poly_eval_bytes(b, n, TAU))
                for _ in range(2):      # adaptive: the second call of a kind runs in slices
                    assert _msm_dev(engine, params.gs, buf, n) == want, (name, mode)
    finally:
        engine.set_option("heavy_bins", 0)
        params.gs.free()
        buf.free()


@pytest.mark.parametrize("n", [5, 300, 3000, 9000, 40000, 140000])
def test_default_window_choice(engine, n):
    """The widths the engine picks by itself (8, 10, 13, 17 by size) give the right commitment."""
    rng = random.Random(n)
    params = kzg_amd.setup(engine, TAU, n)
    c, W = params.gs.window_info()
    assert c in (8, 10, 13, 17) and W == (15 if c == 17 else -(-256 // c))
    sc = rand_scalars(rng, n)
    assert engine.msm(params.gs, sc) == C.g1_mul(C.g1_generator(), C.poly_eval(sc, TAU))
    params.gs.free()


@pytest.mark.parametrize("wb", [16, 17, 18, 19, 20])
def test_wide_windows(engine, wb):
    """Window widths chosen by option (window_bits): 16 (the default below 2^19 points tops out at 15) and the widths above it.  18..20: two-pass sort, 2^(wb-1) buckets, row/column bucket reduction; 20 by default through its own two-level sort (1024 bins of 512 buckets, 13 windows).
    17: single-pass sort that walks the scalars twice (half the 2^16 buckets per walk), 15 windows, scalars >= 2^254 replaced by -(r - k).
    Random, adversarial (one bucket, extreme digits, values around 2^254 and r) and u64-valued scalars at 2^16 terms,
    sub-ranges with an offset, and the batched entry point, each against the known-tau identity."""
    n = 1 << 16
    rng = random.Random(wb)
    engine.set_option("window_bits", wb)
    try:
        params = kzg_amd.setup(engine, TAU, n, g2_len=0)
        c, W = params.gs.window_info()
        assert c == wb and W == (15 if wb == 17 else -(-256 // wb))
        pw = [1]
        for _ in range(n - 1):
            pw.append(pw[-1] * TAU % R)
        G = C.g1_generator()

        def want(sc, off=0):
            return C.g1_mul(G, sum(s * pw[off + i] for i, s in enumerate(sc)) % R)

        cases = {
            "random": rand_scalars(rng, n),
            "u64": rand_scalars(rng, n, "u64"),
            "all_equal": [rng.randrange(R)] * n,
            "extreme_digits": [int(("8" + "0" * (wb // 4 - 1)) * (255 // wb), 16) % R] * n,
            "r_minus_1_and_zero": [(R - 1) * (i & 1) for i in range(n)],
            "around_2^254": [((1 << 254) + d) % R for d in (-2, -1, 0, 1, 2, (1 << 237), -(1 << 237))] * (n // 7) + [R - 2] * (n % 7),
            "top_window_max": [((1 << 254) - 1) - (i % 3)  for i in range(n)],
        }
        for single_pass in ((0, 1) if wb in (17, 20) else (0,)):   # 17, 20: the two-level sort (default) and the older one
            engine.set_option("sort_single_pass", single_pass)
            for name, sc in cases.items():
                assert engine.msm(params.gs, sc) == want(sc), (wb, name, single_pass)
        engine.set_option("sort_single_pass", 0)
        sub = cases["random"][:5000]
        assert engine.msm(params.gs, sub, offset=12345) == want(sub, 12345)
        assert engine.msm(params.gs, [], n=0) == bytes(96)
        batch = [rand_scalars(rng, 3000) for _ in range(5)]
        got = engine.msm_batch(params.gs, [x for b in batch for x in b], 3000, 5)
        assert got == [want(b) for b in batch]
        params.gs.free()
    finally:
        engine.set_option("sort_single_pass", 0)
        engine.set_option("window_bits", 0)


def _view(engine, buf, first, n):
    """A DeviceBuffer view of buf[first, first + n) (no ownership)."""
    v = kzg_amd.DeviceBuffer.__new__(kzg_amd.DeviceBuffer)
    v.engine, v.n, v.sfmt, v.ptr = engine, n, buf.sfmt, ctypes.c_void_p(buf.ptr.value + 32 * first)
    return v


def test_config5_2_24_sharded_8_ways_and_whole(engine):
    """configs[4]: degree-2^24 coeff-form commit with the SRS sharded contiguously over 8 ranks, 2^21 terms each.
    The test box has one GPU, so the 8 shards (kzg_shard_range + kzg_srs_setup_g1_shard: c = 17, 15 window rows, 3.5 GiB each)
    are built and reduced one after the other on it, exactly as rank r would; the 8 Jacobian partials are added as the
    group's combine step does (kzg_g1_sum_batch over the [world][batch] layout) and must equal
      * [p(tau)]G with p(tau) = sum_r tau^(r 2^21) p_r(tau)  (known-tau identity, per-shard Horner evaluations), and
      * the commitment of the same 2^24 coefficients against the whole SRS resident on ONE GPU (26 GiB)."""
    from kzg_amd.distributed import shard_range
    n, world = 1 << 24, 8
    buf = engine.alloc_scalars(n).fill_random(2024)
    parts, ptau = [], 0
    for r in range(world):
        lo, hi = shard_range(n, r, world)
        assert hi - lo == 1 << 21
        shard = kzg_amd.setup_shard(engine, TAU, lo, hi - lo)
        assert shard.window_info() == (17, 15)
        view = _view(engine, buf, lo, hi - lo)
        out = ctypes.create_string_buffer(144)
        rc = engine.lib.kzg_msm_g1(engine.ctx, shard.handle, 0, view.ptr, hi - lo, buf.sfmt, L.IN_DEVICE, out, L.G1_JACOBIAN_MONT)
        assert rc == 0, engine.last_error()
        parts.append(out.raw)
        ptau = (ptau + pow(TAU, lo, R) * oeval(view, TAU)) % R
        shard.free()
    want = C.g1_mul(C.g1_generator(), ptau)
    out = ctypes.create_string_buffer(96)
    rc = engine.lib.kzg_g1_sum_batch(engine.ctx, b"".join(parts), world, 1, L.G1_JACOBIAN_MONT, 0, out, L.G1_AFFINE_MONT)
    assert rc == 0 and out.raw == want
    # the whole SRS on one GPU
    params = kzg_amd.setup(engine, TAU, n, g2_len=0)
    assert params.gs.window_info() == (20, 13)     # from 2^23 points on: 13 windows of 20 bits
    assert _msm_dev(engine, params.gs, buf, n) == want
    params.gs.free()
    engine.set_option("window_bits", 17)           # and the 15-window layout at the same size
    try:
        params = kzg_amd.setup(engine, TAU, n, g2_len=0)
        assert params.gs.window_info() == (17, 15)
        assert _msm_dev(engine, params.gs, buf, n) == want
        params.gs.free()
    finally:
        engine.set_option("window_bits", 0)
    _fresh(buf); buf.free()


def test_compute_lagrange_basis_2_20_without_tau(engine, big):
    """f1: compute_lagrange_basis at production size from the monomial SRS alone (what a ceremony SRS allows): the inverse
    group-NTT of gs equals the known-tau closed form point for point, and commits eval-form data to the coeff-form commitment."""
    n, params, lag = big
    got = kzg_amd.compute_lagrange_basis(params)
    assert len(got) == n
    # the closed form it is compared with below, pinned by the oracle at a few indices: L_i(tau) G with
    # L_i(tau) = (tau^n - 1) w^i / (n (tau - w^i))
    _, _, omega = kzg_amd.compute_omega(n)
    for i in (0, 1, 54321, n - 1):
        wi = pow(omega, i, R)
        li = (pow(TAU, n, R) - 1) * wi % R * pow(n * (TAU - wi) % R, -1, R) % R
        assert lag.download(i, 1) == C.g1_mul(C.g1_generator(), li), i
    step = 1 << 14                                   # 64 windows of 2^14 points: 96 MiB compared piecewise
    for off in range(0, n, step):
        assert got.download(off, step) == lag.download(off, step), off
    buf = engine.alloc_scalars(n).fill_random(77)
    c1 = _msm_dev(engine, params.gs, buf, n)
    engine.ntt(buf, 20)
    assert _msm_dev(engine, got, buf, n) == c1
    _fresh(buf); buf.free()
    got.free()


def test_config4_secondary_witness_many_2_20_k256(engine, big):
    """configs[3], secondary reading: 256 independent create_witness calls on one degree-2^20 polynomial sharing one SRS,
    through the pipelined kzg_witness_coeff_many; every witness against the known-tau identity, one wrong y flagged."""
    n, params, _ = big
    k = 256
    buf = engine.alloc_scalars(n).fill_random(41)
    ptau = oeval(buf, TAU)
    xs = [kzg_amd.splitmix_scalar(4242, i) for i in range(k)]
    ys = [oeval(buf, x) for x in xs]
    ys[100] = (ys[100] + 1) % R
    prover = kzg_amd.KZGProver(params)
    ws, ok = prover.create_witness_many(None, list(zip(xs, ys)), coeffs_device=buf)
    assert ok == [j != 100 for j in range(k)]
    G = C.g1_generator()
    for j in range(k):
        yj = ys[j] if j != 100 else (ys[j] - 1) % R
        assert ws[j] == C.g1_mul(G, (ptau - yj) * pow(TAU - xs[j], -1, R) % R), j
    _fresh(buf); buf.free()
