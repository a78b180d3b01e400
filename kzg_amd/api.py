"""Host-side mirror of the reference's prover surface over the C ABI (include/kzg_mi355x.h).

Names, argument meaning and error behaviour follow proxima-one/kzg:
  setup / KZGParams                      src/lib.rs:14-55
  Polynomial                             src/polynomial.rs:24-165
  EvaluationDomain                       src/ft.rs:17-140
  KZGProver / KZGVerifier.verify_poly    src/coeff_form.rs:37-124
  KZGProverEvalForm / verify_poly        src/eval_form.rs:39-171
Scalars cross this layer as python ints (canonical, < r) or as packed 32-byte little-endian blobs;
G1 points as 96-byte affine-Montgomery blobs (`bytes`), identity = 96 zero bytes.  All arithmetic
happens on the GPU in libkzg_mi355x.so; nothing here computes field or curve operations.
"""
import ctypes
import os

from . import _lib as L

R_MODULUS = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
FR_MONT_R = (1 << 256) % R_MODULUS  # blst_fr / KZG_FR_MONT_LE_32: a * 2^256 mod r


class KZGError(Exception):
    """src/lib.rs:26-36"""


class PointNotOnPolynomial(KZGError):
    pass


class PolynomialDegreeTooLarge(KZGError):
    pass


class ReferencePanic(Exception):
    """A condition on which the reference panics (slice out of range, failed assert!, ...)."""


class EngineError(RuntimeError):
    pass


def _raise(engine, rc):
    msg = engine.last_error() if engine is not None else ""
    if rc == L.KZG_ERR_POINT_NOT_ON_POLY:
        raise PointNotOnPolynomial(msg or "point not on polynomial!")
    if rc == L.KZG_ERR_DEGREE_TOO_LARGE:
        raise PolynomialDegreeTooLarge(msg or "polynomial degree too large")
    if rc == L.KZG_ERR_SHAPE:
        raise ReferencePanic(msg)
    raise EngineError(f"kzg_mi355x error {rc}: {msg}")


def pack_scalars(xs):
    """list of ints -> canonical LE blob; bytes-like passes through."""
    if isinstance(xs, (bytes, bytearray, memoryview)):
        return bytes(xs)
    return b"".join((int(x) % R_MODULUS).to_bytes(32, "little") for x in xs)


def unpack_scalars(b):
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


class DeviceBuffer:
    """Scalars resident in HBM (n x 32 B), owned by an Engine."""

    def __init__(self, engine, n, sfmt=L.FR_CANONICAL):
        self.engine, self.n, self.sfmt = engine, n, sfmt
        p = ctypes.c_void_p()
        rc = engine.lib.kzg_dev_alloc(engine.ctx, n * 32, ctypes.byref(p))
        if rc:
            _raise(engine, rc)
        self.ptr = p

    def upload(self, blob):
        assert len(blob) == self.n * 32
        rc = self.engine.lib.kzg_dev_upload(self.engine.ctx, self.ptr, blob, len(blob))
        if rc:
            _raise(self.engine, rc)
        return self

    def download(self, n=None, offset=0):
        n = self.n - offset if n is None else n
        out = ctypes.create_string_buffer(n * 32)
        src = ctypes.c_void_p(self.ptr.value + offset * 32)
        rc = self.engine.lib.kzg_dev_download(self.engine.ctx, out, src, n * 32)
        if rc:
            _raise(self.engine, rc)
        return out.raw

    def fill_random(self, seed, u64_valued=False):
        rc = self.engine.lib.kzg_fill_random_fr(self.engine.ctx, self.ptr, self.n, seed, 1 if u64_valued else 0, self.sfmt)
        if rc:
            _raise(self.engine, rc)
        return self

    def free(self):
        if self.ptr:
            self.engine.lib.kzg_dev_free(self.engine.ctx, self.ptr)
            self.ptr = None


def splitmix_scalar(seed, i, u64_valued=False):
    """The element kzg_fill_random_fr writes at index i (definition in include/kzg_mi355x.h)."""
    M = (1 << 64) - 1

    def sm(z):
        z = (z + 0x9E3779B97F4A7C15) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)

    v = 0
    for k in range(1 if u64_valued else 4):
        v |= sm((seed + 4 * i + k) & M) << (64 * k)
    return v % R_MODULUS


class Engine:
    """One kzg_ctx bound to one GPU."""

    def __init__(self, device=0):
        self.lib = L.load()
        ctx = ctypes.c_void_p()
        rc = self.lib.kzg_ctx_create(device, ctypes.byref(ctx))
        if rc:
            raise EngineError(f"kzg_ctx_create(device={device}) failed with {rc}: no usable HIP device "
                              "(kzg_amd has no CPU fallback)")
        self.ctx = ctx
        self.device = device

    def close(self):
        if self.ctx and not getattr(self, "_borrowed", False):
            self.lib.kzg_ctx_destroy(self.ctx)
        self.ctx = None

    def last_error(self):
        return (self.lib.kzg_last_error(self.ctx) or b"").decode()

    def set_option(self, key, value):
        rc = self.lib.kzg_ctx_set_option(self.ctx, key.encode(), value)
        if rc:
            _raise(self, rc)

    def info(self):
        """the batched pipeline's plan and whether the process' hardware-queue pool narrowed it (kzg_ctx_info)"""
        buf = ctypes.create_string_buffer(512)
        rc = self.lib.kzg_ctx_info(self.ctx, buf, 512)
        if rc:
            _raise(self, rc)
        return buf.value.decode()

    def sync(self):
        rc = self.lib.kzg_sync(self.ctx)
        if rc:
            _raise(self, rc)

    def alloc_scalars(self, n, sfmt=L.FR_CANONICAL):
        return DeviceBuffer(self, n, sfmt)

    # --- profiling ---
    def prof_enable(self, on=True):
        """True / 1: HIP events around every kernel; 2: around the bucket-accumulation kernel only; False / 0: off"""
        self.lib.kzg_prof_enable(self.ctx, int(on))

    def prof_reset(self):
        self.lib.kzg_prof_reset(self.ctx)

    def prof_get(self, name):
        n, ms = ctypes.c_uint64(), ctypes.c_double()
        self.lib.kzg_prof_get(self.ctx, name.encode(), ctypes.byref(n), ctypes.byref(ms))
        return n.value, ms.value

    def prof_all(self):
        buf = ctypes.create_string_buffer(8192)
        self.lib.kzg_prof_names(self.ctx, buf, 8192)
        names = [s for s in buf.value.decode().split(",") if s]
        return {k: self.prof_get(k) for k in names}

    # --- raw operations used by the classes below and by bench.py ---
    def _scalars_arg(self, scalars):
        if isinstance(scalars, DeviceBuffer):
            return scalars.ptr, scalars.n, scalars.sfmt, L.IN_DEVICE, None
        blob = pack_scalars(scalars)
        return blob, len(blob) // 32, L.FR_CANONICAL, 0, blob

    def msm(self, srs, scalars, n=None, offset=0, ofmt=L.G1_AFFINE_MONT):
        ptr, nn, sfmt, flags, _keep = self._scalars_arg(scalars)
        n = nn if n is None else n
        out = ctypes.create_string_buffer(L.POINT_BYTES[ofmt])
        rc = self.lib.kzg_msm_g1(self.ctx, srs.handle, offset, ptr, n, sfmt, flags, out, ofmt)
        if rc:
            _raise(self, rc)
        return out.raw

    def msm_batch(self, srs, scalars, n, batch, offset=0, ofmt=L.G1_AFFINE_MONT):
        ptr, nn, sfmt, flags, _keep = self._scalars_arg(scalars)
        assert nn >= n * batch
        out = ctypes.create_string_buffer(L.POINT_BYTES[ofmt] * batch)
        rc = self.lib.kzg_msm_g1_batch(self.ctx, srs.handle, offset, ptr, n, batch, sfmt, flags, out, ofmt)
        if rc:
            _raise(self, rc)
        psz = L.POINT_BYTES[ofmt]
        return [out.raw[i * psz:(i + 1) * psz] for i in range(batch)]

    def g1_sum(self, blobs, pfmt=L.G1_AFFINE_MONT, ofmt=L.G1_AFFINE_MONT):
        raw = b"".join(blobs)
        out = ctypes.create_string_buffer(L.POINT_BYTES[ofmt])
        rc = self.lib.kzg_g1_sum(self.ctx, raw, len(blobs), pfmt, 0, out, ofmt)
        if rc:
            _raise(self, rc)
        return out.raw

    def g1_sum_batch(self, blobs, count, groups, pfmt=L.G1_AFFINE_MONT, ofmt=L.G1_AFFINE_MONT):
        """out[g] = sum_i blobs[g*count + i]."""
        raw = b"".join(blobs)
        psz = L.POINT_BYTES[ofmt]
        out = ctypes.create_string_buffer(psz * groups)
        rc = self.lib.kzg_g1_sum_batch(self.ctx, raw, count, groups, pfmt, 0, out, ofmt)
        if rc:
            _raise(self, rc)
        return [out.raw[i * psz:(i + 1) * psz] for i in range(groups)]

    def ntt(self, data, log_n, inverse=False):
        """data: DeviceBuffer (in place) or ints/blob (returns list of ints)."""
        if isinstance(data, DeviceBuffer):
            rc = self.lib.kzg_ntt_fr(self.ctx, data.ptr, log_n, 1 if inverse else 0, L.IN_DEVICE)
            if rc:
                _raise(self, rc)
            return data
        blob = pack_scalars(data)
        assert len(blob) == 32 << log_n
        buf = ctypes.create_string_buffer(blob, len(blob))
        rc = self.lib.kzg_ntt_fr(self.ctx, buf, log_n, 1 if inverse else 0, 0)
        if rc:
            _raise(self, rc)
        return unpack_scalars(buf.raw)

    def coset_ntt(self, data, log_n, inverse=False):
        blob = pack_scalars(data)
        buf = ctypes.create_string_buffer(blob, len(blob))
        rc = self.lib.kzg_coset_ntt_fr(self.ctx, buf, log_n, 1 if inverse else 0, L.FR_CANONICAL, 0)
        if rc:
            _raise(self, rc)
        return unpack_scalars(buf.raw)

    def poly_eval(self, coeffs, x, n=None):
        ptr, nn, sfmt, flags, _keep = self._scalars_arg(coeffs)
        n = nn if n is None else n
        out = ctypes.create_string_buffer(32)
        xb = self._host_scalar(x, sfmt)
        rc = self.lib.kzg_poly_eval(self.ctx, ptr, n, xb, sfmt, flags, out)
        if rc:
            _raise(self, rc)
        return self._scalar_from(out.raw, sfmt)

    def quotient_linear(self, coeffs, x, y):
        blob = pack_scalars(coeffs)
        n = len(blob) // 32
        out = ctypes.create_string_buffer(max(1, n - 1) * 32)
        rc = self.lib.kzg_quotient_linear(self.ctx, blob, n, (x % R_MODULUS).to_bytes(32, "little"),
                                          (y % R_MODULUS).to_bytes(32, "little"), L.FR_CANONICAL, 0, out)
        if rc:
            _raise(self, rc)
        return unpack_scalars(out.raw[: (n - 1) * 32])

    def poly_mul(self, a, b):
        """Polynomial::fft_mul / best_mul: coefficient list of a * b."""
        ab, bb = pack_scalars(a), pack_scalars(b)
        na, nb = len(ab) // 32, len(bb) // 32
        out = ctypes.create_string_buffer(32 * (na + nb - 1))
        rc = self.lib.kzg_poly_mul(self.ctx, ab, na, bb, nb, L.FR_CANONICAL, 0, out)
        if rc:
            _raise(self, rc)
        return unpack_scalars(out.raw)

    def quotient_eval(self, evals, i):
        blob = pack_scalars(evals)
        d = len(blob) // 32
        out = ctypes.create_string_buffer(d * 32)
        rc = self.lib.kzg_quotient_eval(self.ctx, blob, d, i, L.FR_CANONICAL, 0, out)
        if rc:
            _raise(self, rc)
        return unpack_scalars(out.raw)

    _MONT_R = (1 << 256) % R_MODULUS

    def _host_scalar(self, x, sfmt):
        x %= R_MODULUS
        if sfmt == L.FR_MONT:
            x = x * self._MONT_R % R_MODULUS
        return x.to_bytes(32, "little")

    def _scalar_from(self, b, sfmt):
        v = int.from_bytes(b, "little")
        if sfmt == L.FR_MONT:
            v = v * pow(self._MONT_R, -1, R_MODULUS) % R_MODULUS
        return v


def compute_omega(d):
    """EvaluationDomain::compute_omega (src/ft.rs:55-76) -> (m, exp, omega)."""
    lib = L.load()
    m, exp, w = ctypes.c_size_t(), ctypes.c_uint32(), ctypes.create_string_buffer(32)
    rc = lib.kzg_compute_omega(d, ctypes.byref(m), ctypes.byref(exp), w, L.FR_CANONICAL)
    if rc:
        _raise(None, rc)
    return m.value, exp.value, int.from_bytes(w.raw, "little")


class Srs:
    """A resident G1 SRS (kzg_srs)."""

    def __init__(self, engine, handle):
        self.engine, self.handle = engine, handle

    def __len__(self):
        return self.engine.lib.kzg_srs_len(self.handle)

    def table_rows(self):
        return self.engine.lib.kzg_srs_table_rows(self.handle)

    def window_info(self):
        c, w = ctypes.c_int(), ctypes.c_int()
        self.engine.lib.kzg_srs_window_info(self.handle, ctypes.byref(c), ctypes.byref(w))
        return c.value, w.value

    def download(self, offset=0, n=None):
        n = len(self) - offset if n is None else n
        out = ctypes.create_string_buffer(96 * max(n, 1))
        rc = self.engine.lib.kzg_srs_download_g1(self.engine.ctx, self.handle, offset, n, out, L.G1_AFFINE_MONT)
        if rc:
            _raise(self.engine, rc)
        return out.raw[: 96 * n]

    def free(self):
        if self.handle:
            self.engine.lib.kzg_srs_free(self.engine.ctx, self.handle)
            self.handle = None

    @staticmethod
    def upload(engine, blob, n, pfmt=L.G1_AFFINE_MONT):
        h = ctypes.c_void_p()
        assert len(blob) == n * L.POINT_BYTES[pfmt]
        rc = engine.lib.kzg_srs_upload_g1(engine.ctx, blob, n, pfmt, ctypes.byref(h))
        if rc:
            _raise(engine, rc)
        return Srs(engine, h)


class SrsG2:
    """Resident G2 points (kzg_srs_g2): the `hs` half of KZGParams or a G2 Lagrange basis."""

    def __init__(self, engine, handle):
        self.engine, self.handle = engine, handle

    def __len__(self):
        return self.engine.lib.kzg_srs_g2_len(self.handle)

    def download(self, offset=0, n=None, pfmt=L.G2_AFFINE_MONT):
        n = len(self) - offset if n is None else n
        sz = L.G2_POINT_BYTES[pfmt]
        out = ctypes.create_string_buffer(sz * max(n, 1))
        rc = self.engine.lib.kzg_srs_download_g2(self.engine.ctx, self.handle, offset, n, out, pfmt)
        if rc:
            _raise(self.engine, rc)
        return out.raw[: sz * n]

    def msm(self, scalars, offset=0, ofmt=L.G2_AFFINE_MONT):
        """G2Projective::multi_exp(&hs[offset..offset+len], scalars) (call site src/coeff_form.rs:156)"""
        blob = pack_scalars(scalars)
        out = ctypes.create_string_buffer(L.G2_POINT_BYTES[ofmt])
        rc = self.engine.lib.kzg_msm_g2(self.engine.ctx, self.handle, offset, blob, len(blob) // 32, L.FR_CANONICAL, out, ofmt)
        if rc:
            _raise(self.engine, rc)
        return out.raw

    def free(self):
        if self.handle:
            self.engine.lib.kzg_srs_g2_free(self.engine.ctx, self.handle)
            self.handle = None

    @staticmethod
    def upload(engine, blob, n, pfmt=L.G2_AFFINE_MONT):
        h = ctypes.c_void_p()
        assert len(blob) == n * L.G2_POINT_BYTES[pfmt]
        rc = engine.lib.kzg_srs_upload_g2(engine.ctx, blob, n, pfmt, ctypes.byref(h))
        if rc:
            _raise(engine, rc)
        return SrsG2(engine, h)


class KZGParams:
    """src/lib.rs:14-19: gs = [s^i]G (resident G1 SRS), hs = [s^i]H (resident G2 points; None if not generated)."""

    def __init__(self, gs, hs=None):
        self.gs, self.hs = gs, hs


def setup_g2(engine, s, n):
    """the hs half of setup() (src/lib.rs:48-52): hs[i] = [s^i]H for i < n."""
    h = ctypes.c_void_p()
    rc = engine.lib.kzg_srs_setup_g2(engine.ctx, (s % R_MODULUS).to_bytes(32, "little"), L.FR_CANONICAL, n, ctypes.byref(h))
    if rc:
        _raise(engine, rc)
    return SrsG2(engine, h)


def setup_lagrange_g2(engine, s, d):
    """lagrange_basis_h for a known secret: same elements as compute_lagrange_basis(&setup(s, d)).1."""
    h = ctypes.c_void_p()
    rc = engine.lib.kzg_srs_setup_lagrange_g2(engine.ctx, (s % R_MODULUS).to_bytes(32, "little"), L.FR_CANONICAL, d,
                                              ctypes.byref(h))
    if rc:
        _raise(engine, rc)
    return SrsG2(engine, h)


def setup(engine, s, num_coeffs, g2_len=None):
    """setup(s, num_coeffs) (src/lib.rs:38-55): gs[i] = [s^i]G and hs[i] = [s^i]H, generated on the GPU.
    The reference always builds num_coeffs G2 powers; only the verifier reads them (hs[0], hs[1], and hs[..k+1] for a
    k-point batched opening), so `g2_len` caps that half (default min(num_coeffs, 257): enough for 256-point batches;
    pass g2_len=num_coeffs for the reference's full vector, 0 to skip it)."""
    h = ctypes.c_void_p()
    rc = engine.lib.kzg_srs_setup_g1(engine.ctx, (s % R_MODULUS).to_bytes(32, "little"), L.FR_CANONICAL, num_coeffs,
                                     ctypes.byref(h))
    if rc:
        _raise(engine, rc)
    if g2_len is None:
        g2_len = min(num_coeffs, 257)
    return KZGParams(Srs(engine, h), setup_g2(engine, s, g2_len) if g2_len else None)


def setup_shard(engine, s, first, n):
    """gs[first .. first+n) of setup(s, first+n): the contiguous SRS shard one rank holds."""
    h = ctypes.c_void_p()
    rc = engine.lib.kzg_srs_setup_g1_shard(engine.ctx, (s % R_MODULUS).to_bytes(32, "little"), L.FR_CANONICAL, first, n,
                                           ctypes.byref(h))
    if rc:
        _raise(engine, rc)
    return Srs(engine, h)


def setup_lagrange(engine, s, d):
    """lagrange_basis_g for a known secret: same elements as compute_lagrange_basis(&setup(s, d)).0."""
    h = ctypes.c_void_p()
    rc = engine.lib.kzg_srs_setup_lagrange_g1(engine.ctx, (s % R_MODULUS).to_bytes(32, "little"), L.FR_CANONICAL, d,
                                              ctypes.byref(h))
    if rc:
        _raise(engine, rc)
    return Srs(engine, h)


def compute_lagrange_basis(params):
    """compute_lagrange_basis (src/eval_form.rs:254-280), G1 half, from the monomial SRS."""
    e = params.gs.engine
    h = ctypes.c_void_p()
    rc = e.lib.kzg_srs_lagrange_from_monomial_g1(e.ctx, params.gs.handle, ctypes.byref(h))
    if rc:
        _raise(e, rc)
    return Srs(e, h)


def compute_lagrange_basis_g2(params):
    """compute_lagrange_basis (src/eval_form.rs:254-280), G2 half, from hs alone (hs must hold all d powers)."""
    e = params.gs.engine
    if params.hs is None:
        raise ReferencePanic("KZGParams.hs is empty")
    h = ctypes.c_void_p()
    rc = e.lib.kzg_srs_lagrange_from_monomial_g2(e.ctx, params.hs.handle, ctypes.byref(h))
    if rc:
        _raise(e, rc)
    return SrsG2(e, h)


class ShardedSrs:
    """An SRS sharded contiguously over a DeviceGroup (kzg_msrs)."""

    def __init__(self, group, handle):
        self.group, self.handle = group, handle

    def __len__(self):
        return self.group.lib.kzg_msrs_len(self.handle)

    def shard(self, local_index=0):
        """(Srs view of the resident shard of local GPU `local_index`, index of its first point); owned by this object."""
        first = ctypes.c_size_t()
        h = self.group.lib.kzg_msrs_shard(self.handle, local_index, ctypes.byref(first))
        return Srs(self.group.engine(local_index), ctypes.c_void_p(h)), first.value

    def free(self):
        if self.handle:
            self.group.lib.kzg_msrs_free(self.group.handle, self.handle)
            self.handle = None


class DeviceGroup:
    """kzg_mctx: a group of GPUs holding a sharded SRS; partial commitments are combined over RCCL inside the library.
    DeviceGroup(devices=[0, 1, ...]) -- one process drives all of them;
    DeviceGroup.for_rank(device, rank, world, unique_id) -- one process per GPU (unique_id from DeviceGroup.unique_id()
    on rank 0, distributed by the host)."""

    def __init__(self, devices=(0,), _handle=None):
        self.lib = L.load()
        if _handle is None:
            self._host_env()
            arr = (ctypes.c_int * len(devices))(*devices)
            h = ctypes.c_void_p()
            rc = self.lib.kzg_mctx_create(arr, len(devices), ctypes.byref(h))
            if rc:
                raise EngineError(f"kzg_mctx_create({list(devices)}) failed with {rc}: {self._create_error(self.lib)} "
                                  "(kzg_amd has no CPU fallback)")
            _handle = h
            self._flush_c_stdio()
        self.handle = _handle
        self._engines = {}
        self._flushed = False

    @staticmethod
    def _flush_c_stdio():
        """RCCL prints its version banner (NCCL_DEBUG=VERSION, exported on some images) with printf: into a pipe or file that is
        block-buffered and would surface at process exit, after whatever the host printed last.  Flush it where it was caused."""
        try:
            ctypes.CDLL(None).fflush(None)
        except (OSError, AttributeError):
            pass

    @staticmethod
    def _host_env():
        """This module is the host: the one-node RCCL knobs go into the environment before the first RCCL call."""
        from .distributed import single_node_rccl_env
        return single_node_rccl_env()

    @staticmethod
    def _create_error(lib):
        return (lib.kzg_mctx_create_error() or b"").decode()

    @staticmethod
    def _node_local(single_node):
        """single_node=None (default): the world is taken to live on this node unless the launcher says otherwise -- MASTER_ADDR set and
        not a loopback address.  A world that spans nodes must NOT get the loopback bootstrap knobs (NCCL_SOCKET_IFNAME=lo,
        NCCL_IB_DISABLE=1, NCCL_NET_PLUGIN=none): with them it cannot form its communicator and just meets the formation deadline
        (ADVICE r5).  Pass single_node=False (or export KZG_RCCL_SINGLE_NODE_ENV=0) there and bring your own NCCL_* environment."""
        if single_node is not None:
            return bool(single_node)
        addr = os.environ.get("MASTER_ADDR", "")
        return addr in ("", "localhost") or addr.startswith("127.") or addr == "::1"

    @staticmethod
    def unique_id(single_node=None):
        lib = L.load()
        if DeviceGroup._node_local(single_node):
            DeviceGroup._host_env()
        buf = ctypes.create_string_buffer(128)
        rc = lib.kzg_mctx_unique_id(buf)
        if rc:
            raise EngineError(f"kzg_mctx_unique_id failed with {rc}: {DeviceGroup._create_error(lib)}")
        return buf.raw

    @staticmethod
    def for_rank(device, rank, world, unique_id, single_node=None):
        """One process per GPU.  The one-node RCCL environment is applied only when the world is node-local (_node_local)."""
        lib = L.load()
        if DeviceGroup._node_local(single_node):
            DeviceGroup._host_env()
        h = ctypes.c_void_p()
        rc = lib.kzg_mctx_create_rank(device, rank, world, unique_id, ctypes.byref(h))
        if rc:
            raise EngineError(f"kzg_mctx_create_rank(device={device}, rank={rank}/{world}) failed with {rc}: "
                              f"{DeviceGroup._create_error(lib)}")
        DeviceGroup._flush_c_stdio()
        return DeviceGroup(_handle=h)

    def last_error(self):
        return (self.lib.kzg_mctx_last_error(self.handle) or b"").decode()

    def _check(self, rc):
        # the communicator is formed inside the first call that needs it: RCCL's banner is flushed ONCE, after the first call that
        # succeeded -- not a dlopen and a flush of all the host's stdio on every group call
        if not self._flushed:
            if rc == 0:
                self._flushed = True
            self._flush_c_stdio()
        if rc:
            _raise(self, rc)

    @property
    def world(self):
        return self.lib.kzg_mctx_world(self.handle)

    @property
    def local_count(self):
        return self.lib.kzg_mctx_local_count(self.handle)

    def rank(self, local_index=0):
        return self.lib.kzg_mctx_rank(self.handle, local_index)

    def engine(self, local_index=0):
        """The single-GPU Engine of a local GPU (borrowed: closed with the group)."""
        if local_index not in self._engines:
            e = Engine.__new__(Engine)
            e.lib = self.lib
            e.ctx = ctypes.c_void_p(self.lib.kzg_mctx_ctx(self.handle, local_index))
            e.device = None
            e._borrowed = True
            self._engines[local_index] = e
        return self._engines[local_index]

    def set_option(self, key, value):
        self._check(self.lib.kzg_mctx_set_option(self.handle, key.encode(), value))

    def setup(self, s, n):
        """setup(s, n).gs sharded over the group (src/lib.rs:38-47)."""
        h = ctypes.c_void_p()
        self._check(self.lib.kzg_srs_setup_g1_sharded(self.handle, (s % R_MODULUS).to_bytes(32, "little"), L.FR_CANONICAL, n,
                                                      ctypes.byref(h)))
        return ShardedSrs(self, h)

    def upload(self, blob, n, pfmt=L.G1_AFFINE_MONT):
        assert len(blob) == n * L.POINT_BYTES[pfmt]
        h = ctypes.c_void_p()
        self._check(self.lib.kzg_srs_upload_g1_sharded(self.handle, blob, n, pfmt, ctypes.byref(h)))
        return ShardedSrs(self, h)

    def commit(self, srs, coeffs, ofmt=L.G1_AFFINE_MONT):
        """KZGProver::commit over the group; coeffs: ints / canonical blob (host)."""
        blob = pack_scalars(coeffs)
        return self.commit_batch(srs, blob, len(blob) // 32, 1, ofmt=ofmt)[0]

    def commit_batch(self, srs, coeffs, n, batch, ofmt=L.G1_AFFINE_MONT, sfmt=L.FR_CANONICAL):
        """coeffs: host blob of batch * n scalars, or a list of DeviceBuffer (one per local GPU, [batch][shard] slices)."""
        psz = L.POINT_BYTES[ofmt]
        out = ctypes.create_string_buffer(psz * max(batch, 1))
        if isinstance(coeffs, (list, tuple)) and coeffs and isinstance(coeffs[0], DeviceBuffer):
            ptrs = (ctypes.c_void_p * len(coeffs))(*[c.ptr.value for c in coeffs])
            rc = self.lib.kzg_commit_coeff_sharded_batch(self.handle, srs.handle, ptrs, n, batch, coeffs[0].sfmt, L.IN_DEVICE, out, ofmt)
        else:
            blob = pack_scalars(coeffs)
            assert len(blob) == 32 * n * batch
            rc = self.lib.kzg_commit_coeff_sharded_batch(self.handle, srs.handle, blob, n, batch, sfmt, 0, out, ofmt)
        self._check(rc)
        return [out.raw[i * psz:(i + 1) * psz] for i in range(batch)]

    def _whole_poly_arg(self, coeffs):
        """host blob / ints, or a list of DeviceBuffer (one per local GPU, each the WHOLE polynomial)"""
        if isinstance(coeffs, (list, tuple)) and coeffs and isinstance(coeffs[0], DeviceBuffer):
            ptrs = (ctypes.c_void_p * len(coeffs))(*[c.ptr.value for c in coeffs])
            return ptrs, coeffs[0].n, coeffs[0].sfmt, L.IN_DEVICE
        blob = pack_scalars(coeffs)
        return blob, len(blob) // 32, L.FR_CANONICAL, 0

    def create_witness(self, srs, coeffs, point, ofmt=L.G1_AFFINE_MONT):
        """KZGProver::create_witness over the group (replicated quotient, sharded MSM)."""
        arg, n, sfmt, flags = self._whole_poly_arg(coeffs)
        x, y = point
        conv = (lambda v: (v % R_MODULUS).to_bytes(32, "little")) if sfmt == L.FR_CANONICAL else (lambda v: (v * FR_MONT_R % R_MODULUS).to_bytes(32, "little"))
        out = ctypes.create_string_buffer(L.POINT_BYTES[ofmt])
        self._check(self.lib.kzg_witness_coeff_sharded(self.handle, srs.handle, arg, n, conv(x), conv(y), sfmt, flags, out, ofmt))
        return out.raw

    def create_witness_batched(self, srs, coeffs, points, ofmt=L.G1_AFFINE_MONT):
        """KZGProver::create_witness_batched over the group: (witness bytes, interpolant coefficients)."""
        arg, n, sfmt, flags = self._whole_poly_arg(coeffs)
        k = len(points)
        conv = (lambda v: (v % R_MODULUS).to_bytes(32, "little")) if sfmt == L.FR_CANONICAL else (lambda v: (v * FR_MONT_R % R_MODULUS).to_bytes(32, "little"))
        xs = b"".join(conv(p[0]) for p in points)
        ys = b"".join(conv(p[1]) for p in points)
        out = ctypes.create_string_buffer(L.POINT_BYTES[ofmt])
        rbuf = ctypes.create_string_buffer(32 * max(k, 2))
        rlen = ctypes.c_size_t()
        self._check(self.lib.kzg_witness_coeff_batched_sharded(self.handle, srs.handle, arg, n, xs, ys, k, sfmt, flags, out, ofmt,
                                                               rbuf, ctypes.byref(rlen)))
        r = unpack_scalars(rbuf.raw[:32 * rlen.value])
        if sfmt != L.FR_CANONICAL:
            rinv = pow(FR_MONT_R, -1, R_MODULUS)
            r = [v * rinv % R_MODULUS for v in r]
        return out.raw, r

    def create_witness_eval(self, lagrange_srs, evals, index, ofmt=L.G1_AFFINE_MONT):
        """KZGProverEvalForm::create_witness over the group (src/eval_form.rs:124-140): replicated div_by_omega_i, sharded MSM
        against the Lagrange-basis SRS `lagrange_srs` (DeviceGroup.upload of the basis)."""
        arg, d, sfmt, flags = self._whole_poly_arg(evals)
        out = ctypes.create_string_buffer(L.POINT_BYTES[ofmt])
        self._check(self.lib.kzg_witness_eval_sharded(self.handle, lagrange_srs.handle, arg, d, index, sfmt, flags, out, ofmt))
        return out.raw

    def info(self):
        """which RCCL / HIP runtime the group runs on and what forming its communicator cost (kzg_mctx_info)"""
        buf = ctypes.create_string_buffer(2048)
        self._check(self.lib.kzg_mctx_info(self.handle, buf, 2048))
        return buf.value.decode()

    def formation(self):
        """{'formation_ms': f, 'load': .., 'uid': .., 'init': .., 'first_exchange': .., 'destroy': ..} parsed from info() (ms; -1 =
        has not happened)"""
        import re
        s = self.info()
        d = {"formation_ms": float(re.search(r"formation_ms=(-?[0-9.]+)", s).group(1))}
        for k, v in re.findall(r"(load|uid|init|first_exchange|destroy)=(-?[0-9.]+)", s):
            d[k] = float(v)
        return d

    def close(self):
        if self.handle:
            self.lib.kzg_mctx_destroy(self.handle)
            self.handle = None
            for e in self._engines.values():
                e.ctx = None


class Polynomial:
    """src/polynomial.rs:24-27: dense coefficients + explicit degree."""

    def __init__(self, coeffs, degree=None):
        self.coeffs = [int(c) % R_MODULUS for c in coeffs]
        if degree is None:  # Polynomial::new (:83-87) via compute_degree (:94-105)
            degree = len(self.coeffs) - 1
            while degree > 0 and self.coeffs[degree] == 0:
                degree -= 1
        self.degree = degree

    @staticmethod
    def new_from_coeffs(coeffs, degree):
        return Polynomial(coeffs, degree)

    def num_coeffs(self):  # :135-137
        return self.degree + 1

    def slice_coeffs(self):  # :148-150
        return self.coeffs[: self.num_coeffs()]

    def eval(self, engine, x):  # :156-165
        return engine.poly_eval(self.slice_coeffs(), x)

    def fft_mul(self, engine, other):  # :167-183 (and best_mul :185-191: the product is the same polynomial)
        coeffs = engine.poly_mul(self.slice_coeffs(), other.slice_coeffs())
        return Polynomial(coeffs)  # From<EvaluationDomain> = Polynomial::new (src/ft.rs:27-31)

    best_mul = fft_mul

    def __eq__(self, other):  # :29-40
        return self.degree == other.degree and all(a == b for a, b in zip(self.coeffs, other.coeffs))


class EvaluationDomain:
    """src/ft.rs:17-25"""

    def __init__(self, coeffs, d, exp, omega):
        self.coeffs, self.d, self.exp, self.omega = [int(c) % R_MODULUS for c in coeffs], d, exp, omega

    @staticmethod
    def from_coeffs(coeffs):  # :94-109
        m, exp, omega = compute_omega(len(coeffs))
        coeffs = list(coeffs) + [0] * (m - len(coeffs))
        return EvaluationDomain(coeffs, m, exp, omega)

    def __len__(self):
        return len(self.coeffs)

    def fft(self, engine):  # :111-113
        self.coeffs = engine.ntt(self.coeffs, self.exp, inverse=False)

    def ifft(self, engine):  # :115-140
        self.coeffs = engine.ntt(self.coeffs, self.exp, inverse=True)

    def coset_fft(self, engine):  # :168-171
        self.coeffs = engine.coset_ntt(self.coeffs, self.exp, inverse=False)

    def icoset_fft(self, engine):  # :173-178
        self.coeffs = engine.coset_ntt(self.coeffs, self.exp, inverse=True)

    def z(self, tau):  # :182-187
        out = ctypes.create_string_buffer(32)
        rc = L.load().kzg_domain_z(len(self.coeffs), (tau % R_MODULUS).to_bytes(32, "little"), L.FR_CANONICAL, out)
        if rc:
            _raise(None, rc)
        return int.from_bytes(out.raw, "little")

    def _vec(self, engine, fn, other=None):
        buf = ctypes.create_string_buffer(pack_scalars(self.coeffs), 32 * len(self.coeffs))
        if other is None:
            rc = fn(engine.ctx, buf, self.exp, L.FR_CANONICAL, 0)
        else:
            if len(other.coeffs) != len(self.coeffs):
                raise ReferencePanic("assert_eq!(self.coeffs.len(), other.coeffs.len())")
            rc = fn(engine.ctx, buf, pack_scalars(other.coeffs), len(self.coeffs), L.FR_CANONICAL, 0)
        if rc:
            _raise(engine, rc)
        self.coeffs = unpack_scalars(buf.raw)

    def divide_by_z_on_coset(self, engine):  # :192-217
        self._vec(engine, engine.lib.kzg_divide_by_z_on_coset)

    def mul_assign(self, engine, other):  # :220-244
        self._vec(engine, engine.lib.kzg_fr_vec_mul, other)

    def sub_assign(self, engine, other):  # :247-271
        self._vec(engine, engine.lib.kzg_fr_vec_sub, other)


class KZGBatchWitness:
    """src/coeff_form.rs:12-35"""

    def __init__(self, r, w):
        self.r, self.w = r, w

    def elem(self):
        return self.w

    def polynomial(self):
        return self.r


class KZGProver:
    """src/coeff_form.rs:37-112"""

    def __init__(self, parameters):
        self.parameters = parameters
        self.engine = parameters.gs.engine

    def commit(self, polynomial, ofmt=L.G1_AFFINE_MONT):  # :59-64
        e = self.engine
        blob = pack_scalars(polynomial.slice_coeffs())
        out = ctypes.create_string_buffer(L.POINT_BYTES[ofmt])
        rc = e.lib.kzg_commit_coeff(e.ctx, self.parameters.gs.handle, blob, polynomial.num_coeffs(), L.FR_CANONICAL, 0,
                                    out, ofmt)
        if rc:
            _raise(e, rc)
        return out.raw

    def create_witness(self, polynomial, point, ofmt=L.G1_AFFINE_MONT):  # :66-81
        e = self.engine
        x, y = point
        blob = pack_scalars(polynomial.slice_coeffs())
        out = ctypes.create_string_buffer(L.POINT_BYTES[ofmt])
        rc = e.lib.kzg_witness_coeff(e.ctx, self.parameters.gs.handle, blob, polynomial.num_coeffs(),
                                     (x % R_MODULUS).to_bytes(32, "little"), (y % R_MODULUS).to_bytes(32, "little"),
                                     L.FR_CANONICAL, 0, out, ofmt)
        if rc:
            _raise(e, rc)
        return out.raw

    def create_witness_many(self, polynomial, points, ofmt=L.G1_AFFINE_MONT, coeffs_device=None):
        """Throughput form of create_witness (not a reference method): one witness per (x, y) in `points`, all for the same
        polynomial, pipelined on the engine's lanes.  Returns (witnesses, ok) with ok[j] False where the reference would
        return Err(PointNotOnPolynomial).  coeffs_device: a DeviceBuffer already holding the coefficients."""
        e = self.engine
        k = len(points)
        n = polynomial.num_coeffs() if coeffs_device is None else coeffs_device.n
        out = ctypes.create_string_buffer(L.POINT_BYTES[ofmt] * max(k, 1))
        status = (ctypes.c_int * max(k, 1))()
        if coeffs_device is None:
            src, sfmt, flags = pack_scalars(polynomial.slice_coeffs()), L.FR_CANONICAL, 0
        else:
            src, sfmt, flags = coeffs_device.ptr, coeffs_device.sfmt, L.IN_DEVICE
        if sfmt != L.FR_CANONICAL:
            raise ValueError("create_witness_many takes canonical scalars")
        rc = e.lib.kzg_witness_coeff_many(e.ctx, self.parameters.gs.handle, src, n, pack_scalars([p[0] for p in points]),
                                          pack_scalars([p[1] for p in points]), k, sfmt, flags, out, ofmt, status)
        if rc:
            _raise(e, rc)
        psz = L.POINT_BYTES[ofmt]
        return [out.raw[j * psz:(j + 1) * psz] for j in range(k)], [status[j] == 0 for j in range(k)]

    def create_witness_batched(self, polynomial, xs, ys, ofmt=L.G1_AFFINE_MONT):  # :83-111
        e = self.engine
        assert len(xs) == len(ys)
        k = len(xs)
        blob = pack_scalars(polynomial.slice_coeffs())
        out = ctypes.create_string_buffer(L.POINT_BYTES[ofmt])
        r = ctypes.create_string_buffer(32 * max(k, 2))
        rlen = ctypes.c_size_t()
        rc = e.lib.kzg_witness_coeff_batched(e.ctx, self.parameters.gs.handle, blob, polynomial.num_coeffs(),
                                             pack_scalars(xs), pack_scalars(ys), k, L.FR_CANONICAL, 0, out, ofmt, r,
                                             ctypes.byref(rlen))
        if rc:
            _raise(e, rc)
        coeffs = unpack_scalars(r.raw[: 32 * rlen.value])
        return KZGBatchWitness(Polynomial.new_from_coeffs(coeffs, len(coeffs) - 1), out.raw)


class KZGVerifier:
    """src/coeff_form.rs:114-183; the pairing checks run on the GPU, one thread per opening."""

    def __init__(self, parameters):
        self.parameters = parameters
        self.engine = parameters.gs.engine

    def _hs(self):
        if self.parameters.hs is None:
            raise ReferencePanic("KZGParams.hs is empty (index out of bounds)")
        return self.parameters.hs

    def verify_eval(self, point, commitment, witness, pfmt=L.G1_AFFINE_MONT):  # :126-142
        return self.verify_eval_many([point], [commitment], [witness], pfmt)[0]

    def verify_eval_many(self, points, commitments, witnesses, pfmt=L.G1_AFFINE_MONT):
        """verify_eval for many independent openings in one launch -> list of bool"""
        e = self.engine
        n = len(points)
        assert len(commitments) == n and len(witnesses) == n
        ok = ctypes.create_string_buffer(max(n, 1))
        rc = e.lib.kzg_verify_eval(e.ctx, self.parameters.gs.handle, self._hs().handle,
                                   pack_scalars([p[0] for p in points]), pack_scalars([p[1] for p in points]),
                                   L.FR_CANONICAL, b"".join(commitments), b"".join(witnesses), pfmt, n, ok)
        if rc:
            _raise(e, rc)
        return [bool(b) for b in ok.raw[:n]]

    def verify_eval_batched(self, xs, commitment, witness, pfmt=L.G1_AFFINE_MONT):  # :144-182
        e = self.engine
        r = witness.r
        ok = ctypes.c_int()
        rc = e.lib.kzg_verify_eval_batched(e.ctx, self.parameters.gs.handle, self._hs().handle, pack_scalars(xs), len(xs),
                                           pack_scalars(r.slice_coeffs()), r.num_coeffs(), L.FR_CANONICAL, commitment,
                                           witness.w, pfmt, ctypes.byref(ok))
        if rc:
            _raise(e, rc)
        return bool(ok.value)

    def verify_poly(self, commitment, polynomial, pfmt=L.G1_AFFINE_MONT):
        e = self.engine
        ok = ctypes.c_int()
        rc = e.lib.kzg_verify_poly_coeff(e.ctx, self.parameters.gs.handle, commitment, pfmt,
                                         pack_scalars(polynomial.slice_coeffs()), polynomial.num_coeffs(),
                                         L.FR_CANONICAL, 0, ctypes.byref(ok))
        if rc:
            _raise(e, rc)
        return bool(ok.value)


class KZGProverEvalForm:
    """src/eval_form.rs:39-147"""

    def __init__(self, parameters, lagrange_basis_g):  # :88-100
        self.parameters = parameters
        self.lagrange_basis_g = lagrange_basis_g
        self.engine = parameters.gs.engine
        self.d, self.exp, self._omega = compute_omega(len(parameters.gs))

    def degree(self):
        return self.d

    def omega(self):
        return self._omega

    def commit(self, evals, ofmt=L.G1_AFFINE_MONT):  # :114-122
        e = self.engine
        if self.d != evals.d:
            raise ReferencePanic("assert!(self.d == evals.d) (src/eval_form.rs:115)")
        out = ctypes.create_string_buffer(L.POINT_BYTES[ofmt])
        rc = e.lib.kzg_commit_eval(e.ctx, self.lagrange_basis_g.handle, pack_scalars(evals.coeffs), len(evals),
                                   L.FR_CANONICAL, 0, out, ofmt)
        if rc:
            _raise(e, rc)
        return out.raw

    def create_witness(self, evals, i, ofmt=L.G1_AFFINE_MONT):  # :124-140
        e = self.engine
        out = ctypes.create_string_buffer(L.POINT_BYTES[ofmt])
        rc = e.lib.kzg_witness_eval(e.ctx, self.lagrange_basis_g.handle, pack_scalars(evals.coeffs), len(evals), i,
                                    L.FR_CANONICAL, 0, out, ofmt)
        if rc:
            _raise(e, rc)
        return out.raw

    def create_witness_many(self, evals, indices, ofmt=L.G1_AFFINE_MONT):
        """Throughput form of create_witness (not a reference method): one witness per index, same evaluation vector."""
        e = self.engine
        k = len(indices)
        out = ctypes.create_string_buffer(L.POINT_BYTES[ofmt] * max(k, 1))
        idx = (ctypes.c_size_t * max(k, 1))(*indices)
        rc = e.lib.kzg_witness_eval_many(e.ctx, self.lagrange_basis_g.handle, pack_scalars(evals.coeffs), len(evals), idx, k,
                                         L.FR_CANONICAL, 0, out, ofmt)
        if rc:
            _raise(e, rc)
        psz = L.POINT_BYTES[ofmt]
        return [out.raw[j * psz:(j + 1) * psz] for j in range(k)]

    def create_witness_all(self):  # :142-146: identity
        return bytes(96)


class KZGVerifierEvalForm:
    """src/eval_form.rs:149-218"""

    def __init__(self, parameters, lagrange_basis_g, lagrange_basis_h=None):
        self.parameters = parameters
        self.lagrange_basis_g = lagrange_basis_g
        self.lagrange_basis_h = lagrange_basis_h
        self.engine = parameters.gs.engine
        self.d, self.exp, self.omega = compute_omega(len(parameters.gs))

    def verify_eval(self, point, commitment, witness, pfmt=L.G1_AFFINE_MONT):  # :173-190
        i, y = point
        return KZGVerifier(self.parameters).verify_eval((pow(self.omega, i, R_MODULUS), y), commitment, witness, pfmt)

    def verify_eval_all(self, ys, commitment, witness, pfmt=L.G1_AFFINE_MONT):  # :192-217
        e = self.engine
        if self.lagrange_basis_h is None or self.parameters.hs is None:
            raise ReferencePanic("lagrange_basis_h / hs missing (index out of bounds)")
        ok = ctypes.c_int()
        rc = e.lib.kzg_verify_eval_all(e.ctx, self.lagrange_basis_g.handle, self.lagrange_basis_h.handle,
                                       self.parameters.hs.handle, pack_scalars(ys), len(ys), L.FR_CANONICAL, commitment,
                                       witness, pfmt, ctypes.byref(ok))
        if rc:
            _raise(e, rc)
        return bool(ok.value)

    def verify_poly(self, commitment, evals, pfmt=L.G1_AFFINE_MONT):
        e = self.engine
        ok = ctypes.c_int()
        rc = e.lib.kzg_verify_poly_eval(e.ctx, self.parameters.gs.handle, commitment, pfmt, pack_scalars(evals.coeffs),
                                        len(evals), L.FR_CANONICAL, 0, ctypes.byref(ok))
        if rc:
            _raise(e, rc)
        return bool(ok.value)
