"""
ORACLE (test infrastructure) -- ctypes binding of oracle/libkzg_oracle.so (the C restatement).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Formats: scalars are python ints (canonical); G1 points are the 96-byte affine-Montgomery blobs
(`bytes`), identity = 96 zero bytes.  Helpers convert to/from the python model's (x, y) tuples.
"""
import ctypes
import os
import subprocess

from . import kzg_model as M

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libkzg_oracle.so")


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "kzg_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


NATIVE_BUILD = None   # the command line build_native() used (bench.py quotes it)


def build_native():
    """bench.py's cpu_baseline leg only: the same source compiled for the host it runs on (-O3 -march=native, mulx/adx),
    into a temp directory (the committed recipe builds the portable -O2 library, which must run on any box).  ROCm's clang
    is preferred when present (its code for the unrolled Montgomery multiplication is ~25 % faster than gcc 11's: 40 ns
    against 46 ns per multiplication on the build box), gcc otherwise.  Returns the path, or None when no compiler works."""
    global NATIVE_BUILD
    import tempfile
    out = os.path.join(tempfile.gettempdir(), "libkzg_oracle_native_%d.so" % os.getuid())
    src = os.path.join(_HERE, "kzg_oracle.c")
    compilers = [c for c in ("/opt/rocm/lib/llvm/bin/clang", "gcc") if c == "gcc" or os.path.exists(c)]
    for cc in compilers:
        try:
            stamp = out + ".cc"
            if (not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src) or not os.path.exists(stamp)
                    or open(stamp).read() != cc):
                subprocess.check_call([cc, "-O3", "-march=native", "-fPIC", "-shared", "-Wno-unused-function", "-o", out + ".tmp", src],
                                      stderr=subprocess.DEVNULL)
                os.replace(out + ".tmp", out)
                with open(stamp, "w") as f:
                    f.write(cc)
            NATIVE_BUILD = "%s -O3 -march=native" % os.path.basename(cc)
            return out
        except Exception:
            continue
    return None


def use_library(path):
    """Load `path` instead of the portable build (bench.py's cpu_baseline workers)."""
    global _lib
    _lib = None
    _lib = _bind(ctypes.CDLL(path))


def _bind(l):
    l.orc_compute_omega.restype = ctypes.c_int
    l.orc_poly_long_division.restype = ctypes.c_int
    l.orc_witness_quotient.restype = ctypes.c_int
    l.orc_g1_on_curve.restype = ctypes.c_int
    return l


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _bind(ctypes.CDLL(build()))
    return _lib


def scalars_to_bytes(xs):
    return b"".join(M.fr_to_le(x) for x in xs)


def bytes_to_scalars(b):
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


def point_to_blob(P):
    return M.g1_to_affine_mont(P)


def blob_to_point(b):
    if b == bytes(96):
        return None
    rinv = pow(M.FQ_MONT_R, -1, M.Q)
    return (int.from_bytes(b[:48], "little") * rinv % M.Q, int.from_bytes(b[48:], "little") * rinv % M.Q)


def _buf(n):
    return ctypes.create_string_buffer(n)


def msm_g1(points_blob, scalars, naive=False):
    n = len(scalars)
    assert len(points_blob) == 96 * n
    out = _buf(96)
    f = lib().orc_msm_g1_naive if naive else lib().orc_msm_g1
    f(points_blob, scalars_to_bytes(scalars), ctypes.c_size_t(n), out)
    return out.raw


def msm_g1_raw(points_blob, scalar_bytes, n):
    out = _buf(96)
    lib().orc_msm_g1(points_blob, scalar_bytes, ctypes.c_size_t(n), out)
    return out.raw


def msm_g1_fast_raw(points_blob, scalar_bytes, n):
    """bench.py's cpu_baseline TIMING leg: signed 16-bit windows, batch-affine bucket accumulation, unrolled Montgomery
    multiplication (orc_msm_g1_fast).  Not a checker: tests compare it with msm_g1_raw."""
    out = _buf(96)
    lib().orc_msm_g1_fast(points_blob, scalar_bytes, ctypes.c_size_t(n), out)
    return out.raw


def g1_generator():
    out = _buf(96)
    lib().orc_g1_generator(out)
    return out.raw


def g1_mul(blob, k):
    out = _buf(96)
    lib().orc_g1_mul(blob, M.fr_to_le(k), out)
    return out.raw


def g1_add(a, b):
    out = _buf(96)
    lib().orc_g1_add(a, b, out)
    return out.raw


def g1_on_curve(a):
    return bool(lib().orc_g1_on_curve(a))


def g1_to_uncompressed(a):
    out = _buf(96)
    lib().orc_g1_to_uncompressed(a, out)
    return out.raw


def g1_from_uncompressed(b):
    out = _buf(96)
    lib().orc_g1_from_uncompressed(b, out)
    return out.raw


def setup_g1(tau, n):
    """src/lib.rs:38-47 (G1 half): returns n*96 bytes."""
    out = _buf(96 * n)
    lib().orc_setup_g1(M.fr_to_le(tau), ctypes.c_size_t(n), out)
    return out.raw


def compute_omega(d):
    m, exp, w = ctypes.c_uint64(), ctypes.c_uint32(), _buf(32)
    rc = lib().orc_compute_omega(ctypes.c_uint64(d), ctypes.byref(m), ctypes.byref(exp), w)
    if rc:
        raise M.PolynomialDegreeTooLarge()
    return m.value, exp.value, int.from_bytes(w.raw, "little")


def fft_bytes(data, log_n, inverse=False):
    buf = ctypes.create_string_buffer(data, len(data))
    lib().orc_fft(buf, ctypes.c_uint32(log_n), ctypes.c_int(1 if inverse else 0))
    return buf.raw


def fft(xs, inverse=False):
    log_n = (len(xs) - 1).bit_length()
    assert len(xs) == 1 << log_n
    return bytes_to_scalars(fft_bytes(scalars_to_bytes(xs), log_n, inverse))


def poly_eval(coeffs, x):
    out = _buf(32)
    lib().orc_poly_eval(scalars_to_bytes(coeffs), ctypes.c_size_t(len(coeffs)), M.fr_to_le(x), out)
    return int.from_bytes(out.raw, "little")


def poly_eval_bytes(coeff_bytes, n, x):
    out = _buf(32)
    lib().orc_poly_eval(coeff_bytes, ctypes.c_size_t(n), M.fr_to_le(x), out)
    return int.from_bytes(out.raw, "little")


def long_division(num, den):
    """Polynomial::long_division for exact-degree inputs; returns (quot, rem or None)."""
    n, m = len(num), len(den)
    assert n >= m >= 1 and den[-1] % M.R != 0
    q, r = _buf(32 * (n - m + 1)), _buf(32 * max(1, m - 1))
    nz = lib().orc_poly_long_division(scalars_to_bytes(num), ctypes.c_size_t(n), scalars_to_bytes(den),
                                      ctypes.c_size_t(m), q, r)
    return bytes_to_scalars(q.raw), (bytes_to_scalars(r.raw)[: m - 1] if nz else None)


def witness_quotient_bytes(coeff_bytes, n, x, y):
    q = _buf(32 * (n - 1))
    nz = lib().orc_witness_quotient(coeff_bytes, ctypes.c_size_t(n), M.fr_to_le(x), M.fr_to_le(y), q)
    return q.raw, bool(nz)


def div_by_omega_i_bytes(eval_bytes, d, m):
    out = _buf(32 * d)
    lib().orc_div_by_omega_i(eval_bytes, ctypes.c_size_t(d), ctypes.c_size_t(m), out)
    return out.raw
