#!/usr/bin/env python3
"""Times every hot-path entry point at BASELINE sizes with inputs resident in HBM, and checks each result
with a size-independent property (known-tau identities evaluated on the GPU + one oracle scalar-mul)."""
import ctypes
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")  # before HIP initialises: one hardware queue per engine stream
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kzg_amd  # noqa: E402
from kzg_amd import _lib as L  # noqa: E402

R = kzg_amd.api.R_MODULUS
TAU = 0x5EED5EED5EED5EED


def timeit(f, reps=3):
    f()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    check = "--check" in sys.argv
    n = 1 << log_n
    e = kzg_amd.Engine(0)
    lib, ctx = e.lib, e.ctx
    t0 = time.perf_counter(); params = kzg_amd.setup(e, TAU, n, g2_len=0); t_setup = time.perf_counter() - t0
    t0 = time.perf_counter(); lag = kzg_amd.setup_lagrange(e, TAU, n); t_lag = time.perf_counter() - t0
    srs = params.gs
    res = {"log_n": log_n, "setup_s": round(t_setup, 3), "setup_lagrange_s": round(t_lag, 3), "window": srs.window_info()}
    coeffs = e.alloc_scalars(n).fill_random(11)
    out = ctypes.create_string_buffer(96)

    def b32(v):
        return (v % R).to_bytes(32, "little")

    # commit
    def commit():
        assert lib.kzg_commit_coeff(ctx, srs.handle, coeffs.ptr, n, coeffs.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT) == 0, e.last_error()
    res["commit_coeff_ms"] = round(timeit(commit), 3)
    commitment = out.raw
    ptau = e.poly_eval(coeffs, TAU)
    if check:
        from oracle import c_oracle as C
        assert commitment == C.g1_mul(C.g1_generator(), ptau), "commit != [p(tau)]G"
    # the same commit with HOST-resident coefficients: one 32*n-byte PCIe copy per call (never bench.py's `value`)
    host_coeffs = coeffs.download()
    def commit_host():
        assert lib.kzg_commit_coeff(ctx, srs.handle, host_coeffs, n, coeffs.sfmt, 0, out, L.G1_AFFINE_MONT) == 0, e.last_error()
    res["commit_coeff_host_resident_ms"] = round(timeit(commit_host), 3)
    assert out.raw == commitment
    # witness (coeff form)
    x = kzg_amd.splitmix_scalar(99, 0)
    y = e.poly_eval(coeffs, x)
    def witness():
        rc = lib.kzg_witness_coeff(ctx, srs.handle, coeffs.ptr, n, b32(x), b32(y), coeffs.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
        assert rc == 0, e.last_error()
    res["witness_coeff_ms"] = round(timeit(witness), 3)
    if check:
        assert out.raw == C.g1_mul(C.g1_generator(), (ptau - y) * pow(TAU - x, -1, R) % R), "witness mismatch"
        rc = lib.kzg_witness_coeff(ctx, srs.handle, coeffs.ptr, n, b32(x), b32(y + 1), coeffs.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
        assert rc == L.KZG_ERR_POINT_NOT_ON_POLY
    # 256 independent single-point openings of the same polynomial, pipelined (secondary reading)
    K = 256
    e.set_option("streams", 16)
    xs = [kzg_amd.splitmix_scalar(1234, j) for j in range(K)]
    ys = [e.poly_eval(coeffs, xx) for xx in xs]
    xb, yb = b"".join(b32(v) for v in xs), b"".join(b32(v) for v in ys)
    outs = ctypes.create_string_buffer(96 * K)
    st = (ctypes.c_int * K)()
    def witness_many():
        rc = lib.kzg_witness_coeff_many(ctx, srs.handle, coeffs.ptr, n, xb, yb, K, coeffs.sfmt, L.IN_DEVICE, outs, L.G1_AFFINE_MONT, st)
        assert rc == 0, e.last_error()
    t_many = timeit(witness_many, reps=2)
    res["witness_coeff_many_k256_ms"] = round(t_many, 2)
    res["witnesses_per_s_k256"] = round(K / t_many * 1e3, 1)
    if check:
        assert all(v == 0 for v in st)
        for j in (0, 1, K - 1):
            assert outs.raw[96 * j: 96 * j + 96] == C.g1_mul(C.g1_generator(), (ptau - ys[j]) * pow(TAU - xs[j], -1, R) % R), "witness_many mismatch"
    # NTT
    ev = e.alloc_scalars(n)
    def ntt():
        assert lib.kzg_ntt_fr(ctx, ev.ptr, log_n, 0, L.IN_DEVICE) == 0, e.last_error()
    ev.upload(coeffs.download())
    res["ntt_ms"] = round(timeit(ntt), 3)
    ev.upload(coeffs.download()); ntt()
    # eval-form commit == coeff-form commit (config 3)
    def commit_eval():
        assert lib.kzg_commit_eval(ctx, lag.handle, ev.ptr, n, ev.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT) == 0, e.last_error()
    res["commit_eval_ms"] = round(timeit(commit_eval), 3)
    assert out.raw == commitment, "eval-form commit != coeff-form commit"
    # eval-form witness at index m == coeff-form witness at omega^m
    m = 12345 % n
    def witness_eval():
        assert lib.kzg_witness_eval(ctx, lag.handle, ev.ptr, n, m, ev.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT) == 0, e.last_error()
    res["witness_eval_ms"] = round(timeit(witness_eval), 3)
    w_eval = out.raw
    omega = kzg_amd.compute_omega(n)[2]
    xm = pow(omega, m, R)
    ym = e.poly_eval(coeffs, xm)
    rc = lib.kzg_witness_coeff(ctx, srs.handle, coeffs.ptr, n, b32(xm), b32(ym), coeffs.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
    assert rc == 0 and out.raw == w_eval, "eval-form witness != coeff-form witness at omega^m"
    # batched witness, k = 256
    k = 256 if n > 512 else 4
    xs = [kzg_amd.splitmix_scalar(7, i) for i in range(k)]
    ys = [e.poly_eval(coeffs, v) for v in xs]
    xb, yb = kzg_amd.pack_scalars(xs), kzg_amd.pack_scalars(ys)
    rbuf = ctypes.create_string_buffer(32 * k)
    rlen = ctypes.c_size_t()
    def batched():
        rc = lib.kzg_witness_coeff_batched(ctx, srs.handle, coeffs.ptr, n, xb, yb, k, coeffs.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT,
                                           rbuf, ctypes.byref(rlen))
        assert rc == 0, e.last_error()
    res["witness_batched_k%d_ms" % k] = round(timeit(batched, reps=2), 3)
    I = kzg_amd.unpack_scalars(rbuf.raw)
    Itau = e.poly_eval(I, TAU)
    assert all(e.poly_eval(I, xs[i]) == ys[i] for i in (0, 1, k - 1)), "interpolant does not pass through the points"
    if check:
        Z = 1
        for v in xs:
            Z = Z * (TAU - v) % R
        assert out.raw == C.g1_mul(C.g1_generator(), (ptau - Itau) * pow(Z, -1, R) % R), "batched witness mismatch"
        yb2 = kzg_amd.pack_scalars(ys[:-1] + [(ys[-1] + 1) % R])
        rc = lib.kzg_witness_coeff_batched(ctx, srs.handle, coeffs.ptr, n, xb, yb2, k, coeffs.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT,
                                           rbuf, ctypes.byref(rlen))
        assert rc == L.KZG_ERR_POINT_NOT_ON_POLY
    # per-kernel breakdown of one commit + one NTT + one witness
    e.prof_enable(True); e.prof_reset()
    commit(); ntt(); witness(); batched()
    res["kernel_ms"] = {kname: round(v[1], 4) for kname, v in sorted(e.prof_all().items())}
    e.prof_enable(False)
    res["checked"] = check
    print(json.dumps(res))
    e.close()


if __name__ == "__main__":
    main()
