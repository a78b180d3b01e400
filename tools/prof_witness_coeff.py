#!/usr/bin/env python3
"""A lone kzg_witness_coeff (KZGProver::create_witness, the call benches/create_witness_coeff_form.rs:28-31 times) beside a lone
kzg_commit_coeff on one box: wall time of the blocking call and the per-kernel breakdown (HIP events on the lane's stream), and the
same for kzg_witness_coeff_batched with k = 256.  `KZG_AMD_LIBRARY=<other build> python tools/prof_witness_coeff.py` profiles another
build of the library (A/B).   python tools/prof_witness_coeff.py [log_n]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kzg_amd
from kzg_amd import _lib as L

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log_n
R = kzg_amd.api.R_MODULUS
e = kzg_amd.Engine(0)
params = kzg_amd.setup(e, 0x5EED5EED5EED5EED, n, g2_len=0)
coeffs = e.alloc_scalars(n).fill_random(1)
x = kzg_amd.splitmix_scalar(99, 0)
y = e.poly_eval(coeffs, x)
xb, yb = (x % R).to_bytes(32, "little"), (y % R).to_bytes(32, "little")
k = 256
xs = [kzg_amd.splitmix_scalar(7, i) for i in range(k)]
ys = [e.poly_eval(coeffs, v) for v in xs]
xsb, ysb = kzg_amd.pack_scalars(xs), kzg_amd.pack_scalars(ys)
out = ctypes.create_string_buffer(96)
rbuf, rlen = ctypes.create_string_buffer(32 * k), ctypes.c_size_t()


def commit():
    assert e.lib.kzg_commit_coeff(e.ctx, params.gs.handle, coeffs.ptr, n, coeffs.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT) == 0


def witness():
    rc = e.lib.kzg_witness_coeff(e.ctx, params.gs.handle, coeffs.ptr, n, xb, yb, coeffs.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
    assert rc == 0, e.last_error()


def batched():
    rc = e.lib.kzg_witness_coeff_batched(e.ctx, params.gs.handle, coeffs.ptr, n, xsb, ysb, k, coeffs.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT, rbuf,
                                         ctypes.byref(rlen))
    assert rc == 0, e.last_error()


print("library:", os.environ.get("KZG_AMD_LIBRARY", "kzg_amd/libkzg_mi355x.so"), " 2^%d" % log_n)
walls = {}
for name, f in (("commit_coeff", commit), ("witness_coeff", witness), ("witness_coeff_batched k=256", batched)):
    for _ in range(3):
        f()
    ws = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(10):
            f()
        ws.append((time.perf_counter() - t0) / 10 * 1e3)
    wall = sorted(ws)[len(ws) // 2]
    walls[name] = wall
    e.prof_enable(True)
    e.prof_reset()
    for _ in range(5):
        f()
    pr = e.prof_all()
    e.prof_enable(False)
    tot = sum(v[1] for v in pr.values()) / 5
    print("%s: wall %.3f ms (profiling off, median of 5 x 10 calls); kernels %.3f ms per call; wall - kernels %.3f ms" % (name, wall, tot, wall - tot))
    for kn, v in sorted(pr.items(), key=lambda kv: -kv[1][1]):
        if v[1] / 5 >= 0.004:
            print("   %-28s %5.1f launches/call  %.4f ms/call" % (kn, v[0] / 5, v[1] / 5))
print("witness_coeff - commit_coeff = %.3f ms;  witness_coeff_batched - commit_coeff = %.3f ms" %
      (walls["witness_coeff"] - walls["commit_coeff"], walls["witness_coeff_batched k=256"] - walls["commit_coeff"]))
