#!/usr/bin/env python3
"""Per-kernel breakdown (HIP events on the lane's stream) of one lone kzg_witness_coeff_batched call at 2^20, k = 256, next to a lone
commit: where the ~1.1 ms over a commit go.   python tools/prof_witness.py [log_n] [k]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kzg_amd
from kzg_amd import _lib as L

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
k = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n = 1 << log_n
e = kzg_amd.Engine(0)
params = kzg_amd.setup(e, 0x5EED5EED5EED5EED, n, g2_len=0)
coeffs = e.alloc_scalars(n).fill_random(1)
xs = [kzg_amd.splitmix_scalar(7, i) for i in range(k)]
ys = [e.poly_eval(coeffs, v) for v in xs]
xb, yb = kzg_amd.pack_scalars(xs), kzg_amd.pack_scalars(ys)
out = ctypes.create_string_buffer(96)
rbuf, rlen = ctypes.create_string_buffer(32 * k), ctypes.c_size_t()


def batched():
    rc = e.lib.kzg_witness_coeff_batched(e.ctx, params.gs.handle, coeffs.ptr, n, xb, yb, k, coeffs.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT, rbuf, ctypes.byref(rlen))
    assert rc == 0, e.last_error()


def commit():
    assert e.lib.kzg_commit_coeff(e.ctx, params.gs.handle, coeffs.ptr, n, coeffs.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT) == 0


for name, f in (("commit", commit), ("witness_batched", batched)):
    for _ in range(3):
        f()
    t0 = time.perf_counter()
    for _ in range(10):
        f()
    wall = (time.perf_counter() - t0) / 10 * 1e3
    e.prof_enable(True)
    e.prof_reset()
    for _ in range(5):
        f()
    pr = e.prof_all()
    e.prof_enable(False)
    tot = sum(v[1] for v in pr.values()) / 5
    print("%s: wall %.3f ms (profiling off); kernels %.3f ms per call:" % (name, wall, tot))
    for kn, v in sorted(pr.items(), key=lambda kv: -kv[1][1]):
        print("   %-28s %5.1f launches/call  %.4f ms/call" % (kn, v[0] / 5, v[1] / 5))
