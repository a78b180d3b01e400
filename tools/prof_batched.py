#!/usr/bin/env python3
"""Per-kernel breakdown of one kzg_witness_coeff_batched call at 2^log_n, k opening points (HIP events on the call's lane).
   python tools/prof_batched.py [log_n] [k]"""
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
buf = e.alloc_scalars(n).fill_random(3)
xs = [kzg_amd.splitmix_scalar(7, i) for i in range(k)]
ys = [e.poly_eval(buf, v) for v in xs]
xb, yb = kzg_amd.pack_scalars(xs), kzg_amd.pack_scalars(ys)
out, rbuf, rlen = ctypes.create_string_buffer(96), ctypes.create_string_buffer(32 * k), ctypes.c_size_t()


def call():
    rc = e.lib.kzg_witness_coeff_batched(e.ctx, params.gs.handle, buf.ptr, n, xb, yb, k, buf.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT, rbuf, ctypes.byref(rlen))
    assert rc == 0, e.last_error()


for _ in range(3):
    call()
t0 = time.perf_counter()
reps = 10
for _ in range(reps):
    call()
wall = (time.perf_counter() - t0) / reps * 1e3
e.prof_enable(True)
e.prof_reset()
for _ in range(reps):
    call()
prof = e.prof_all()
e.prof_enable(False)
tot = 0.0
for name, (launches, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
    print("  %-24s %3.1f launches/call  %8.4f ms/call" % (name, launches / reps, ms / reps))
    tot += ms / reps
msm = sum(ms for name, (l, ms) in prof.items() if name.startswith(("k_accum", "k_bin", "k_scan", "k_fold", "k_rc", "k_weighted", "k_reduce", "k_hist", "k_scatter"))) / reps
print("2^%d, k = %d: wall %.3f ms; kernels %.3f ms (MSM kernels %.3f, everything else %.3f)" % (log_n, k, wall, tot, msm, tot - msm))
