#!/usr/bin/env python3
"""Per-kernel HIP-event breakdown of create_witness_batched (k = 256) at degree 2^log_n: python tools/prof_batched.py [log_n]"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kzg_amd
from kzg_amd import _lib as L
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log_n
TAU = 0x5EED5EED5EED5EED
e = kzg_amd.Engine(0)
params = kzg_amd.setup(e, TAU, n, g2_len=0)
coeffs = e.alloc_scalars(n).fill_random(11)
k = 256
xs = [kzg_amd.splitmix_scalar(7, i) for i in range(k)]
ys = [e.poly_eval(coeffs, v) for v in xs]
xb, yb = kzg_amd.pack_scalars(xs), kzg_amd.pack_scalars(ys)
out = ctypes.create_string_buffer(96)
rbuf, rlen = ctypes.create_string_buffer(32 * k), ctypes.c_size_t()
def run():
    rc = e.lib.kzg_witness_coeff_batched(e.ctx, params.gs.handle, coeffs.ptr, n, xb, yb, k, coeffs.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT, rbuf, ctypes.byref(rlen))
    assert rc == 0, e.last_error()
run()
import time
t0 = time.perf_counter(); run(); run(); t = (time.perf_counter() - t0) / 2
e.prof_enable(True); e.prof_reset(); run()
prof = e.prof_all()
print(json.dumps({"ms": round(t * 1e3, 3), "kernels": {kk: [v[0], round(v[1], 4)] for kk, v in sorted(prof.items(), key=lambda x: -x[1][1])}}))
