#!/usr/bin/env python3
"""NTT timing probe: kernel times (HIP events on the engine's stream) and wall time of a device-resident 2^log_n transform for the
tile widths ntt_vec_log = 2, 1, 0.   python tools/ntt_probe.py [log_n ...]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kzg_amd
from kzg_amd import _lib as L

e = kzg_amd.Engine(0)
for log_n in [int(a) for a in sys.argv[1:]] or [20]:
    n = 1 << log_n
    buf = e.alloc_scalars(n).fill_random(3)
    for vec in (2, 1, 0):
        e.set_option("ntt_vec_log", vec)
        for inv in (0, 1):
            assert e.lib.kzg_ntt_fr(e.ctx, buf.ptr, log_n, inv, L.IN_DEVICE) == 0
        e.prof_enable(True)
        e.prof_reset()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            assert e.lib.kzg_ntt_fr(e.ctx, buf.ptr, log_n, 0, L.IN_DEVICE) == 0
        wall = (time.perf_counter() - t0) / reps * 1e3
        prof = e.prof_all()
        e.prof_enable(False)
        ks = {k: round(v[1] / reps, 4) for k, v in sorted(prof.items()) if k.startswith("k_ntt")}
        t0 = time.perf_counter()
        for _ in range(reps):
            assert e.lib.kzg_ntt_fr(e.ctx, buf.ptr, log_n, 0, L.IN_DEVICE) == 0
        wall2 = (time.perf_counter() - t0) / reps * 1e3
        print("2^%d vec_log=%d kernels %s sum %.4f ms; wall %.4f ms (profiled run %.4f)" % (log_n, vec, ks, sum(ks.values()), wall2, wall), flush=True)
    buf.free()
