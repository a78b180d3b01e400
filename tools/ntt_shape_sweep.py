#!/usr/bin/env python3
"""Round 6: tile widths and tile order of the fused NTT kernels (ntt_kernel 1 / 2), one process, kernel times by HIP events.
   python tools/ntt_shape_sweep.py [log_n ...]"""
import os, sys, time, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kzg_amd
from kzg_amd import _lib as L
e = kzg_amd.Engine(0)
for log_n in [int(a) for a in sys.argv[1:]] or [20]:
    n = 1 << log_n
    buf = e.alloc_scalars(n).fill_random(3)
    rows = []
    for kern, v1, v2, xcd in itertools.product((1, 2), (2, 1), (2, 1, 0), (1, 3, 0)):
        for k, v in (("ntt_kernel", kern), ("ntt_vec_log", v1), ("ntt_vec2_log", v2), ("ntt_xcd", xcd)):
            e.set_option(k, v)
        reps = 20 if log_n <= 22 else 6
        best = None
        for rnd in range(3):
            e.prof_enable(True); e.prof_reset()
            for _ in range(reps):
                assert e.lib.kzg_ntt_fr(e.ctx, buf.ptr, log_n, 0, L.IN_DEVICE) == 0
            prof = e.prof_all(); e.prof_enable(False)
            p1, p2 = prof.get("k_ntt_pass1", (0, 0))[1] / reps, prof.get("k_ntt_pass2", (0, 0))[1] / reps
            if best is None or p1 + p2 < best[0] + best[1]:
                best = (p1, p2)
        rows.append((best[0] + best[1], kern, v1, v2, xcd, best))
    rows.sort()
    for tot, kern, v1, v2, xcd, b in rows[:8] + rows[-2:]:
        print("2^%d kernel=%d vec_log=%d vec2_log=%d xcd=%d  pass1 %.4f pass2 %.4f sum %.4f ms" % (log_n, kern, v1, v2, xcd, b[0], b[1], tot), flush=True)
    buf.free()
