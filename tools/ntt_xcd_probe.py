#!/usr/bin/env python3
"""NTT kernel times (HIP events) for the XCD-aware tile orders: option ntt_xcd = 0 (plain), 1 (pass 1; pass 2 at 2^24), 2 (pass 2), 3 (both); rounds
interleaved in one process.   python tools/ntt_xcd_probe.py [log_n ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kzg_amd
from kzg_amd import _lib as L
e = kzg_amd.Engine(0)
for log_n in [int(a) for a in sys.argv[1:]] or [20]:
    n = 1 << log_n
    buf = e.alloc_scalars(n).fill_random(3)
    MODES = (0, 1, 2, 3)
    acc = {m: [] for m in MODES}
    for rnd in range(5):
        for m in MODES:
            e.set_option("ntt_xcd", m)
            for inv in (0, 1):
                assert e.lib.kzg_ntt_fr(e.ctx, buf.ptr, log_n, inv, L.IN_DEVICE) == 0
            e.prof_enable(True); e.prof_reset()
            reps = 20 if log_n <= 22 else 6
            for _ in range(reps):
                assert e.lib.kzg_ntt_fr(e.ctx, buf.ptr, log_n, 0, L.IN_DEVICE) == 0
            pr = e.prof_all(); e.prof_enable(False)
            acc[m].append((pr["k_ntt_pass1"][1] / reps, pr["k_ntt_pass2"][1] / reps))
    med = lambda v: sorted(v)[len(v) // 2]
    print("2^%d " % log_n + "  ".join("xcd=%d: p1 %.4f p2 %.4f" % (m, med([a for a, _ in acc[m]]), med([b for _, b in acc[m]])) for m in MODES), flush=True)
    buf.free()
