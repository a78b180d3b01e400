#!/usr/bin/env python3
"""A/B of k_accum_affine variants in one process (interleaved rounds): packed 96-B vs 128-B-aligned table rows,
2 vs 3 waves/SIMD.  All variants must return the same commitment."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kzg_amd
from kzg_amd import _lib as L

TAU = 0x5EED5EED5EED5EED
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log_n
configs = [("packed96_occ2", 0, 2), ("packed96_occ3", 0, 3), ("pad128_occ2", 1, 2), ("pad128_occ3", 1, 3)]
engines = {}
for name, pad, occ in configs:
    e = kzg_amd.Engine(0)
    e.set_option("pad_rows", pad)
    e.set_option("accum_occupancy", occ)
    params = kzg_amd.setup(e, TAU, n)
    scal = e.alloc_scalars(n * 4).fill_random(1)
    engines[name] = (e, params, scal)
res = {k: {"accum_ms": [], "latency_ms": [], "batch4_ms": []} for k in engines}
outs = {}
for rnd in range(4):
    for name, (e, params, scal) in engines.items():
        out = ctypes.create_string_buffer(96 * 4)
        e.prof_enable(True); e.prof_reset()
        t0 = time.perf_counter()
        rc = e.lib.kzg_msm_g1(e.ctx, params.gs.handle, 0, scal.ptr, n, scal.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
        lat = time.perf_counter() - t0
        assert rc == 0, e.last_error()
        cnt, ms = e.prof_get("k_accum_affine")
        e.prof_enable(False)
        outs[name] = out.raw[:96]
        t0 = time.perf_counter()
        rc = e.lib.kzg_msm_g1_batch(e.ctx, params.gs.handle, 0, scal.ptr, n, 4, scal.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
        b4 = time.perf_counter() - t0
        assert rc == 0
        if rnd:
            res[name]["accum_ms"].append(round(ms / cnt, 4)); res[name]["latency_ms"].append(round(lat * 1e3, 3))
            res[name]["batch4_ms"].append(round(b4 * 1e3, 3))
assert len(set(outs.values())) == 1, "variants disagree"
for k, v in res.items():
    print(k, json.dumps(v))
