#!/usr/bin/env python3
"""Same-box A/B of several builds of libkzg_mi355x.so (interleaved rounds in one process):
   python tools/ab_libs.py tools/bin/lib_A.so tools/bin/lib_B.so [...] [log_n]"""
import ctypes, json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kzg_amd import _lib as L

paths = [a for a in sys.argv[1:] if not a.isdigit()]
log_n = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 20
n = 1 << log_n
TAU = (0x5EED5EED5EED5EED).to_bytes(32, "little")
vp, sz, i32, u64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint64
libs = []
for p in paths:
    lib = ctypes.CDLL(os.path.abspath(p))
    for name, (res, args) in {
        "kzg_ctx_create": (i32, [i32, ctypes.POINTER(vp)]), "kzg_srs_setup_g1": (i32, [vp, vp, i32, sz, ctypes.POINTER(vp)]),
        "kzg_dev_alloc": (i32, [vp, sz, ctypes.POINTER(vp)]), "kzg_fill_random_fr": (i32, [vp, vp, sz, u64, i32, i32]),
        "kzg_msm_g1": (i32, [vp, vp, sz, vp, sz, i32, i32, vp, i32]), "kzg_msm_g1_batch": (i32, [vp, vp, sz, vp, sz, sz, i32, i32, vp, i32]),
        "kzg_prof_enable": (i32, [vp, i32]), "kzg_prof_reset": (i32, [vp]),
        "kzg_prof_get": (i32, [vp, ctypes.c_char_p, ctypes.POINTER(u64), ctypes.POINTER(ctypes.c_double)]),
        "kzg_last_error": (ctypes.c_char_p, [vp]), "kzg_ctx_set_option": (i32, [vp, ctypes.c_char_p, ctypes.c_int64]),
    }.items():
        f = getattr(lib, name); f.restype = res; f.argtypes = args
    ctx, srs, buf = vp(), vp(), vp()
    assert lib.kzg_ctx_create(0, ctypes.byref(ctx)) == 0
    assert lib.kzg_srs_setup_g1(ctx, TAU, L.FR_CANONICAL, n, ctypes.byref(srs)) == 0, lib.kzg_last_error(ctx)
    B = 64
    assert lib.kzg_ctx_set_option(ctx, b"streams", 16) == 0
    assert lib.kzg_dev_alloc(ctx, n * 32 * B, ctypes.byref(buf)) == 0
    assert lib.kzg_fill_random_fr(ctx, buf, n * B, 1, 0, L.FR_CANONICAL) == 0
    libs.append((lib, ctx, srs, buf))
res = [{"accum_ms": [], "latency_ms": [], "batch64_commits_per_s": [], "kernels": None} for _ in libs]
outs = []
for rnd in range(5):
    for k, (lib, ctx, srs, buf) in enumerate(libs):
        out = ctypes.create_string_buffer(96 * 64)
        lib.kzg_prof_enable(ctx, 1); lib.kzg_prof_reset(ctx)
        t0 = time.perf_counter()
        assert lib.kzg_msm_g1(ctx, srs, 0, buf, n, L.FR_CANONICAL, L.IN_DEVICE, out, L.G1_AFFINE_MONT) == 0
        lat = time.perf_counter() - t0
        cnt, ms = u64(), ctypes.c_double()
        lib.kzg_prof_get(ctx, b"k_accum_affine", ctypes.byref(cnt), ctypes.byref(ms))
        kern = {}
        for kn in (b"k_hist", b"k_scan_blocks", b"k_scan_buckets", b"k_scan_a", b"k_scan_b", b"k_scatter", b"k_accum_xyzz", b"k_level_scan", b"k_bucket_reduce",
                   b"k_sum_level", b"k_fold_dense", b"k_fold_overflow", b"k_rc_sums", b"k_weighted_bits", b"k_reduce_final", b"k_emit_points"):
            c2, m2 = u64(), ctypes.c_double()
            lib.kzg_prof_get(ctx, kn, ctypes.byref(c2), ctypes.byref(m2)); kern[kn.decode()] = round(m2.value, 4)
        lib.kzg_prof_enable(ctx, 0)
        if rnd == 0:
            outs.append(out.raw[:96])
        t0 = time.perf_counter()
        for _ in range(3):
            assert lib.kzg_msm_g1_batch(ctx, srs, 0, buf, n, 64, L.FR_CANONICAL, L.IN_DEVICE, out, L.G1_AFFINE_MONT) == 0
        b8 = (time.perf_counter() - t0) / 3
        if rnd:
            res[k]["accum_ms"].append(round(ms.value / max(1, cnt.value), 4)); res[k]["latency_ms"].append(round(lat * 1e3, 3))
            res[k]["batch64_commits_per_s"].append(round(64 / b8, 1)); res[k]["kernels"] = {a: b for a, b in kern.items() if b}
assert len(set(outs)) == 1, "builds disagree on the result"
for p, r in zip(paths, res):
    print(os.path.basename(p), json.dumps(r))
