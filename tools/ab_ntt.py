#!/usr/bin/env python3
"""Same-box A/B of the NTT of several builds of libkzg_mi355x.so (interleaved rounds in one process): kernel times by HIP events.
   python tools/ab_ntt.py tools/bin/lib_A.so tools/bin/lib_B.so [...] [log_n]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kzg_amd import _lib as L

paths = [a for a in sys.argv[1:] if not a.isdigit()]
log_n = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 20
n = 1 << log_n
vp, sz, i32, u32, u64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint64
libs = []
for p in paths:
    lib = ctypes.CDLL(os.path.abspath(p))
    for name, (res, args) in {
        "kzg_ctx_create": (i32, [i32, ctypes.POINTER(vp)]), "kzg_dev_alloc": (i32, [vp, sz, ctypes.POINTER(vp)]),
        "kzg_fill_random_fr": (i32, [vp, vp, sz, u64, i32, i32]), "kzg_ntt_fr": (i32, [vp, vp, u32, i32, i32]),
        "kzg_prof_enable": (i32, [vp, i32]), "kzg_prof_reset": (i32, [vp]),
        "kzg_prof_get": (i32, [vp, ctypes.c_char_p, ctypes.POINTER(u64), ctypes.POINTER(ctypes.c_double)]),
        "kzg_dev_download": (i32, [vp, vp, vp, sz]),
    }.items():
        f = getattr(lib, name); f.restype = res; f.argtypes = args
    ctx, buf = vp(), vp()
    assert lib.kzg_ctx_create(0, ctypes.byref(ctx)) == 0
    assert lib.kzg_dev_alloc(ctx, n * 32, ctypes.byref(buf)) == 0
    assert lib.kzg_fill_random_fr(ctx, buf, n, 1, 0, L.FR_CANONICAL) == 0
    libs.append((lib, ctx, buf))
outs = []
for lib, ctx, buf in libs:   # same input, one transform each: the builds must agree bit for bit
    assert lib.kzg_ntt_fr(ctx, buf, log_n, 0, L.IN_DEVICE) == 0
    h = ctypes.create_string_buffer(32 * min(n, 4096))
    lib.kzg_dev_download(ctx, h, buf, len(h))
    outs.append(h.raw)
print("outputs equal across builds:", all(o == outs[0] for o in outs))
res = [{"pass1": [], "pass2": [], "wall": []} for _ in libs]
for rnd in range(6):
    for k, (lib, ctx, buf) in enumerate(libs):
        lib.kzg_prof_enable(ctx, 1); lib.kzg_prof_reset(ctx)
        reps = 30
        for _ in range(reps):
            assert lib.kzg_ntt_fr(ctx, buf, log_n, rnd & 1, L.IN_DEVICE) == 0
        for nm, key in ((b"k_ntt_pass1", "pass1"), (b"k_ntt_pass2", "pass2")):
            l, ms = u64(), ctypes.c_double()
            lib.kzg_prof_get(ctx, nm, ctypes.byref(l), ctypes.byref(ms))
            res[k][key].append(ms.value / max(l.value, 1))
        lib.kzg_prof_enable(ctx, 0)
        t0 = time.perf_counter()
        for _ in range(reps):
            assert lib.kzg_ntt_fr(ctx, buf, log_n, rnd & 1, L.IN_DEVICE) == 0
        res[k]["wall"].append((time.perf_counter() - t0) / reps * 1e3)
for p, r in zip(paths, res):
    med = lambda v: sorted(v)[len(v) // 2]
    print("%-28s 2^%d  pass1 %.4f  pass2 %.4f  sum %.4f ms   wall %.4f ms" % (os.path.basename(p), log_n, med(r["pass1"]), med(r["pass2"]),
                                                                          med(r["pass1"]) + med(r["pass2"]), med(r["wall"])))
