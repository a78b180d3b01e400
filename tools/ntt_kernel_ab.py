#!/usr/bin/env python3
"""Round 6: the three NTT pass kernels of ONE build, same box, same process (option ntt_kernel): 0 = three-phase passes
(k_ntt_pass1 / k_ntt_pass2), 1 = load / store fused into the first / last stage pair (k_ntt_tile), 2 = 1 with two butterflies per thread.
Outputs compared bit for bit (forward, inverse, short-input transform), then kernel times by HIP events, rounds interleaved.
   python tools/ntt_kernel_ab.py [log_n ...]"""
import ctypes, os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kzg_amd
from kzg_amd import _lib as L

# variants: NTT_VARIANTS="label:opt=value,opt=value;label:..." (default: the three kernels, two-pass everywhere)
_V = os.environ.get("NTT_VARIANTS", "0:ntt_kernel=0,ntt_three_from=0;1:ntt_kernel=1,ntt_three_from=0;2:ntt_kernel=2,ntt_three_from=0")
VARIANTS = {}
for item in _V.split(";"):
    label, opts = item.split(":")
    VARIANTS[label] = [(kv.split("=")[0], int(kv.split("=")[1])) for kv in opts.split(",")]
KERNELS = list(VARIANTS)


def select(e, label):
    for k, v in VARIANTS[label]:
        e.set_option(k, v)
e = kzg_amd.Engine(0)
for log_n in [int(a) for a in sys.argv[1:]] or [20]:
    n = 1 << log_n
    src = e.alloc_scalars(n).fill_random(3)
    raw = src.download()
    buf = e.alloc_scalars(n)
    digests = {}
    for kern in KERNELS:
        select(e, kern)
        d = []
        for inv in (0, 1):
            buf.upload(raw)
            assert e.lib.kzg_ntt_fr(e.ctx, buf.ptr, log_n, inv, L.IN_DEVICE) == 0
            d.append(hashlib.sha256(bytes(buf.download())).hexdigest()[:16])
        digests[kern] = d
    ok = all(v == digests[KERNELS[0]] for v in digests.values())
    print("2^%d outputs equal across kernels %s: %s %s" % (log_n, KERNELS, ok, "" if ok else digests), flush=True)
    res = {k: {"p1": [], "p2": [], "wall": []} for k in KERNELS}
    for rnd in range(6):
        for kern in KERNELS:
            select(e, kern)
            reps = 30 if log_n <= 22 else 8
            e.prof_enable(True); e.prof_reset()
            for _ in range(reps):
                assert e.lib.kzg_ntt_fr(e.ctx, buf.ptr, log_n, rnd & 1, L.IN_DEVICE) == 0
            prof = e.prof_all()
            e.prof_enable(False)
            res[kern]["p1"].append((prof.get("k_ntt_pass1", (0, 0))[1] + prof.get("k_ntt_pass1b", (0, 0))[1]) / reps)
            res[kern]["p2"].append(prof.get("k_ntt_pass2", (0, 0))[1] / reps)
            t0 = time.perf_counter()
            for _ in range(reps):
                assert e.lib.kzg_ntt_fr(e.ctx, buf.ptr, log_n, rnd & 1, L.IN_DEVICE) == 0
            e.sync() if hasattr(e, "sync") else None
            res[kern]["wall"].append((time.perf_counter() - t0) / reps * 1e3)
    med = lambda v: sorted(v)[len(v) // 2]
    for kern in KERNELS:
        r = res[kern]
        print("  variant %-6s 2^%d  pass1(+1b) %.4f  pass2 %.4f  sum %.4f ms   wall %.4f ms" % (kern, log_n, med(r["p1"]), med(r["p2"]), med(r["p1"]) + med(r["p2"]), med(r["wall"])), flush=True)
    src.free(); buf.free()
