#!/usr/bin/env python3
"""Concurrent blocking callers with coefficients of changing shape (uniform, u64, bits, all-ones, all-equal) on one context: every
result against the known-tau identity by the oracle.  The calls are a mix of everything that leases a lane: commit (device- and
host-resident), create_witness, create_witness_batched (k = 64), fft + ifft round trip against the oracle's serial_fft, verify_poly.  Exercises the adaptive slicing of oversized sort bins (option heavy_bins = 0)
while lanes are leased by many threads.  python tools/stress_callers.py [seconds] [log_n] [threads]"""
import ctypes, os, random, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kzg_amd
from kzg_amd import _lib as L
from oracle import c_oracle as C

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
log_n = int(sys.argv[2]) if len(sys.argv) > 2 else 18
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 16
n = 1 << log_n
TAU = 0xABCDEF12345
e = kzg_amd.Engine(0)
params = kzg_amd.setup(e, TAU, n, g2_len=0)
rng = np.random.default_rng(99)


def blob_u64(vals):
    a = np.zeros((n, 4), dtype="<u8")
    a[:, 0] = vals
    return a.tobytes()


full = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
full[:, 3] &= (1 << 60) - 1
kinds = {
    "uniform": full.astype("<u8").tobytes(),
    "u64": blob_u64(rng.integers(0, 1 << 63, size=n, dtype=np.uint64)),
    "bits": blob_u64(rng.integers(0, 2, size=n, dtype=np.uint64)),
    "all_ones": blob_u64(np.ones(n, dtype=np.uint64)),
    "all_equal": np.tile(full[:1], (n, 1)).astype("<u8").tobytes(),
}
G = C.g1_generator()
R = kzg_amd.api.R_MODULUS
ptau = {k: C.poly_eval_bytes(b, n, TAU) for k, b in kinds.items()}
want = {k: C.g1_mul(G, ptau[k]) for k in kinds}
b32 = lambda v: (v % R).to_bytes(32, "little")  # noqa: E731
KB = 64
extra = {}
for k, b in kinds.items():
    x = kzg_amd.splitmix_scalar(5, len(extra))
    y = C.poly_eval_bytes(b, n, x)
    xs = [kzg_amd.splitmix_scalar(6, 1000 * len(extra) + i) for i in range(KB)]
    ys = [C.poly_eval_bytes(b, n, v) for v in xs]
    extra[k] = dict(x=b32(x), y=b32(y), wit=C.g1_mul(G, (ptau[k] - y) * pow(TAU - x, -1, R) % R), xs=xs, ys=ys,
                    xb=kzg_amd.pack_scalars(xs), yb=kzg_amd.pack_scalars(ys), fft=C.fft_bytes(b, log_n))
dev = {}
for k, b in kinds.items():
    dev[k] = e.alloc_scalars(n)
    dev[k].upload(b)
host = {k: ctypes.create_string_buffer(b, len(b)) for k, b in kinds.items()}
scratch = [e.alloc_scalars(n) for _ in range(threads)]
t_end = time.time() + budget
fails, calls = [], [0] * threads


def work(t):
    r = random.Random(t)
    out = ctypes.create_string_buffer(96)
    names = list(kinds)
    while time.time() < t_end:
        # phases: mostly one kind for a while (the adaptive window flips), sometimes a random one
        k = names[int(time.time() / 2) % len(names)] if r.random() < 0.7 else r.choice(names)
        ex = extra[k]
        u = r.random()
        good = True
        if u < 0.3:
            rc = e.lib.kzg_commit_coeff(e.ctx, params.gs.handle, dev[k].ptr, n, dev[k].sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
            good = out.raw == want[k]
        elif u < 0.5:
            rc = e.lib.kzg_commit_coeff(e.ctx, params.gs.handle, host[k], n, L.FR_CANONICAL, 0, out, L.G1_AFFINE_MONT)
            good = out.raw == want[k]
        elif u < 0.65:
            rc = e.lib.kzg_witness_coeff(e.ctx, params.gs.handle, dev[k].ptr, n, ex["x"], ex["y"], dev[k].sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
            good = out.raw == ex["wit"]
        elif u < 0.8:
            rbuf, rlen = ctypes.create_string_buffer(32 * KB), ctypes.c_size_t()
            rc = e.lib.kzg_witness_coeff_batched(e.ctx, params.gs.handle, dev[k].ptr, n, ex["xb"], ex["yb"], KB, dev[k].sfmt, L.IN_DEVICE, out,
                                                 L.G1_AFFINE_MONT, rbuf, ctypes.byref(rlen))
            if rc == 0:
                I = kzg_amd.unpack_scalars(rbuf.raw)
                Z = 1
                for v in ex["xs"]:
                    Z = Z * (TAU - v) % R
                good = rlen.value == KB and out.raw == C.g1_mul(G, (ptau[k] - C.poly_eval(I, TAU)) * pow(Z, -1, R) % R)
        elif u < 0.9:
            w = scratch[t]
            w.upload(kinds[k])
            rc = e.lib.kzg_ntt_fr(e.ctx, w.ptr, log_n, 0, L.IN_DEVICE)
            good = w.download() == ex["fft"]
            rc = rc or e.lib.kzg_ntt_fr(e.ctx, w.ptr, log_n, 1, L.IN_DEVICE)
            good = good and w.download() == kinds[k]
        else:
            okf = ctypes.c_int(0)
            rc = e.lib.kzg_verify_poly_coeff(e.ctx, params.gs.handle, want[k], L.G1_AFFINE_MONT, dev[k].ptr, n, dev[k].sfmt, L.IN_DEVICE, ctypes.byref(okf))
            good = okf.value == 1
        calls[t] += 1
        if rc != 0 or not good:
            fails.append((t, k, rc, round(u, 2)))


th = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
for x in th:
    x.start()
for x in th:
    x.join()
print("stress_callers: %d threads, %.0f s, 2^%d: %d calls, failures: %d %s" % (threads, budget, log_n, sum(calls), len(fails), fails[:5]), flush=True)
sys.exit(1 if fails else 0)
