#!/usr/bin/env python3
"""Lone 2^log_n MSM times for skewed scalar distributions (small values, bits, all-equal, sparse), each checked against the
known-tau identity.  python tools/skew_probe.py [log_n] [window_bits]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kzg_amd
from kzg_amd import _lib as L
from oracle import c_oracle as C, kzg_model as M

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
wb = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = 1 << log_n
R = M.R
TAU = 0x1234567
e = kzg_amd.Engine(0)
e.set_option("window_bits", wb)
params = kzg_amd.setup(e, TAU, n, g2_len=0)
print("window", params.gs.window_info())
rng = np.random.default_rng(5)

def blob_u64(vals):  # u64 numpy array -> canonical 32-byte little-endian scalars
    a = np.zeros((len(vals), 4), dtype="<u8")
    a[:, 0] = vals
    return a.tobytes()

full = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
full[:, 3] &= (1 << 60) - 1
cases = {
    "uniform (252-bit)": full.astype("<u8").tobytes(),
    "u64": blob_u64(rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * 2 + 1),
    "u32": blob_u64(rng.integers(0, 1 << 32, size=n, dtype=np.uint64)),
    "u16": blob_u64(rng.integers(0, 1 << 16, size=n, dtype=np.uint64)),
    "u8": blob_u64(rng.integers(0, 256, size=n, dtype=np.uint64)),
    "bits": blob_u64(rng.integers(0, 2, size=n, dtype=np.uint64)),
    "all ones": blob_u64(np.ones(n, dtype=np.uint64)),
    "sparse 1/16 uniform": (full * (rng.integers(0, 16, size=(n, 1), dtype=np.uint64) == 0)).astype("<u8").tobytes(),
    "all equal (252-bit)": np.tile(full[:1], (n, 1)).astype("<u8").tobytes(),
}
buf = e.alloc_scalars(n)
for name, blob in cases.items():
    if os.environ.get("SKEW_ONLY") and not name.startswith(os.environ["SKEW_ONLY"]):
        continue
    buf.upload(blob)
    got = e.msm(params.gs, buf, n)
    want = C.g1_mul(C.g1_generator(), C.poly_eval_bytes(blob, n, TAU))
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); e.msm(params.gs, buf, n); ts.append(time.perf_counter() - t0)
    print(f"{name:24s} {min(ts) * 1e3:8.3f} ms   matches oracle: {got == want}", flush=True)
    if os.environ.get("SKEW_PROF"):
        e.prof_reset(); e.prof_enable(True)
        e.msm(params.gs, buf, n)
        e.prof_enable(False)
        rows = sorted(e.prof_all().items(), key=lambda kv: -kv[1][1])[:int(os.environ.get('SKEW_ROWS', '6'))]
        print("      " + "  ".join(f"{k} {v[1] / max(v[0], 1) * 1e3:.0f}us" for k, v in rows), flush=True)
