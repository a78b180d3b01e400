#!/usr/bin/env python3
"""Commit throughput / latency sweep over polynomial sizes (north-star: degrees 2^16..2^24), inputs resident in
HBM, both scalar distributions; every timed result is checked against the known-tau identity [p(tau)]G."""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kzg_amd  # noqa: E402
from kzg_amd import _lib as L  # noqa: E402
from oracle import c_oracle as C  # noqa: E402  (checker only)

TAU = 0x5EED5EED5EED5EED
R = kzg_amd.api.R_MODULUS


def main():
    logs = [int(a) for a in sys.argv[1:]] or [16, 17, 18, 19, 20, 21, 22, 23, 24]
    e = kzg_amd.Engine(0)
    rows = []
    for log_n in logs:
        n = 1 << log_n
        # the pipeline's depth where memory allows: 64 commitments per call up to 2^20, 2 GiB of resident scalars above
        batch = int(os.environ.get("KZG_SWEEP_BATCH", "0")) or (64 if log_n <= 20 else max(2, 64 >> (log_n - 20)))
        t0 = time.perf_counter()
        params = kzg_amd.setup(e, TAU, n, g2_len=0)
        t_setup = time.perf_counter() - t0
        c, W = params.gs.window_info()
        for u64 in (False, True):
            scal = e.alloc_scalars(n * batch).fill_random(3, u64_valued=u64)
            out = ctypes.create_string_buffer(96 * batch)

            def step():
                rc = e.lib.kzg_msm_g1_batch(e.ctx, params.gs.handle, 0, scal.ptr, n, batch, scal.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
                assert rc == 0, e.last_error()
            step()
            reps = 8 if log_n <= 18 else (4 if log_n <= 21 else 2)
            t0 = time.perf_counter()
            for _ in range(reps):
                step()
            dt = (time.perf_counter() - t0) / reps
            one = ctypes.create_string_buffer(96)
            t0 = time.perf_counter()
            rc = e.lib.kzg_msm_g1(e.ctx, params.gs.handle, 0, scal.ptr, n, scal.sfmt, L.IN_DEVICE, one, L.G1_AFFINE_MONT)
            lat = time.perf_counter() - t0
            assert rc == 0
            first = kzg_amd.DeviceBuffer.__new__(kzg_amd.DeviceBuffer)
            first.engine, first.n, first.sfmt, first.ptr = e, n, scal.sfmt, scal.ptr
            ok = one.raw == out.raw[:96] == C.g1_mul(C.g1_generator(), C.poly_eval_bytes(first.download(), n, TAU))  # oracle Horner
            rows.append({"log_n": log_n, "scalars": "u64" if u64 else "full", "window_bits": c, "windows": W, "batch": batch,
                         "commitments_per_s": round(batch / dt, 2), "terms_per_s": round(batch * n / dt, 0),
                         "latency_ms": round(lat * 1e3, 3), "hbm_frac": round(128.0 * n * batch / dt / 8e12, 5),
                         "setup_s": round(t_setup, 3), "matches_known_tau": bool(ok)})
            print(json.dumps(rows[-1]), flush=True)
            scal.free()
        params.gs.free()
    e.close()


if __name__ == "__main__":
    main()
