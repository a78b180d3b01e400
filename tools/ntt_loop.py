#!/usr/bin/env python3
"""Runs forward+inverse NTTs of one size in a loop on device-resident data (for rocprofv3 passes)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kzg_amd
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
e = kzg_amd.Engine(0)
if len(sys.argv) > 3:
    e.set_option('ntt_vec_log', int(sys.argv[3]))
buf = e.alloc_scalars(1 << log_n).fill_random(5)
orig = buf.download()
e.ntt(buf, log_n); e.ntt(buf, log_n, inverse=True)
t0 = time.perf_counter()
for _ in range(reps):
    e.ntt(buf, log_n); e.ntt(buf, log_n, inverse=True)
dt = (time.perf_counter() - t0) / (2 * reps)
assert buf.download() == orig
print(f"log_n={log_n} ntt {dt*1e3:.4f} ms  {64*(1<<log_n)/dt/1e9:.1f} GB/s algorithmic ({64*(1<<log_n)/dt/8e12*100:.2f}% of 8 TB/s)")
