#!/usr/bin/env python3
"""u64-valued coefficients (the reference benches' own distribution, benches/commit_coeff_form.rs:16-21): batched commitments/s and the
per-kernel breakdown of one batch step (HIP events), with optional engine options KEY=VALUE on the command line.
   python tools/u64_probe.py [key=value ...]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kzg_amd
from kzg_amd import _lib as L
from tools.benchlib.common import view, timeit, TAU, SEED

e = kzg_amd.Engine(0)
U64 = os.environ.get("PROBE_FULL_WIDTH", "") == ""      # PROBE_FULL_WIDTH=1: uniform full-width scalars instead (the headline's distribution)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    e.set_option(k, int(v))
n, batch = 1 << 20, 64
params = kzg_amd.setup(e, TAU, n, g2_len=0)
sc = e.alloc_scalars(n * batch)
for b in range(batch):
    view(kzg_amd, sc, b * n, n).fill_random(SEED + 31000 + 1000 * b, u64_valued=U64)
out = ctypes.create_string_buffer(96 * batch)


def step():
    rc = e.lib.kzg_msm_g1_batch(e.ctx, params.gs.handle, 0, sc.ptr, n, batch, sc.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
    assert rc == 0, e.last_error()


ms = timeit(step, reps=5, warm=3)
print("options %s: %s batch of %d: %.3f ms per step = %.1f commitments/s" % (sys.argv[1:], "u64" if U64 else "full-width", batch, ms, batch / ms * 1e3))
e.prof_enable(True); e.prof_reset()
step()
pr = e.prof_all(); e.prof_enable(False)
tot = sum(v[1] for v in pr.values())
print("kernels of one profiled step: %.3f ms summed over streams (%.4f per MSM)" % (tot, tot / batch))
for k, v in sorted(pr.items(), key=lambda kv: -kv[1][1]):
    print("   %-24s %6d launches  %8.4f ms total  %.4f ms/launch  %.4f ms/MSM" % (k, v[0], v[1], v[1] / v[0], v[1] / batch))
one = ctypes.create_string_buffer(96)
def single():
    assert e.lib.kzg_msm_g1(e.ctx, params.gs.handle, 0, sc.ptr, n, sc.sfmt, L.IN_DEVICE, one, L.G1_AFFINE_MONT) == 0
print("lone commit: %.3f ms" % timeit(single))
