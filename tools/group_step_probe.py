#!/usr/bin/env python3
"""A device group of one GPU (RCCL all-gather forced on, process-per-GPU entry like the bench's sharded block): wall time per batch step of
kzg_commit_coeff_sharded_batch and the HIP-event kernel times of one step.  KZG_AMD_LIBRARY=<other build> for an A/B."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kzg_amd
from kzg_amd import _lib as L
from kzg_amd.api import DeviceGroup
from tools.benchlib.common import view, TAU, SEED

uid = DeviceGroup.unique_id()
group = DeviceGroup.for_rank(0, 0, 1, uid)
group.set_option("always_gather", 1)
eng = group.engine(0)
if os.environ.get("PROBE_STREAMS"):
    group.set_option("streams", int(os.environ["PROBE_STREAMS"]))
n, batch = 1 << 20, 64
scal = eng.alloc_scalars(n * batch)
for b in range(batch):
    view(kzg_amd, scal, b * n, n).fill_random(SEED + 1000 * b)
msrs = group.setup(TAU, n)
out = ctypes.create_string_buffer(96 * batch)
ptrs = (ctypes.c_void_p * 1)(scal.ptr.value)


def step():
    rc = group.lib.kzg_commit_coeff_sharded_batch(group.handle, msrs.handle, ptrs, n, batch, scal.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT)
    assert rc == 0, group.last_error()


step(); step()
ts = []
for _ in range(4):
    t0 = time.perf_counter(); step(); ts.append((time.perf_counter() - t0) * 1e3)
print("library", os.environ.get("KZG_AMD_LIBRARY", "shipped"), "ms per step", [round(t, 1) for t in ts], "=> %.1f commitments/s" % (batch / (sum(ts) / len(ts)) * 1e3))
eng.prof_enable(True); eng.prof_reset()
t0 = time.perf_counter(); step(); wall = (time.perf_counter() - t0) * 1e3
pr = eng.prof_all(); eng.prof_enable(False)
print("profiled step %.1f ms;" % wall, {k: (v[0], round(v[1], 2)) for k, v in sorted(pr.items(), key=lambda kv: -kv[1][1])[:8]})
print(group.info())
