#!/usr/bin/env python3
"""The device group's batched commit at world 1 (RCCL all-gather forced on) against the plain kzg_msm_g1_batch on the same polynomials, in a
process of its own: the ratio of the two rates for each lane count given (default: the engine's default).  A guard for the process'
hardware-queue budget (profiles/r06_group_exchange_stream.txt): 0.997 at 13 lanes, 0.634 at 14, where the group's exchange stream is the
25th stream of a 24-queue process.  Run it ALONE on the box (not from the test session: a pytest process that has used the GPU -- its
contexts released or not -- skews the two rates differently, which is why this is a tool and not a test).
   python tools/group_guard_check.py [lanes ...]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kzg_amd
from kzg_amd import _lib as L

TAU = 0x5EED
for streams in [int(a) for a in sys.argv[1:]] or [0]:
    n, batch = 1 << 18, 28
    group = kzg_amd.DeviceGroup([0])
    group.set_option("always_gather", 1)
    if streams:
        group.set_option("streams", streams)
    eng = group.engine(0)
    sc = eng.alloc_scalars(n * batch).fill_random(4242)
    msrs = group.setup(TAU, n)
    out, out2 = ctypes.create_string_buffer(96 * batch), ctypes.create_string_buffer(96 * batch)
    ptrs = (ctypes.c_void_p * 1)(sc.ptr.value)

    def group_step():
        assert group.lib.kzg_commit_coeff_sharded_batch(group.handle, msrs.handle, ptrs, n, batch, sc.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT) == 0, group.last_error()
    shard, _ = msrs.shard(0)

    def plain_step():
        assert eng.lib.kzg_msm_g1_batch(eng.ctx, shard.handle, 0, sc.ptr, n, batch, sc.sfmt, L.IN_DEVICE, out2, L.G1_AFFINE_MONT) == 0, eng.last_error()

    def rate(f):
        for _ in range(3):
            f()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(4):
                f()
            best = min(best, (time.perf_counter() - t0) / 4)
        return batch / best
    rp, rg = rate(plain_step), rate(group_step)
    print("GUARD streams=%s plain %.0f group %.0f ratio %.3f same_results %s" % (streams or "default", rp, rg, rg / rp, out.raw == out2.raw), flush=True)
    sc.free()
    msrs.free()
    group.close()
