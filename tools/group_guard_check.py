import ctypes, sys, time, os
sys.path.insert(0, os.getcwd())
import kzg_amd
from kzg_amd import _lib as L
TAU=0x5EED
for streams in (13, 14):
    n, batch = 1 << 18, 28
    group = kzg_amd.DeviceGroup([0]); group.set_option("always_gather", 1); group.set_option("streams", streams)
    eng = group.engine(0)
    sc = eng.alloc_scalars(n * batch).fill_random(4242)
    msrs = group.setup(TAU, n)
    out = ctypes.create_string_buffer(96 * batch); ptrs = (ctypes.c_void_p * 1)(sc.ptr.value)
    def group_step():
        assert group.lib.kzg_commit_coeff_sharded_batch(group.handle, msrs.handle, ptrs, n, batch, sc.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT) == 0
    shard, _ = msrs.shard(0)
    def plain_step():
        assert eng.lib.kzg_msm_g1_batch(eng.ctx, shard.handle, 0, sc.ptr, n, batch, sc.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT) == 0
    def rate(f):
        for _ in range(3): f()
        best=1e9
        for _ in range(3):
            t0=time.perf_counter()
            for _ in range(4): f()
            best=min(best,(time.perf_counter()-t0)/4)
        return batch/best
    rp, rg = rate(plain_step), rate(group_step)
    print("streams", streams, "plain %.0f group %.0f ratio %.3f" % (rp, rg, rg/rp))
    sc.free(); msrs.free(); group.close()
