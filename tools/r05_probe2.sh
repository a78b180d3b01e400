mkdir -p gpurun_out/r05
python tools/rccl_formation_ab.py 2 > gpurun_out/r05/rccl_formation_ab.txt 2>gpurun_out/r05/rccl_formation_ab.err
# formation deadline: hooks build, injected 5 s stall, 2 s deadline
KZG_DEBUG=1 KZG_AMD_LIBRARY=$PWD/kzg_amd/libkzg_mi355x_hooks.so KZG_TEST_FORMATION_STALL_MS=5000 python - > gpurun_out/r05/deadline.txt 2>&1 <<'P'
import time, kzg_amd
g = kzg_amd.DeviceGroup([0]); g.set_option("always_gather", 1); g.set_option("comm_timeout_ms", 2000)
s = g.setup(5, 1024)
t = time.time()
try:
    g.commit(s, list(range(1024)))
    print("NO ERROR?")
except Exception as e:
    print("error after %.2f s: %s" % (time.time() - t, e))
try:
    g.commit(s, list(range(1024)))
except Exception as e:
    print("second call:", e)
print(g.info())
t = time.time(); s.free(); g.close(); print("close %.2f s" % (time.time() - t))
try:
    g2 = kzg_amd.DeviceGroup([0]); g2.set_option("always_gather", 1); s2 = g2.setup(5, 1024); g2.commit(s2, list(range(1024)))
except Exception as e:
    print("new group in the wedged process:", e)
e = kzg_amd.Engine(0); p = kzg_amd.setup(e, 5, 1024, g2_len=0); print("plain engine still works:", len(kzg_amd.KZGProver(p).commit(kzg_amd.Polynomial(list(range(1024))))))
P
(time python -c "import __graft_entry__ as g; g.smoke()") > gpurun_out/r05/smoke_time2.txt 2>&1
cat gpurun_out/r05/rccl_formation_ab.txt gpurun_out/r05/deadline.txt; tail -n 5 gpurun_out/r05/smoke_time2.txt
