"""Times each phase of forming an RCCL communicator at world 1 on this box, directly against librccl (ctypes; no torch, not
the product library): dlopen, ncclGetUniqueId, ncclCommInitRank, first and second ncclAllGather, ncclCommDestroy.
Usage: python tools/rccl_formation_probe.py [path-to-librccl]   (environment knobs are the caller's: this is the A/B tool)"""
import ctypes
import json
import os
import sys
import time

t00 = time.time()
out = {"env": {k: v for k, v in os.environ.items() if k.startswith(("NCCL_", "RCCL_", "HSA_", "GPU_", "HIP_"))}}
path = sys.argv[1] if len(sys.argv) > 1 else "/opt/rocm/lib/librccl.so.1"
t = time.time()
hip = ctypes.CDLL("libamdhip64.so", mode=ctypes.RTLD_GLOBAL)
out["dlopen_hip_s"] = round(time.time() - t, 3)
t = time.time()
assert hip.hipSetDevice(0) == 0
p = ctypes.c_void_p()
assert hip.hipMalloc(ctypes.byref(p), 1 << 20) == 0
out["hip_init_s"] = round(time.time() - t, 3)
t = time.time()
rccl = ctypes.CDLL(path)
out["dlopen_rccl_s"] = round(time.time() - t, 3)


class Uid(ctypes.Structure):
    _fields_ = [("b", ctypes.c_char * 128)]


uid = Uid()
t = time.time()
rc = rccl.ncclGetUniqueId(ctypes.byref(uid))
out["get_unique_id_s"] = round(time.time() - t, 3)
assert rc == 0, rc
comm = ctypes.c_void_p()
t = time.time()
rccl.ncclCommInitRank.argtypes = [ctypes.c_void_p, ctypes.c_int, Uid, ctypes.c_int]
rc = rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0)
out["comm_init_rank_s"] = round(time.time() - t, 3)
assert rc == 0, rc
st = ctypes.c_void_p()
assert hip.hipStreamCreate(ctypes.byref(st)) == 0
rccl.ncclAllGather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
for i in range(3):
    t = time.time()
    rc = rccl.ncclAllGather(p, ctypes.c_void_p(p.value + 4096), 1024, 1, comm, st)  # ncclUint8 = 1
    assert rc == 0, rc
    assert hip.hipStreamSynchronize(st) == 0
    out[f"all_gather_{i}_s"] = round(time.time() - t, 4)
t = time.time()
rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
rccl.ncclCommDestroy(comm)
out["comm_destroy_s"] = round(time.time() - t, 3)
out["total_s"] = round(time.time() - t00, 3)
print(json.dumps(out))
