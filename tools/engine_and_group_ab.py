#!/usr/bin/env python3
"""A plain Engine and a DeviceGroup alive in ONE process (what INTEGRATION.md section 5b implies for a host that keeps a KZGProver context
and a device group): batched commit rates of each alone, of the group with the engine alive but idle, and of both committing at once --
and what kzg_ctx_info / kzg_mctx_info say about the pipelines' plans.  Each scenario in a fresh child process.
   python tools/engine_and_group_ab.py [log_n] [batch]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import ctypes, json, sys, threading, time
sys.path.insert(0, %r)
import kzg_amd
from kzg_amd import _lib as L
scenario, log_n, batch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
n = 1 << log_n
TAU = 0x5EED5EED
out = {"scenario": scenario}
eng = grp = None
if scenario in ("engine", "engine_then_group", "both"):
    eng = kzg_amd.Engine(0)
    import os
    if os.environ.get("KZG_STREAMS"): eng.set_option("streams", int(os.environ["KZG_STREAMS"]))
    ep = kzg_amd.setup(eng, TAU, n, g2_len=0)
    esc = eng.alloc_scalars(n * batch).fill_random(1)
    eout = ctypes.create_string_buffer(96 * batch)
    def estep():
        assert eng.lib.kzg_msm_g1_batch(eng.ctx, ep.gs.handle, 0, esc.ptr, n, batch, esc.sfmt, L.IN_DEVICE, eout, L.G1_AFFINE_MONT) == 0, eng.last_error()
    estep(); estep()
if scenario in ("group", "engine_then_group", "both"):
    grp = kzg_amd.DeviceGroup([0]); grp.set_option("always_gather", 1)
    ms = grp.setup(TAU, n)
    ge = grp.engine(0)
    gsc = ge.alloc_scalars(n * batch).fill_random(1)
    gout = ctypes.create_string_buffer(96 * batch)
    ptrs = (ctypes.c_void_p * 1)(gsc.ptr.value)
    def gstep():
        rc = grp.lib.kzg_commit_coeff_sharded_batch(grp.handle, ms.handle, ptrs, n, batch, gsc.sfmt, L.IN_DEVICE, gout, L.G1_AFFINE_MONT)
        assert rc == 0, grp.last_error()
    gstep(); gstep()
def rate(step, reps=4):
    t0 = time.perf_counter()
    for _ in range(reps): step()
    return batch * reps / (time.perf_counter() - t0)
if scenario == "engine": out["engine_per_s"] = round(rate(estep), 1)
elif scenario in ("group", "engine_then_group"): out["group_per_s"] = round(rate(gstep), 1)
else:
    res = {}
    th = [threading.Thread(target=lambda: res.__setitem__("e", rate(estep))), threading.Thread(target=lambda: res.__setitem__("g", rate(gstep)))]
    for t in th: t.start()
    for t in th: t.join()
    out["engine_per_s"], out["group_per_s"] = round(res["e"], 1), round(res["g"], 1)
    out["sum_per_s"] = round(res["e"] + res["g"], 1)
if eng: out["engine_info"] = eng.info()
if grp: out["group_info"] = grp.info()
if grp and eng: out["same_results"] = gout.raw == eout.raw
print(json.dumps(out), flush=True)
''' % ROOT


def run(scenario, log_n, batch):
    r = subprocess.run([sys.executable, "-c", CHILD, scenario, str(log_n), str(batch)], capture_output=True, text=True, timeout=300)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    res = json.loads(line[-1]) if line else {"scenario": scenario, "rc": r.returncode}
    res["stderr_kzg_lines"] = [ln for ln in r.stderr.splitlines() if ln.startswith("kzg:")][:4]
    if not line:
        res["stderr_tail"] = r.stderr[-1500:]
    return res


if __name__ == "__main__":
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    for sc in ("engine", "group", "engine_then_group", "both"):
        print(json.dumps(run(sc, log_n, batch)), flush=True)
