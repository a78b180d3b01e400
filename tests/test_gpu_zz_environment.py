"""The tests that judge the ENVIRONMENT and the library's behaviour when it misbehaves -- last in the suite (conftest.py ORDER), all
in child processes:

  * RCCL forms a world-1 communicator on this box within 30 s (on the driver's box it took five minutes); when it does
    not, the failure message carries the library's per-phase timings and RCCL's own log;
  * a formation slower than the deadline is ABANDONED: error with the phases, dead group, quick teardown, no second formation in the
    process, the plain single-GPU path unaffected (hooks build: the formation is stalled artificially);
  * a dead group whose stream an aborted collective still holds is destroyed without synchronising on it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOOKS = os.path.join(ROOT, "kzg_amd", "libkzg_mi355x_hooks.so")


@pytest.fixture(autouse=True)
def _children_get_the_gpu(released_gpu):
    pass


def _child(code, env=None, timeout=120):
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\n" % ROOT + code], env=dict(os.environ, **(env or {})),
                       capture_output=True, text=True, timeout=timeout)
    return r


def test_formation_deadline_abandons_names_the_phase_and_spares_the_process():
    code = r"""
import os, time, json
import kzg_amd
g = kzg_amd.DeviceGroup([0]); g.set_option("always_gather", 1); g.set_option("comm_timeout_ms", 1500)
s = g.setup(5, 1024)
out = {}
t = time.time()
try:
    g.commit(s, list(range(1024))); out["first"] = "no error"
except Exception as e:
    out["first"] = str(e)
out["first_s"] = time.time() - t
t = time.time()
try:
    g.commit(s, list(range(1024))); out["second"] = "no error"
except Exception as e:
    out["second"] = str(e)
out["second_s"] = time.time() - t
out["info"] = g.info()
t = time.time(); s.free(); g.close(); out["close_s"] = time.time() - t
t = time.time()
try:
    g2 = kzg_amd.DeviceGroup([0]); g2.set_option("always_gather", 1); s2 = g2.setup(5, 1024); g2.commit(s2, list(range(1024))); out["new_group"] = "no error"
except Exception as e:
    out["new_group"] = str(e)
out["new_group_s"] = time.time() - t
try:
    kzg_amd.DeviceGroup.unique_id(); out["uid"] = "no error"
except Exception as e:
    out["uid"] = str(e)
e = kzg_amd.Engine(0); p = kzg_amd.setup(e, 5, 1024, g2_len=0)
out["plain_commit"] = kzg_amd.KZGProver(p).commit(kzg_amd.Polynomial(list(range(1024)))).hex()
print(json.dumps(out), flush=True)
os._exit(0)      # (a helper thread may still sit inside the abandoned ncclCommInitAll: no static destructors under its feet)
"""
    r = _child(code, {"KZG_AMD_LIBRARY": HOOKS, "KZG_TEST_FORMATION_STALL_MS": "6000"})
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    import re
    assert "did not return within 1500 ms" in out["first"] and "load=" in out["first"], out["first"]
    assert 1500 <= float(re.search(r"init=([0-9.]+)", out["first"]).group(1)) < 5000, out["first"]     # the phase that consumed the time, named
    assert "NCCL_SOCKET_IFNAME=lo" in out["first"]
    assert out["first_s"] < 12
    assert "dead" in out["second"] and out["second_s"] < 0.5
    assert "dead=1" in out["info"] and "comm_timeout_ms=1500" in out["info"]
    assert out["close_s"] < 3
    assert "never returned" in out["new_group"] and out["new_group_s"] < 3
    assert "never returned" in out["uid"]
    from oracle import c_oracle as C
    want = C.g1_mul(C.g1_generator(), C.poly_eval(list(range(1024)), 5))
    assert out["plain_commit"] == want.hex()        # the single-GPU prover is untouched by the wedged RCCL


def test_dead_group_with_a_stuck_stream_is_destroyed_without_a_hang(need_rccl):
    """after a gather time-out the documented recovery is 'destroy it and form a new one' -- and kzg_mctx_destroy
    used to synchronise on the very stream the aborted collective occupied.  Here the exchange sits behind a 12 s spin kernel with
    a 50 ms deadline: the call fails after the bounded abort (~5 s), and destroying the group must return while the spin kernel is
    still running (the context is left behind, with a line on stderr)."""
    code = r"""
import ctypes, json, os, time
import kzg_amd
from kzg_amd import _lib as L
lib = L.load()
h = ctypes.c_void_p()
arr = (ctypes.c_int * 1)(0)
uid = ctypes.create_string_buffer(128)
assert lib.kzg_mctx_unique_id(uid) == 0
assert lib.kzg_mctx_create_rank(0, 0, 1, uid, ctypes.byref(h)) == 0
assert lib.kzg_mctx_set_option(h, b"always_gather", 1) == 0
srs = ctypes.c_void_p()
assert lib.kzg_srs_setup_g1_sharded(h, (7).to_bytes(32, "little"), L.FR_CANONICAL, 64, ctypes.byref(srs)) == 0
blob = b"".join(i.to_bytes(32, "little") for i in range(64))
out = ctypes.create_string_buffer(96)
assert lib.kzg_commit_coeff_sharded(h, srs, blob, 64, L.FR_CANONICAL, 0, out, L.G1_AFFINE_MONT) == 0
assert lib.kzg_mctx_set_option(h, b"gather_timeout_ms", 50) == 0
lib.kzg_test_mctx_inject_stall.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.kzg_test_mctx_inject_stall(h, 12000) == 0
t = time.time()
rc = lib.kzg_commit_coeff_sharded(h, srs, blob, 64, L.FR_CANONICAL, 0, out, L.G1_AFFINE_MONT)
res = {"rc": rc, "call_s": time.time() - t, "msg": lib.kzg_mctx_last_error(h).decode()}
t = time.time()
lib.kzg_msrs_free(h, srs)
lib.kzg_mctx_destroy(h)
res["destroy_s"] = time.time() - t
print(json.dumps(res), flush=True)
os._exit(0)
"""
    r = _child(code, {"KZG_AMD_LIBRARY": HOOKS, "KZG_DEBUG": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["rc"] == -4 and "did not complete within 50 ms" in out["msg"]
    assert out["call_s"] < 9 and out["destroy_s"] < 4, (out, r.stderr[-2500:])
    assert out["call_s"] + out["destroy_s"] < 11.5      # i.e. nobody waited for the 12 s kernel
    assert "left behind" in r.stderr


def test_rccl_forms_a_communicator_within_30_s(rccl_probe):
    """LAST: the environment itself.  A world-1 device group with the RCCL all-gather forced on, formed in a fresh child process by
    the session's probe (conftest.py): the commitment is right, and communicator formation (RCCL load + ncclCommInit*) took less
    than 30 s.  On failure the message names the phase that consumed the time (kzg_mctx_info / KZG_DEBUG lines) and carries the
    tail of RCCL's own log."""
    from oracle import c_oracle as C
    assert rccl_probe["ok"], "RCCL could not form a world-1 communicator: %s" % json.dumps(rccl_probe)[:6000]
    want = C.g1_mul(C.g1_generator(), C.poly_eval(list(range(1, 1025)), 0x5EED))
    assert rccl_probe["commit"] == want.hex()
    f = rccl_probe["formation"]
    assert 0 < f["formation_ms"] < 30000 and f["first_exchange"] < 30000, \
        "slow RCCL formation, phases (ms): %s\n%s" % (f, json.dumps(rccl_probe.get("diagnostics", {}))[:6000])
