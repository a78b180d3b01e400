"""Suite plumbing.  What it guarantees to a driver that runs `pytest -m gpu -x -q` on a box it controls and the builder does not:

  * ORDER (pytest_collection_modifyitems): in-process parity tests first (field / curve arithmetic, goldens, MSM, KZG, NTT, full
    sizes, verifier, validation, concurrency), then the in-process RCCL tests, then the device group over the test transport, then
    everything that starts bench.py / torch.distributed.run / rocprofv3 children, and at the very end the tests that JUDGE the
    environment (RCCL formation time).  Environment trouble can only cost the tests behind it, and the parity tests are in front.
  * TIME: every test has a wall-clock limit (default 150 s, `@pytest.mark.limit(seconds)` for more): SIGALRM raises inside
    Python code, and a C-level watchdog (faulthandler) dumps every thread's stack and ends the process 30 s later if the test sits
    in a native call that never returns -- a hang costs minutes and says where, not the driver's whole step limit.
  * ONE context: `engine` / `hooks_engine` are single objects for the whole session (they used to be one per importing module);
    `released_gpu` closes them before tests whose children need the chip's hardware queues to themselves.
  * RCCL: `rccl_probe` forms a world-1 device group in a CHILD process with a 45 s cap, once per session.  Tests that need a
    communicator take `need_rccl` and are SKIPPED with the probe's diagnostics when formation is broken or slow on this box
    (they cannot pass, and each would burn its whole time limit); the dedicated test in test_gpu_zz_environment.py then fails with
    the phase that consumed the time in its message.
"""
import faulthandler
import json
import os
import signal
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# a formation inside THIS process can never take longer than this (the library's default is 60 s)
os.environ.setdefault("KZG_COMM_TIMEOUT_MS", "30000")

DEFAULT_LIMIT_S = 150
ORDER = ["test_gpu_arith", "test_gpu_golden", "test_gpu_msm", "test_gpu_naf", "test_gpu_kzg", "test_gpu_ntt_poly", "test_gpu_ntt_large",
         "test_gpu_fullsize", "test_gpu_verify", "test_gpu_validation", "test_gpu_concurrent", "test_gpu_bench_paths",
         "test_gpu_mgpu", "test_gpu_mgpu_world", "test_gpu_bench_multi", "test_gpu_rccl_world", "test_gpu_zz_environment"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "limit(seconds): wall-clock limit of this test (default %d s)" % DEFAULT_LIMIT_S)


def pytest_collection_modifyitems(session, config, items):
    def key(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return (ORDER.index(mod) if mod in ORDER else (len(ORDER) - 2 if mod.startswith("test_gpu") else -1))
    items.sort(key=key)     # stable: the order inside a file is kept; CPU-only files (no GPU mark) come first


class TestTimeout(Exception):
    pass


@pytest.fixture(autouse=True)
def _wall_clock_limit(request):
    m = request.node.get_closest_marker("limit")
    limit = int(m.args[0]) if m else DEFAULT_LIMIT_S

    def on_alarm(signum, frame):
        raise TestTimeout("%s exceeded its %d s wall-clock limit" % (request.node.nodeid, limit))
    old = signal.signal(signal.SIGALRM, on_alarm)
    signal.alarm(limit)
    # a native call that never returns cannot be interrupted from Python: the C watchdog says where it sits and ends the run
    faulthandler.dump_traceback_later(limit + 30, exit=True, file=sys.stderr)
    try:
        yield
    finally:
        signal.alarm(0)
        faulthandler.cancel_dump_traceback_later()
        signal.signal(signal.SIGALRM, old)


# ---------------------------------------------------------------------------------------------------------------------
# the session's contexts
# ---------------------------------------------------------------------------------------------------------------------
_SESSION = {"engine": None, "hooks": None}


@pytest.fixture(scope="session")
def engine():
    import kzg_amd
    _SESSION["engine"] = kzg_amd.Engine(0)
    return _SESSION["engine"]


@pytest.fixture(scope="session")
def hooks_engine():
    from tests.gpu_common import HooksEngine
    _SESSION["hooks"] = HooksEngine(0)
    return _SESSION["hooks"]


def _release():
    for k in ("engine", "hooks"):
        if _SESSION[k] is not None:
            _SESSION[k].close()
            _SESSION[k] = None


@pytest.fixture
def released_gpu():
    """Closes the session's contexts (up to 20 streams each): children of this process get the chip's hardware queues to
    themselves.  Only the files at the END of ORDER use it -- their tests work in child processes and no in-process test follows."""
    _release()
    yield


def pytest_sessionfinish(session, exitstatus):
    try:
        _release()
    except Exception:
        pass
    try:    # RCCL's printf banner (NCCL_DEBUG=VERSION on some images) belongs before pytest's summary line, not after it
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


# ---------------------------------------------------------------------------------------------------------------------
# RCCL on this box
# ---------------------------------------------------------------------------------------------------------------------
PROBE_CAP_S = 45
_PROBE_CODE = r"""
import json, os, sys, time
t0 = time.time()
sys.path.insert(0, %r)
import kzg_amd
g = kzg_amd.DeviceGroup([0])
g.set_option("always_gather", 1)
s = g.setup(0x5EED, 1024)
c = g.commit(s, list(range(1, 1025)))
f = g.formation()
info = g.info()
s.free()
g.close()
print(json.dumps({"commit": c.hex(), "seconds": round(time.time() - t0, 2), "formation": f, "info": info}), flush=True)
""" % ROOT
_PROBE = {}


def run_rccl_probe(extra_env=None, cap_s=PROBE_CAP_S):
    """A world-1 device group with the RCCL all-gather forced on, in a child: {ok, seconds, formation, commit, diagnostics}."""
    log = os.path.join("/tmp", "kzg_rccl_probe_%d.%%p.log" % os.getpid())
    env = dict(os.environ, KZG_DEBUG="1", KZG_COMM_TIMEOUT_MS=str(1000 * (cap_s - 10)))
    if env.get("NCCL_DEBUG", "VERSION").upper() in ("VERSION", "WARN"):
        env["NCCL_DEBUG"] = "INFO"
        env.setdefault("NCCL_DEBUG_SUBSYS", "INIT,BOOTSTRAP,NET,ENV")
    env.setdefault("NCCL_DEBUG_FILE", log)
    env.update(extra_env or {})
    t0 = time.time()
    p = subprocess.Popen([sys.executable, "-c", _PROBE_CODE], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         start_new_session=True)
    try:
        out, err = p.communicate(timeout=cap_s)
        timed_out = False
    except subprocess.TimeoutExpired:
        timed_out = True
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            pass
        out, err = p.communicate()
    res = {"ok": False, "seconds": round(time.time() - t0, 2), "timed_out": timed_out, "rc": p.returncode}
    for ln in (out or "").splitlines():
        if ln.startswith("{"):
            res.update(json.loads(ln))
            res["ok"] = p.returncode == 0
    import glob
    logs = sorted(glob.glob(log.replace("%p", "*")))
    if not res["ok"] or res.get("formation", {}).get("formation_ms", 0) > 10000:
        rl = ""
        if logs:
            try:
                rl = open(logs[0], errors="replace").read()[-3000:]
            except OSError:
                pass
        res["diagnostics"] = {"stderr_tail": (err or "")[-3000:], "rccl_log_tail": rl,
                              "env": {k: v for k, v in env.items() if k.startswith(("NCCL_", "RCCL_", "KZG_", "GPU_MAX"))}}
    for f in logs:
        try:
            os.unlink(f)
        except OSError:
            pass
    return res


@pytest.fixture
def rccl_probe():
    if not _PROBE:
        _PROBE.update(run_rccl_probe())
    return _PROBE


@pytest.fixture
def need_rccl(rccl_probe):
    if not rccl_probe["ok"]:
        pytest.skip("RCCL cannot form a world-1 communicator on this box within %d s (see test_gpu_zz_environment.py): %s"
                    % (PROBE_CAP_S, json.dumps(rccl_probe)[:1500]))
    return rccl_probe
