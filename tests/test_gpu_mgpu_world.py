"""The device group (kzg_amd/csrc/mgpu.hip) at world size 2, 3 and 8 -- on the ONE GPU of the test box.

RCCL refuses two ranks on one GPU, so these runs use the test transport of the hooks build (kzg_amd/csrc/test_transport.h: the
eight RCCL entry points over POSIX shared memory).  Everything above the transport is the product code at world > 1: the partition
rule, per-rank SRS shards, per-rank partial MSMs over different slices of the polynomial, the [world][batch + 1] record layout,
the sum of partials that come from different ranks, the status agreement with ONE rank failing (local phase; resource failure
before the exchange, after which the ranks hold buffers of different sizes), the persistent worker threads and the grouped
all-gather of the one-process mode.  RCCL itself runs at world 1 in tests/test_gpu_mgpu.py.

Parity: every rank's results equal the oracle's (C.msm_g1 over setup(tau, n).gs, [p(tau)]G, witness identities) and each other's."""
import ctypes
import hashlib
import json
import os
import random
import subprocess
import sys

import pytest

import kzg_amd
from kzg_amd import _lib as L
from kzg_amd.distributed import shard_range
from oracle import c_oracle as C
from oracle import kzg_model as M

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "mgpu_world_worker.py")
TAU = 0x0BADC0FFEE123457
N, D = 5003, 1 << 11


@pytest.fixture(autouse=True)
def _children_get_the_gpu(released_gpu):
    """every test here works in child processes: the session's own contexts (and their hardware queues) go first"""


def _env():
    env = dict(os.environ)
    env["KZG_TEST_SHM_TRANSPORT"] = "1"
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    return env


def _result(proc_out):
    lines = [ln for ln in proc_out.splitlines() if ln.startswith("RESULT ")]
    assert lines, proc_out[-3000:]
    return json.loads(lines[-1][7:])


def _horner(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % M.R
    return acc


def _expected(seed):
    """What every rank must print, from the oracle (the worker draws the same inputs from the same seed, in the same order)."""
    rng = random.Random(seed)
    G = C.g1_generator()
    blob = C.setup_g1(TAU, N)
    exp = {"blob": blob}
    polys = {m: [rng.randrange(M.R) for _ in range(m)] for m in (N, 777, 1, 0)}
    exp["commit"] = {str(m): C.msm_g1(blob[:96 * m], p).hex() for m, p in polys.items()}
    for m, p in polys.items():   # and the known-tau identity
        assert exp["commit"][str(m)] == C.g1_mul(G, C.poly_eval(p, TAU) if m else 0).hex()
    bp = [[rng.randrange(M.R) for _ in range(N)] for _ in range(3)] + [[0] * N, [M.R - 1] * N]
    exp["batch"] = [C.g1_mul(G, C.poly_eval(p, TAU)).hex() for p in bp]
    exp["batch_compressed"] = [M.g1_to_compressed(C.blob_to_point(bytes.fromhex(b))).hex() for b in exp["batch"]]
    exp["batch_short"] = [C.g1_mul(G, C.poly_eval(p[:4000], TAU)).hex() for p in bp]
    p = polys[N]
    x = rng.randrange(M.R)
    y = _horner(p, x)
    ptau = C.poly_eval(p, TAU)
    exp["witness"] = C.g1_mul(G, (ptau - y) * pow(TAU - x, -1, M.R) % M.R).hex()
    xs = [rng.randrange(M.R) for _ in range(7)]
    pts = [(v, _horner(p, v)) for v in xs]
    ztau = 1
    for v in xs:
        ztau = ztau * (TAU - v) % M.R
    exp["batched_points"] = pts
    exp["batched_ztau"], exp["ptau"] = ztau, ptau
    pe = [rng.randrange(M.R) for _ in range(D)]
    evals = C.fft(pe)
    exp["evals_sha"] = hashlib.sha256(kzg_amd.pack_scalars(evals)).hexdigest()
    _, _, omega = kzg_amd.compute_omega(D)
    petau = C.poly_eval(pe, TAU)
    exp["witness_eval"] = {str(m): C.g1_mul(G, (petau - evals[m]) * pow(TAU - pow(omega, m, M.R), -1, M.R) % M.R).hex()
                           for m in (0, 1, 777, D - 1)}
    exp["commit_eval"] = C.g1_mul(G, petau).hex()
    tp = [rng.randrange(M.R) for _ in range(3)]
    exp["tiny_commit"] = {str(m): C.g1_mul(G, C.poly_eval(tp[:m], TAU) if m else 0).hex() for m in (3, 2, 1, 0)}
    tx = rng.randrange(M.R)
    exp["tiny_witness"] = C.g1_mul(G, (C.poly_eval(tp, TAU) - _horner(tp, tx)) * pow(TAU - tx, -1, M.R) % M.R).hex()
    exp["tiny_poly"] = tp
    return exp


def _check(res, exp, world):
    G = C.g1_generator()
    assert res["world"] == world and "test-shm-transport" in res["info"]
    for (first, ln, sha), rk in zip(res["shards"], res["ranks"]):
        lo, hi = shard_range(N, rk, world)
        assert (first, ln) == (lo, hi - lo)
        assert sha == hashlib.sha256(exp["blob"][96 * lo:96 * hi]).hexdigest()   # the rank's shard is its range of setup(tau, n).gs
    assert res["commit"] == exp["commit"]
    assert res["batch_host"] == exp["batch"] and res["batch_device"] == exp["batch"]
    assert res["batch_compressed"] == exp["batch_compressed"]
    assert res["batch_device_short"] == exp["batch_short"]
    assert res["witness"] == exp["witness"] and res["witness_device"] == exp["witness"]
    assert res["witness_off_poly"] == "PointNotOnPolynomial"
    # create_witness_batched: the interpolant passes through the points with degree < k, and w = [(p(tau) - I(tau)) / Z(tau)]G
    w_hex, r_hex = res["witness_batched"]
    r = [int(c, 16) for c in r_hex]
    assert len(r) == 7
    for x, y in exp["batched_points"]:
        assert _horner(r, x) == y
    itau = _horner(r, TAU)
    assert w_hex == C.g1_mul(G, (exp["ptau"] - itau) * pow(exp["batched_ztau"], -1, M.R) % M.R).hex()
    assert res["evals_sha"] == exp["evals_sha"]
    assert res["witness_eval"] == exp["witness_eval"] and res["commit_eval"] == exp["commit_eval"]
    # one rank's failure is every rank's error, and the group stays in step
    assert res["local_failure"][0] == "EngineError" and "-3" in res["local_failure"][2], res["local_failure"]
    assert res["after_local_failure"] is True
    assert res["alloc_failure"][0] == "EngineError" and "-3" in res["alloc_failure"][2], res["alloc_failure"]
    assert res["after_alloc_failure"] is True and res["after_alloc_failure_66"] is True
    assert res["last_commit"] == exp["commit"][str(N)]
    # the SRS of 3 points: ranks from 3 on hold empty shards
    t = res["tiny"]
    for (first, ln), rk in zip(t["shards"], res["ranks"]):
        lo, hi = shard_range(3, rk, world)
        assert (first, ln) == (lo, hi - lo)
    assert t["commit"] == exp["tiny_commit"] and t["witness"] == exp["tiny_witness"]
    tp = exp["tiny_poly"]
    ptau = C.poly_eval(tp, TAU)
    for k in (1, 2):
        w_hex, r_hex, pts = t["batched_%d" % k]
        r = [int(c, 16) for c in r_hex]
        pts = [(int(a, 16), int(b, 16)) for a, b in pts]
        assert len(r) == (2 if k == 1 else k)       # the one-point interpolant is X + (y - x), degree 1
        z = 1
        for x, y in pts:
            z = z * (TAU - x) % M.R
            if k > 1:
                assert _horner(r, x) == y
        if k == 1:
            assert r == [(pts[0][1] - pts[0][0]) % M.R, 1]
        assert w_hex == C.g1_mul(G, (ptau - _horner(r, TAU)) * pow(z, -1, M.R) % M.R).hex()


def _unique_id():
    """From the hooks build with the transport selected -- in a child process: this process' copy of the hooks library may already
    have adopted the real RCCL (tests/test_gpu_mgpu.py)."""
    code = ("import sys, ctypes; sys.path.insert(0, %r); from kzg_amd import _lib as L; "
            "lib = L.load(%r); b = ctypes.create_string_buffer(128); assert lib.kzg_mctx_unique_id(b) == 0; print(b.raw.hex())"
            % (ROOT, os.path.join(ROOT, "kzg_amd", "libkzg_mi355x_hooks.so")))
    p = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    return p.stdout.strip().splitlines()[-1]


@pytest.mark.limit(300)
@pytest.mark.parametrize("world", [2, 3, 8])
def test_process_per_gpu_group_on_one_gpu(world):
    seed = 100 + world
    uid = _unique_id()
    import tempfile
    files = [(tempfile.TemporaryFile("w+"), tempfile.TemporaryFile("w+")) for _ in range(world)]   # (pipes could fill while waiting)
    procs = [subprocess.Popen([sys.executable, WORKER, "rank", str(r), str(world), uid, str(seed)], env=_env(),
                              stdout=files[r][0], stderr=files[r][1], text=True) for r in range(world)]
    outs = []
    try:
        for p, (fo, fe) in zip(procs, files):
            p.wait(timeout=280)
            fo.seek(0)
            fe.seek(0)
            outs.append((p.returncode, fo.read(), fe.read()))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for rk, (rc, so, se) in enumerate(outs):
        assert rc == 0, f"rank {rk}: rc {rc}\n{se[-3000:]}"
    exp = _expected(seed)
    results = [_result(so) for _, so, _ in outs]
    for rk, res in enumerate(results):
        assert res["ranks"] == [rk]
        _check(res, exp, world)
    # every rank reports the failing rank's error, naming it; the failing rank reports its own
    for rk, res in enumerate(results):
        if rk == world - 1:
            assert "injected" in res["local_failure"][2] and "injected" in res["alloc_failure"][2]
        else:
            assert f"rank {world - 1}" in res["local_failure"][2] and f"rank {world - 1}" in res["alloc_failure"][2]


@pytest.mark.limit(300)
@pytest.mark.parametrize("world", [2, 4])
def test_one_process_group_on_one_gpu(world):
    """kzg_mctx_create(devices, n > 1): one process, `world` contexts, the persistent worker threads, the grouped all-gather."""
    seed = 200 + world
    p = subprocess.run([sys.executable, WORKER, "one", str(world), str(seed)], env=_env(), capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-3000:]
    res = _result(p.stdout)
    assert res["ranks"] == list(range(world)) and "mode=one-process" in res["info"]
    _check(res, _expected(seed), world)
