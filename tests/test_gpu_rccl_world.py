"""The device group over REAL RCCL at world > 1: one process per visible GPU, the product library (no test transport), no torch.
Skipped on a one-GPU box -- costs nothing there -- and runs by itself the day a multi-GPU box appears:
commit / commit_batch (host- and device-resident) / create_witness / create_witness_batched of every rank against the oracle
([p(tau)]G, [(p(tau) - y)/(tau - x)]G, [(p(tau) - I(tau))/Z(tau)]G), and the ranks against each other."""
import json
import os
import random
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "rccl_world_worker.py")


def _device_count():
    import kzg_amd
    return kzg_amd.load().kzg_device_count()


@pytest.fixture(autouse=True)
def _children_get_the_gpus(released_gpu):
    """the ranks are child processes: the session's own contexts (and their hardware queues) go first"""


@pytest.mark.gpu
@pytest.mark.limit(400)
def test_real_rccl_group_over_all_visible_gpus():
    world = _device_count()
    if world < 2:
        pytest.skip("one GPU visible: a real-RCCL group at world > 1 needs at least two (the one-GPU suites run the same product code "
                    "over the test transport: tests/test_gpu_mgpu_world.py)")
    _run_world_and_check(world, 29877)


@pytest.mark.gpu
@pytest.mark.limit(300)
def test_rccl_world_worker_and_checks_at_world_1(need_rccl):
    """The same worker and the same checks with ONE rank (runs on every box): what the multi-GPU test above executes, minus the fabric."""
    _run_world_and_check(1, 29879)


def _run_world_and_check(world, port):
    from oracle import c_oracle as C
    from oracle import kzg_model as M
    seed = 4242
    files = [(tempfile.TemporaryFile("w+"), tempfile.TemporaryFile("w+")) for _ in range(world)]
    env = dict(os.environ, KZG_DEBUG="1")
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), str(port), str(seed)], env=env, stdout=files[r][0], stderr=files[r][1],
                              text=True, start_new_session=True) for r in range(world)]
    outs = []
    try:
        for p, (fo, fe) in zip(procs, files):
            p.wait(timeout=360)
            fo.seek(0)
            fe.seek(0)
            outs.append((p.returncode, fo.read(), fe.read()))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for rk, (rc, so, se) in enumerate(outs):
        assert rc == 0, f"rank {rk}: rc {rc}\n{se[-3000:]}"
    res = [json.loads([ln for ln in so.splitlines() if ln.startswith("{")][-1]) for _, so, _ in outs]
    env = [r for r in res if "environment" in r]
    if env:      # RCCL did not give the ranks a communicator within the library's deadlines: the box's fabric / bootstrap, not the product
        pytest.skip("RCCL could not form a world-%d communicator on this box: %s" % (world, json.dumps(env)[:2000]))
    # the oracle's side, from the same seed
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import rccl_world_worker as W
    R, TAU, N = M.R, W.TAU, W.N
    rng = random.Random(seed)
    p = [rng.randrange(R) for _ in range(N)]
    G = C.g1_generator()
    ptau = C.poly_eval(p, TAU)
    batch = 4
    bp = [[rng.randrange(R) for _ in range(N)] for _ in range(batch - 2)] + [[0] * N, [R - 1] * N]
    want_batch = [C.g1_mul(G, C.poly_eval(q, TAU)).hex() for q in bp]
    x = rng.randrange(R)
    y = C.poly_eval(p, x)
    xs = [rng.randrange(R) for _ in range(5)]
    for rk, r in enumerate(res):
        assert r["rank"] == rk and r["world"] == world and r["torch_imported"] is False
        assert "rccl=" in r["info"] and "test-shm-transport" not in r["info"]
        assert r["commit"] == C.g1_mul(G, ptau).hex()
        assert r["batch"] == want_batch and r["batch_device"] == want_batch
        assert r["witness"] == C.g1_mul(G, (ptau - y) * pow(TAU - x, -1, R) % R).hex()
        assert r["witness_off_poly"] == "PointNotOnPolynomial"
        w_hex, r_hex = r["witness_batched"]
        I = [int(c, 16) for c in r_hex]
        z = 1
        for v in xs:
            z = z * (TAU - v) % R
            assert C.poly_eval(I, v) == C.poly_eval(p, v)
        assert w_hex == C.g1_mul(G, (ptau - C.poly_eval(I, TAU)) * pow(z, -1, R) % R).hex()
        assert 0 < r["formation"]["formation_ms"] < 60000
