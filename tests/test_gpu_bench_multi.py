"""bench.py's N > 1 control flow on a one-GPU box: two ranks under torch.distributed.run (the launcher only: bench.py itself imports no
torch), both on device 0 (KZG_BENCH_SHARED_GPU; RCCL refuses two ranks on one GPU, so the device-group tests use the test transport).  What it pins: the launch contract (RANK / WORLD_SIZE from the
environment, one JSON line from rank 0), the default N > 1 mode (replicas: commitments sharded over the ranks, no data-path
collective, "scaling": "weak"), the max-over-ranks time, and every commitment of the last step against [p(tau)]G by the oracle."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _children_get_the_gpu(released_gpu):
    """every test here works in child processes: the session's own contexts (and their hardware queues) go first"""


@pytest.mark.gpu
@pytest.mark.limit(400)
def test_bench_two_ranks_replicas_on_one_gpu():
    env = dict(os.environ, KZG_BENCH_SHARED_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29617", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8",
           "--log-n", "16", "--check", "--no-paths", "--no-sharded-block"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=380)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2 and d["warmup"] == 1
    assert "replicas" in d["config"]["workload"] and d["config"]["mode"] == "replicas"
    assert d["all_results_match_known_tau"] is True
    assert d["value"] > 0 and d["unit"] == "commitments/s"
    # whole-job aggregate: both ranks' commitments over the slowest rank's time
    assert abs(d["value"] - 2 * 8 * 2 / (d["ms_per_step"] * 2 / 1e3)) / d["value"] < 0.02
    assert d["timed_results_checked"]["ok"] is True and d["timed_results_checked"]["every_rank"] is True
    # no torch in any rank: the library binds the system's HIP runtime, the ranks talk over tools/benchlib/control.py's TCP star
    assert "hip=" in d["hip_runtime"]["library"] and d["hip_runtime"]["torch_imported"] is False
    assert "torch" not in d["hip_runtime"]["library"] and "TCP star" in d["hip_runtime"]["control_plane"]


_BLOCK = {}


def _bench_with_sharded_block():
    """bench.py at N = 1 with the `sharded` block forced on, once for the two tests below."""
    if not _BLOCK:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "8", "--log-n", "16", "--no-paths",
               "--no-cpu-baseline", "--sharded-block", "--sharded-batch", "4", "--sharded-steps", "2", "--sharded-timeout", "90"]
        r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ), capture_output=True, text=True, timeout=280)
        _BLOCK.update(rc=r.returncode, stderr=r.stderr[-2000:], lines=[ln for ln in r.stdout.splitlines() if ln.startswith("{")],
                      stdout=r.stdout[-2000:])
    return _BLOCK


@pytest.mark.gpu
@pytest.mark.limit(300)
def test_bench_line_survives_the_sharded_block_and_imports_no_torch():
    """The N = 1 run imports no torch, says which HIP runtime the library is bound to, and sample-checks the
    timed region's own results unconditionally (weak #3).  The `sharded` block runs in a fresh child process of the bench: whatever
    that child does (RCCL took five minutes to form a communicator on the driver's box), the parent prints exactly one
    line, exits 0, and the line carries the block -- measured, or with a note and diagnostics."""
    b = _bench_with_sharded_block()
    assert b["rc"] == 0, b["stderr"]
    assert len(b["lines"]) == 1, b["stdout"]
    d = json.loads(b["lines"][0])
    assert d["config"]["mode"] == "single" and d["n_gpus"] == 1
    assert d["hip_runtime"]["torch_imported"] is False and "hip=" in d["hip_runtime"]["library"]
    assert d["timed_results_checked"]["ok"] is True
    sh = d["sharded"]
    assert sh["child"]["process"] == "fresh child per rank" and sh["child"]["wall_s"] < 100
    if "note" in sh or "error" in sh:       # the environment's fault is judged by the next test; here: it is DIAGNOSED
        assert sh["diagnostics"]["env"].get("NCCL_SOCKET_IFNAME") and "stderr_tail" in sh["diagnostics"], sh


@pytest.mark.gpu
@pytest.mark.limit(300)
def test_bench_sharded_block_world1(need_rccl):
    """The default N > 1 line carries `sharded` = {strong, config5}: the sharded-SRS + RCCL design north_star names, measured by
    the same ranks right after the replicas' timed region.  Here at world size 1 with the block forced on
    (--sharded-block: RCCL all-gather forced on in a group of one): presence, the RCCL it ran on, what forming the communicator
    cost, and every commitment of each mode's last step against [p(tau)]G by the oracle."""
    b = _bench_with_sharded_block()
    assert b["rc"] == 0 and len(b["lines"]) == 1, b["stderr"]
    sh = json.loads(b["lines"][0])["sharded"]
    assert "note" not in sh and "error" not in sh, sh
    assert sh["child"]["rc"] == 0 and not sh["child"]["timed_out"]
    assert 0 < sh["formation"]["formation_ms"] < 30000 and sh["formation"]["first_exchange"] > 0, sh["formation"]
    assert sh["rccl_ranks"] == 1 and "rccl=" in sh["rccl"]
    assert sh["strong"]["terms_per_rank"] == 1 << 16 and sh["strong"]["scaling"] == "strong"
    assert sh["config5"]["terms_per_rank"] == 1 << 21 and sh["config5"]["polynomial_coefficients"] == 1 << 21
    for mode in ("strong", "config5"):
        assert sh[mode]["all_results_match_known_tau"] is True and sh[mode]["value"] > 0


@pytest.mark.gpu
@pytest.mark.limit(420)
def test_bench_line_carries_live_pmc_traffic():
    """roofline.traffic: HBM bytes per launch of the dominant kernel from the PMC counters, collected DURING the bench run by two
    child rocprofv3 passes (FETCH_SIZE, WRITE_SIZE; gfx950 correction) -- not a number copied from profiles/.  Small size here;
    the figure must be a positive byte count of the order of (table rows x 128 B + 4 B) per sorted entry."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "8", "--log-n", "16", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ), capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    rf = d["roofline"]
    assert isinstance(rf["traffic"], int) and rf["traffic"] > 0, rf.get("traffic_note")
    tm = rf["traffic_measured"]
    assert tm["launches_sampled"] >= 1 and "rocprofv3" in tm["method"]
    entries = (1 << 16) * d["config"]["windows"]
    assert 64 * entries < rf["traffic"] < 512 * entries
    assert rf["frac_of_nominal"] > 0 and d["paths"]["ntt_2e16_ms"] > 0
    # BASELINE configs[2] from 16 threads (fft in place, then the Lagrange-SRS commit): every commitment equals the coefficient-form one
    assert d["paths"]["blocking_callers_16_fft_commit_eval_match_commit_coeff"] is True
    assert d["paths"]["blocking_callers_16_fft_commit_eval_per_s"] > 0


def _torchrun(nproc, port, bench_args, timeout=280):
    """bench.py under torch.distributed.run with `nproc` ranks on the one GPU: the TCP control plane, and the device group over the
    hooks build's test transport (kzg_amd/csrc/test_transport.h -- RCCL refuses two ranks on one GPU)."""
    env = dict(os.environ, KZG_BENCH_SHARED_GPU="1", MASTER_ADDR="127.0.0.1", KZG_TEST_SHM_TRANSPORT="1",
               KZG_AMD_LIBRARY=os.path.join(ROOT, "kzg_amd", "libkzg_mi355x_hooks.so"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + bench_args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.limit(300)
def test_bench_default_line_at_two_ranks_carries_the_sharded_block():
    """What the driver's `bench.py --gpus N` run prints at N > 1, executed with two ranks: value = replicas, and the `sharded` block
    (strong: each commitment split over the ranks; config5: 2^21 terms per rank) measured through the device group at world 2 with
    every commitment of each mode's last step checked against [p(tau)]G."""
    d = _torchrun(2, 29641, ["--steps", "2", "--warmup", "1", "--batch", "8", "--log-n", "16", "--no-paths", "--sharded-batch", "4",
                             "--sharded-steps", "2"])
    assert d["n_gpus"] == 2 and d["config"]["mode"] == "replicas"
    assert d["timed_results_checked"]["ok"] is True and d["timed_results_checked"]["every_rank"] is True   # the unconditional sample
    sh = d["sharded"]
    assert "note" not in sh and "error" not in sh, sh
    assert sh["rccl_ranks"] == 2 and "test-shm-transport" in sh["rccl"]
    assert sh["strong"]["terms_per_rank"] == 1 << 15 and sh["strong"]["scaling"] == "strong"
    assert sh["config5"]["terms_per_rank"] == 1 << 21 and sh["config5"]["polynomial_coefficients"] == 1 << 22
    for mode in ("strong", "config5"):
        assert sh[mode]["all_results_match_known_tau"] is True and sh[mode]["value"] > 0


@pytest.mark.gpu
@pytest.mark.limit(300)
@pytest.mark.parametrize("nproc,flag", [(4, "--strong"), (2, "--weak")])
def test_bench_sharded_modes_as_the_timed_region(nproc, flag):
    """--strong / --weak at N > 1: the device group IS the timed region (one commitment sharded over the ranks, the exchange inside
    every step), every commitment of the last step checked on every rank."""
    d = _torchrun(nproc, 29643 + nproc, ["--steps", "2", "--warmup", "1", "--batch", "4", "--log-n", "16", "--no-paths", "--check", flag])
    assert d["n_gpus"] == nproc and d["config"]["mode"] == flag[2:]
    assert d["scaling"] == ("strong" if flag == "--strong" else "weak")
    assert d["all_results_match_known_tau"] is True and d["timed_results_checked"]["ok"] is True
    assert d["value"] > 0

