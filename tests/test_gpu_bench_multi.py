"""bench.py's N > 1 control flow on a one-GPU box: two ranks under torch.distributed.run, both on device 0 (KZG_BENCH_SHARED_GPU:
gloo barriers instead of RCCL, which refuses two ranks on one GPU).  What it pins: the launch contract (RANK / WORLD_SIZE from the
environment, one JSON line from rank 0), the default N > 1 mode (replicas: commitments sharded over the ranks, no data-path
collective, "scaling": "weak"), the max-over-ranks time, and every commitment of the last step against [p(tau)]G by the oracle."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_replicas_on_one_gpu():
    env = dict(os.environ, KZG_BENCH_SHARED_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29617", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8",
           "--log-n", "16", "--check", "--no-paths"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
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
