"""bench.py's `paths` block (tools/benchlib/paths.py) on a small polynomial: every reading is checked against the oracle inside the
block, and the NTT roofline's kernel time -- HIP events over a counted number of profiled calls -- must agree with the wall time of the
same blocking call (a warm-up that changed the number of profiled calls once inflated it tenfold)."""
import pytest

import kzg_amd
from kzg_amd import _lib as L

pytestmark = pytest.mark.gpu


@pytest.mark.limit(300)
def test_paths_block_is_self_consistent(engine):
    from tools.benchlib.common import SEED, TAU, view
    from tools.benchlib.paths import measure_paths
    log_n, batch = 16, 2
    n = 1 << log_n
    scal = engine.alloc_scalars(n * batch)
    for b in range(batch):
        view(kzg_amd, scal, b * n, n).fill_random(SEED + 1000 * b)
    params = kzg_amd.setup(engine, TAU, n, g2_len=0)
    res = measure_paths(kzg_amd, L, engine, params, scal, n, log_n, budget_s=120.0)
    assert all(v is True for v in res["checked_against_oracle"].values()), res["checked_against_oracle"]
    assert res["commit_eval_equals_commit_coeff"] and res["witness_eval_equals_witness_coeff"]
    wall, kern = res["ntt_2e%d_ms" % log_n], res["ntt_roofline"]["kernel_ms"]
    assert 0.2 * wall < kern < 1.2 * wall, (wall, kern)
    assert abs(sum(res["ntt_roofline"]["kernels"].values()) - kern) < 0.01 * kern + 1e-3
    for k in ("commit_coeff_ms", "commit_eval_ms", "witness_coeff_ms", "witness_eval_ms"):
        assert 0 < res[k] < 50, (k, res[k])
    scal.free()
    params.gs.free()
