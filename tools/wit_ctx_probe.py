import ctypes, os, sys, time
sys.path.insert(0, os.getcwd())
import kzg_amd
from kzg_amd import _lib as L
from tools.benchlib.common import view, timeit, TAU, SEED
e = kzg_amd.Engine(0)
n = 1 << 20; batch = 64
R = kzg_amd.api.R_MODULUS
scal = e.alloc_scalars(n * batch)
for b in range(batch):
    view(kzg_amd, scal, b * n, n).fill_random(SEED + 1000 * b)
params = kzg_amd.setup(e, TAU, n, g2_len=0)
srs = params.gs
out = ctypes.create_string_buffer(96 * batch)
def step():
    assert e.lib.kzg_msm_g1_batch(e.ctx, srs.handle, 0, scal.ptr, n, batch, scal.sfmt, L.IN_DEVICE, out, L.G1_AFFINE_MONT) == 0
coeffs = view(kzg_amd, scal, 0, n)
o1 = ctypes.create_string_buffer(96)
def commit():
    assert e.lib.kzg_commit_coeff(e.ctx, srs.handle, coeffs.ptr, n, coeffs.sfmt, L.IN_DEVICE, o1, L.G1_AFFINE_MONT) == 0
x = kzg_amd.splitmix_scalar(99, 0); y = e.poly_eval(coeffs, x)
xb, yb = (x % R).to_bytes(32, "little"), (y % R).to_bytes(32, "little")
def witness():
    assert e.lib.kzg_witness_coeff(e.ctx, srs.handle, coeffs.ptr, n, xb, yb, coeffs.sfmt, L.IN_DEVICE, o1, L.G1_AFFINE_MONT) == 0
print("fresh engine: commit %.3f witness %.3f commit %.3f witness %.3f" % (timeit(commit), timeit(witness), timeit(commit), timeit(witness)))
step(); step()
print("after 2 batches: commit %.3f witness %.3f commit %.3f witness %.3f" % (timeit(commit), timeit(witness), timeit(commit), timeit(witness)))
print("reps=10: commit %.3f witness %.3f" % (timeit(commit, reps=10), timeit(witness, reps=10)))
lag = kzg_amd.setup_lagrange(e, TAU, n)
print("after lagrange setup: commit %.3f witness %.3f commit %.3f witness %.3f" % (timeit(commit), timeit(witness), timeit(commit), timeit(witness)))
e.prof_enable(True); e.prof_reset()
for _ in range(3): witness()
pr = e.prof_all(); e.prof_enable(False)
print({k: round(v[1]/3, 4) for k, v in sorted(pr.items(), key=lambda kv: -kv[1][1])[:8]})
# the bench's own measure_paths, twice
from tools.benchlib.paths import measure_paths
for i in range(2):
    r = measure_paths(kzg_amd, L, e, params, scal, n, 20)
    print("measure_paths #%d:" % i, {k: v for k, v in r.items() if k.endswith("_ms")})
print("after measure_paths: commit %.3f witness %.3f commit %.3f witness %.3f" % (timeit(commit), timeit(witness), timeit(commit), timeit(witness)))
