import random, ctypes, sys
sys.path.insert(0, "/root/repo")
import kzg_amd
from oracle import c_oracle as C
from oracle import kzg_model as M
tau = 0x5EED1234
n, k = 1 << 12, 16
e = kzg_amd.Engine(0)
e.set_option("witness_cache_slots", 2)
params = kzg_amd.setup(e, tau, n, g2_len=0)
prover = kzg_amd.KZGProver(params)
rng = random.Random(4242)
sets = [[rng.randrange(M.R) for _ in range(k)] for _ in range(3)]
for i, xs in enumerate((sets[0], sets[0], sets[1], sets[0], sets[2], sets[1], sets[0])):
    coeffs = [rng.randrange(M.R) for _ in range(n)]
    ys = [C.poly_eval(coeffs, x) for x in xs]
    try:
        wit = prover.create_witness_batched(kzg_amd.Polynomial(coeffs), xs, ys)
        print(i, "ok")
    except Exception as ex:
        print(i, "FAIL", type(ex).__name__, ex)
    h, m = ctypes.c_uint64(), ctypes.c_double()
    e.lib.kzg_prof_get(e.ctx, b"point_set_cache", ctypes.byref(h), ctypes.byref(m)); print("   hits", h.value, "misses", m.value)
