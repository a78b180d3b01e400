#!/usr/bin/env python3
"""Times the GPU verifier: kzg_verify_eval for batches of independent openings (one thread per pairing check)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kzg_amd

R = kzg_amd.api.R_MODULUS
e = kzg_amd.Engine(0)
tau = 0x5EED5EED
params = kzg_amd.setup(e, tau, 64)
prover, verifier = kzg_amd.KZGProver(params), kzg_amd.KZGVerifier(params)
p = kzg_amd.Polynomial([kzg_amd.splitmix_scalar(1, i) for i in range(64)])
c = prover.commit(p)
base = []
for i in range(8):
    x = 1000 + i
    y = p.eval(e, x)
    base.append(((x, y), prover.create_witness(p, (x, y))))
out = {}
for count in [int(a) for a in sys.argv[1:]] or [1, 64, 1024, 16384]:
    pts = [base[i % 8][0] if i % 5 else (base[i % 8][0][0], (base[i % 8][0][1] + 1) % R) for i in range(count)]
    ws = [base[i % 8][1] for i in range(count)]
    cs = [c] * count
    got = verifier.verify_eval_many(pts, cs, ws)
    assert got == [bool(i % 5) for i in range(count)]
    t0 = time.perf_counter()
    reps = 3 if count <= 1024 else 1
    for _ in range(reps):
        verifier.verify_eval_many(pts, cs, ws)
    dt = (time.perf_counter() - t0) / reps
    out[count] = {"ms": round(dt * 1e3, 2), "checks_per_s": round(count / dt, 1)}
    print(count, out[count], flush=True)
print(json.dumps(out))
