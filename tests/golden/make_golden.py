#!/usr/bin/env python3
"""Generates tests/golden/*.json from the oracle's pure-Python big-int model (oracle/kzg_model.py).

No reference code exists in a runnable form in this environment (Rust, un-vendored crates), so these
vectors are NOT outputs of the reference binary: they are outputs of the independent python model of
the reference's algorithms + the public BLS12-381 definition, and they are cross-checked by two other
implementations (the C oracle on CPU, the HIP engine on GPU).  Encodings are canonical:
scalars = 32-byte little-endian hex, G1 points = 48-byte zcash compressed hex.

Run:  python tests/golden/make_golden.py     (deterministic; rewrites the JSON files in place)
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import kzg_model as M  # noqa: E402


def sc(x):
    return M.fr_to_le(x).hex()


def pt(P):
    return M.g1_to_compressed(P).hex()


def main():
    rng = random.Random(20261002)
    tau = rng.getrandbits(64)
    gs = M.setup_g1(tau, 40)

    # (ii) MSM vectors incl. zero scalars, r-1, repeated / identity points, all-equal scalars
    msm = []
    for n in (1, 2, 3, 16, 40):
        s = [rng.randrange(M.R) for _ in range(n)]
        msm.append({"name": f"random_{n}", "points": [pt(P) for P in gs[:n]], "scalars": [sc(x) for x in s],
                    "result": pt(M.g1_multi_exp(gs[:n], s))})
    P = gs[7]
    special_pts = [P, P, M.g1_neg(P), None, M.G1, P, None, M.g1_neg(P)]
    for name, s in (("special_points_equal_scalars", [5] * 8), ("special_points_mixed", [1, 1, 2, 9, 0, M.R - 2, 3, 4])):
        msm.append({"name": name, "points": [pt(Q) for Q in special_pts], "scalars": [sc(x) for x in s],
                    "result": pt(M.g1_multi_exp(special_pts, s))})
    for name, s in (("zeros", [0] * 16), ("r_minus_1", [M.R - 1] * 16), ("u64", [rng.getrandbits(64) for _ in range(16)]),
                    ("half_window", [0x8000] * 16), ("carry_chain", [int("7fff" * 15, 16)] * 16)):
        msm.append({"name": name, "points": [pt(Q) for Q in gs[:16]], "scalars": [sc(x) for x in s],
                    "result": pt(M.g1_multi_exp(gs[:16], s))})
    json.dump({"tau": sc(tau), "cases": msm}, open(os.path.join(HERE, "msm.json"), "w"), indent=0)

    # (iii) NTT vectors log n = 0..8 (9, 10 below)
    ntt = []
    for log_n in range(0, 9):
        xs = [rng.randrange(M.R) for _ in range(1 << log_n)]
        e = M.EvaluationDomain.from_coeffs(xs)
        e.fft()
        ntt.append({"log_n": log_n, "omega": sc(e.omega), "input": [sc(x) for x in xs], "fft": [sc(x) for x in e.coeffs]})
    # round 6: log n = 9, 10 (0..10) from a generator of their own, so that the vectors behind them keep their values, and
    # EvaluationDomain::ifft of every input (src/ft.rs:115-140) beside its fft
    rng2 = random.Random(20261004)
    for log_n in (9, 10):
        xs = [rng2.randrange(M.R) for _ in range(1 << log_n)]
        e = M.EvaluationDomain.from_coeffs(xs)
        e.fft()
        ntt.append({"log_n": log_n, "omega": sc(e.omega), "input": [sc(x) for x in xs], "fft": [sc(x) for x in e.coeffs]})
    for case in ntt:
        e = M.EvaluationDomain.from_coeffs([int.from_bytes(bytes.fromhex(h), "little") for h in case["input"]])
        e.ifft()
        case["ifft"] = [sc(x) for x in e.coeffs]
    json.dump({"cases": ntt}, open(os.path.join(HERE, "ntt.json"), "w"), indent=0)

    # (iv)+(v) KZG vectors: commit / create_witness (incl. degree-1 edge, wrong y) / batched / eval form
    kzg = {"tau": sc(tau), "srs_compressed": [pt(P) for P in gs[:16]]}
    params = M.KZGParams(gs[:16])
    prover = M.KZGProver(params)
    coeffs = [rng.randrange(M.R) for _ in range(13)]
    p = M.Polynomial(coeffs)
    x = rng.randrange(M.R)
    y = p.eval(x)
    kzg["coeff"] = {"coeffs": [sc(c) for c in coeffs], "commit": pt(prover.commit(p)), "x": sc(x), "y": sc(y),
                    "witness": pt(prover.create_witness(p, (x, y))), "wrong_y": sc((y + 1) % M.R)}
    p1 = M.Polynomial([3, 1] + [0] * 11)
    kzg["degree1"] = {"coeffs": [sc(3), sc(1)], "x": sc(1), "y": sc(4), "witness": pt(prover.create_witness(p1, (1, 4)))}
    xs = [rng.randrange(M.R) for _ in range(5)]
    ys = [p.eval(v) for v in xs]
    I, w = prover.create_witness_batched(p, xs, ys)
    kzg["batched"] = {"xs": [sc(v) for v in xs], "ys": [sc(v) for v in ys], "r": [sc(c) for c in I.coeffs], "w": pt(w)}
    d = 8
    pe = M.KZGParams(M.setup_g1(tau, d))
    lag = M.compute_lagrange_basis_g1(pe)
    evp = M.KZGProverEvalForm(pe, lag)
    ecoeffs = [rng.getrandbits(64) for _ in range(d)]
    ev = M.EvaluationDomain.from_coeffs(ecoeffs)
    ev.fft()
    kzg["eval"] = {"d": d, "lagrange_compressed": [pt(P) for P in lag], "coeffs": [sc(c) for c in ecoeffs],
                   "evals": [sc(c) for c in ev.coeffs], "commit": pt(evp.commit(ev)), "index": 3,
                   "witness": pt(evp.create_witness(ev, 3, fast=False))}
    json.dump(kzg, open(os.path.join(HERE, "kzg.json"), "w"), indent=0)
    if "--prod" in sys.argv or not os.path.exists(os.path.join(HERE, "prod.json")):
        prod_vectors()
    print("golden vectors written to", HERE)


def prod_vectors():
    """Production-path goldens (prod.json): 48 bytes each.  (tau, seed, n, distribution) -> compressed commit(p) = [p(tau)]G for
    p = the SplitMix64 coefficient stream of kzg_fill_random_fr, at the sizes where the engine runs its production MSM path
    (17-bit windows, two-level sort, 15 table rows: 2^17, 2^20, 2^21), plus one create_witness at 2^20.  Computed by the python
    model alone -- coefficient stream, Horner evaluation, one scalar multiplication of the generator -- in about a minute."""
    tau = 0x5EED5EED5EED5EED
    cases = []
    for log_n, seed, u64 in ((17, 1701, False), (20, 2001, False), (20, 2002, True), (21, 2101, False)):
        n = 1 << log_n
        ptau = M.splitmix_poly_eval(seed, n, tau, u64)
        cases.append({"log_n": log_n, "seed": seed, "u64_valued": u64, "p_tau": sc(ptau), "commit": pt(M.g1_mul(M.G1, ptau))})
    n, seed = 1 << 20, 2001
    x = M.splitmix_scalar(77, 0)
    y = M.splitmix_poly_eval(seed, n, x)
    ptau = M.splitmix_poly_eval(seed, n, tau)
    w = (ptau - y) * M.fr_inv((tau - x) % M.R) % M.R
    wit = {"log_n": 20, "seed": seed, "x": sc(x), "y": sc(y), "witness": pt(M.g1_mul(M.G1, w))}
    json.dump({"tau": sc(tau), "commits": cases, "witness": wit}, open(os.path.join(HERE, "prod.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
