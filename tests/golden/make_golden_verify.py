#!/usr/bin/env python3
"""Generates tests/golden/verify.json from the oracle's python model (oracle/pairing_model.py + kzg_model.py):
G2 parameters, G2 multi-exponentiation, pairing-product verdicts and verifier scenarios.  As for the other golden
files these are NOT outputs of the reference binary (it cannot be built here); they are outputs of the independent
python model, cross-checked by the host build of kzg_amd/csrc/tower.h (tests/test_host_tower.py) and the HIP engine.
Encodings: scalars 32-byte LE hex, G1 48-byte / G2 96-byte zcash compressed hex.

Run:  python tests/golden/make_golden_verify.py     (deterministic; rewrites verify.json in place)
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import kzg_model as M, pairing_model as P  # noqa: E402


def sc(x):
    return M.fr_to_le(x).hex()


def p1(pt):
    return M.g1_to_compressed(pt).hex()


def p2(pt):
    return P.g2_to_compressed(pt).hex()


def main():
    rng = random.Random(20261003)
    tau = rng.getrandbits(64)
    n = 12
    params = P.setup(tau, n)
    out = {"tau": sc(tau), "n": n, "hs_compressed": [p2(h) for h in params.hs[:6]],
           "hs1_uncompressed": P.g2_to_uncompressed(params.hs[1]).hex()}
    s = [rng.randrange(M.R) for _ in range(6)]
    out["msm_g2"] = {"scalars": [sc(x) for x in s], "result": p2(P.g2_multi_exp(params.hs[:6], s))}
    out["lagrange_h_d4"] = [p2(h) for h in P.lagrange_basis_g2_known_tau(tau, 4)]

    a, b = rng.randrange(M.R), rng.randrange(M.R)
    Pa, Qb = M.g1_mul(M.G1, a), P.g2_mul(P.G2, b)
    checks = [[(Pa, Qb), (M.g1_neg(M.g1_mul(M.G1, a * b % M.R)), P.G2)],
              [(Pa, Qb), (M.g1_neg(M.g1_mul(M.G1, (a * b + 1) % M.R)), P.G2)],
              [(None, Qb), (Pa, None)],
              [(M.G1, P.G2), (M.G1, P.G2)]]
    out["pairing_checks"] = [{"g1": [p1(x) for x, _ in c], "g2": [p2(y) for _, y in c],
                              "is_one": P.pairing_product_is_one(c)} for c in checks]

    prover, verifier = M.KZGProver(params), P.KZGVerifier(params)
    coeffs = [rng.randrange(M.R) for _ in range(9)]
    poly = M.Polynomial(coeffs)
    c = prover.commit(poly)
    x = rng.randrange(M.R)
    y = poly.eval(x)
    w = prover.create_witness(poly, (x, y))
    tuples = [(x, y), (x, (y + 1) % M.R), ((x + 1) % M.R, y)]
    out["verify_eval"] = {"coeffs": [sc(v) for v in coeffs], "commitment": p1(c), "witness": p1(w),
                          "points": [[sc(u), sc(v)] for u, v in tuples],
                          "ok": [verifier.verify_eval(t, c, w) for t in tuples]}
    xs = [rng.randrange(M.R) for _ in range(4)]
    r, wb = prover.create_witness_batched(poly, xs, [poly.eval(v) for v in xs])
    xs_bad = list(xs)
    xs_bad[2] = (xs_bad[2] + 1) % M.R
    out["verify_eval_batched"] = {"xs": [sc(v) for v in xs], "xs_bad": [sc(v) for v in xs_bad],
                                  "r": [sc(v) for v in r.slice_coeffs()], "w": p1(wb),
                                  "ok": verifier.verify_eval_batched(xs, c, wb, r),
                                  "ok_bad": verifier.verify_eval_batched(xs_bad, c, wb, r)}
    json.dump(out, open(os.path.join(HERE, "verify.json"), "w"), indent=0)
    print("wrote verify.json", out["verify_eval"]["ok"], out["verify_eval_batched"]["ok"], out["verify_eval_batched"]["ok_bad"],
          [k["is_one"] for k in out["pairing_checks"]])


if __name__ == "__main__":
    main()
