"""The checkers of the bench line: every timed result is compared with the ORACLE's answer after its timer has stopped.  With
cpu_pool the only bench module that imports oracle/; nothing here is ever inside a timed region."""
from .common import TAU, view


def _C():
    from oracle import c_oracle as C
    return C


# the oracle's primitives the other bench modules compare against (they never import oracle/ themselves)
def g1_generator():
    return _C().g1_generator()


def g1_mul(p, k):
    return _C().g1_mul(p, k)


def poly_eval_bytes(blob, n, x):
    return _C().poly_eval_bytes(blob, n, x)


def poly_eval(coeffs, x):
    return _C().poly_eval(coeffs, x)


def known_tau_partials(kzg_amd, scal, n_local, lo, which):
    """tau^lo * p_slice(tau) for the polynomials `which` of a [batch][n_local] device array, by the ORACLE (coefficients
    downloaded, its Horner loop): this rank's share of p_b(tau)."""
    from oracle import c_oracle as C
    R = kzg_amd.api.R_MODULUS
    return [pow(TAU, lo, R) * C.poly_eval_bytes(view(kzg_amd, scal, b * n_local, n_local).download(), n_local, TAU) % R if n_local else 0
            for b in which]


def check_known_tau(kzg_amd, job, scal, n_local, lo, out_raw, which, spans_ranks):
    """out[b] == [p_b(tau)]G for b in `which`; p_b(tau) = sum over ranks of the slices' shares when a commitment spans the ranks.
    The right-hand side is the oracle's alone; every rank checks, all must agree."""
    from oracle import c_oracle as C
    R = kzg_amd.api.R_MODULUS
    mine = known_tau_partials(kzg_amd, scal, n_local, lo, which)
    if spans_ranks and job.world > 1:
        allv = job.gather_objects(mine)
        mine = [sum(v[i] for v in allv) % R for i in range(len(which))]
    G = C.g1_generator()
    ok = all(out_raw[96 * b: 96 * b + 96] == C.g1_mul(G, mine[i]) for i, b in enumerate(which))
    return job.all_agree(ok)


