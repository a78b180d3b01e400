"""Shared fixtures for the -m gpu parity tests: one Engine per session, oracle helpers."""
import random

import pytest

from oracle import c_oracle as C
from oracle import kzg_model as M


@pytest.fixture(scope="session")
def engine():
    import kzg_amd
    e = kzg_amd.Engine(0)
    yield e
    e.close()


def rand_scalars(rng, n, kind="full"):
    if kind == "full":
        return [rng.randrange(M.R) for _ in range(n)]
    if kind == "u64":  # the reference's bench/test distribution (benches/commit_coeff_form.rs:16-21)
        return [rng.getrandbits(64) for _ in range(n)]
    raise ValueError(kind)


def oracle_srs(tau, n):
    """(blob, n) = setup(tau, n).gs as affine-Montgomery bytes, from the C oracle."""
    return C.setup_g1(tau, n)
