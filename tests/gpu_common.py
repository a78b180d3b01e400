"""Shared helpers of the -m gpu parity tests: the hooks-build context, oracle helpers.  (The `engine` / `hooks_engine` fixtures
live in conftest.py: one object each for the whole session.)"""
import random

import pytest

from oracle import c_oracle as C
from oracle import kzg_model as M


class HooksEngine:
    """A context in the -DKZG_TEST_HOOKS build of the library (kzg_amd/libkzg_mi355x_hooks.so): the product library does not
    export the unit-test hooks of include/kzg_mi355x_test.h."""

    def __init__(self, device=0):
        import ctypes
        import os
        import kzg_amd
        kzg_amd.load()  # the product library first: it has asked for the hardware queues
        so = os.path.join(os.path.dirname(kzg_amd.__file__), "libkzg_mi355x_hooks.so")
        self.lib = ctypes.CDLL(so)
        vp, sz, i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        for name, args in {"kzg_test_fr_mul": [vp, vp, vp, sz, vp], "kzg_test_fq_mul": [vp, vp, vp, sz, vp],
                           "kzg_test_fr_inv": [vp, vp, sz, vp], "kzg_test_g1_add": [vp, vp, vp, sz, vp],
                           "kzg_test_g1_mul": [vp, vp, vp, sz, vp], "kzg_ctx_create": [i32, ctypes.POINTER(vp)],
                           "kzg_ctx_destroy": [vp], "kzg_test_mctx_inject_failure": [vp, i32],
                           "kzg_test_mctx_inject_alloc_failure": [vp], "kzg_test_mctx_inject_stall": [vp, i32]}.items():
            f = getattr(self.lib, name)
            f.argtypes = args
            f.restype = None if name == "kzg_ctx_destroy" else i32
        self.lib.kzg_last_error.argtypes = [vp]
        self.lib.kzg_last_error.restype = ctypes.c_char_p
        self.ctx = vp()
        rc = self.lib.kzg_ctx_create(device, ctypes.byref(self.ctx))
        assert rc == 0, f"kzg_ctx_create in the hooks library failed: {rc}"

    def last_error(self):
        return (self.lib.kzg_last_error(self.ctx) or b"").decode()

    def close(self):
        if self.ctx:
            self.lib.kzg_ctx_destroy(self.ctx)
        self.ctx = None


def rand_scalars(rng, n, kind="full"):
    if kind == "full":
        return [rng.randrange(M.R) for _ in range(n)]
    if kind == "u64":  # the reference's bench/test distribution (benches/commit_coeff_form.rs:16-21)
        return [rng.getrandbits(64) for _ in range(n)]
    raise ValueError(kind)


def oracle_srs(tau, n):
    """(blob, n) = setup(tau, n).gs as affine-Montgomery bytes, from the C oracle."""
    return C.setup_g1(tau, n)


def build_cpp_mirror(tmp_path):
    """include/kzg_mi355x.hpp: compiles tests/cpp_mirror_test.cpp against the shared library; returns the executable's path."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "cpp_mirror_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(root, "include"), "-o", exe,
                           os.path.join(root, "tests", "cpp_mirror_test.cpp"), "-L" + os.path.join(root, "kzg_amd"),
                           "-lkzg_mi355x", "-Wl,-rpath," + os.path.join(root, "kzg_amd")])
    return exe
