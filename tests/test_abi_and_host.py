"""CPU-side checks of the product: the C-ABI library loads and exports every symbol the header declares,
the host-only helpers work, and there is no CPU fallback (context creation fails loudly without a GPU)."""
import ctypes
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "kzg_mi355x.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(kzg_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import kzg_amd
    lib = kzg_amd.load()
    names = header_functions()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/kzg_mi355x.h but not exported"
    # and the python binding table covers the whole header
    assert set(names) == set(lib._kzg_signatures.keys())


def test_compute_omega_host_helper():
    import kzg_amd
    from oracle import kzg_model as M
    for d in (1, 2, 3, 10, 1 << 10, (1 << 20) - 5, 1 << 20, 1 << 24):
        assert kzg_amd.compute_omega(d) == M.compute_omega(d)
    with pytest.raises(kzg_amd.PolynomialDegreeTooLarge):
        kzg_amd.compute_omega((1 << 31) + 1)   # exp >= Scalar::S (src/ft.rs:66-68)


def test_no_cpu_fallback():
    import torch
    import kzg_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(kzg_amd.EngineError):
        kzg_amd.Engine(0)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under kzg_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "kzg_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "kzg_oracle" not in txt, f
    so = os.path.join(ROOT, "kzg_amd", "libkzg_mi355x.so")
    out = os.popen(f"ldd {so}").read()
    assert "kzg_oracle" not in out


def test_bench_timed_region_does_not_import_oracle_or_torch():
    """bench.py's parts (tools/benchlib): only `checks` (the after-the-timer checkers) and `cpu_pool` (the CPU baseline's workers) may
    touch oracle/; the module that holds the timed region (headline), the other readings, the rank control plane and bench.py itself
    import neither the oracle nor torch; the control plane is standard-library sockets."""
    import ast
    lib_dir = os.path.join(ROOT, "tools", "benchlib")
    files = {f: os.path.join(lib_dir, f) for f in os.listdir(lib_dir) if f.endswith(".py")}
    files["bench.py"] = os.path.join(ROOT, "bench.py")
    assert {"headline.py", "paths.py", "control.py", "traffic.py", "sharded.py", "checks.py", "cpu_pool.py", "common.py"} <= set(files)
    for name, path in files.items():
        mods = set()
        for node in ast.walk(ast.parse(open(path).read())):
            if isinstance(node, ast.Import):
                mods |= {a.name.split(".")[0] for a in node.names}
            elif isinstance(node, ast.ImportFrom) and node.module and node.level == 0:
                mods.add(node.module.split(".")[0])
        assert "torch" not in mods, name
        if name not in ("checks.py", "cpu_pool.py"):
            assert "oracle" not in mods, name
    assert len(open(files["bench.py"]).read().splitlines()) <= 300


def test_bench_control_plane_star_of_three_ranks():
    """tools/benchlib/control.py: barrier, max-over-ranks, agreement, gather and broadcast over the TCP star, three ranks as threads."""
    import threading
    sys.path.insert(0, ROOT)
    from tools.benchlib.control import TcpStar
    port, world, out = 29911, 3, {}

    def rank_main(r):
        s = TcpStar(r, world, "127.0.0.1", port, timeout_s=30)
        got = [s.all_gather(None), max(s.all_gather(1.5 * (r + 1))), all(s.all_gather(r != 1)), s.all_gather({"rank": r, "blob": bytes([r]) * 200000}),
               s.all_gather(b"id-from-0" if r == 0 else None)[0]]
        s.close()
        out[r] = got
    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in reversed(th):      # the hub starts last: the others retry until it listens
        t.start()
    for t in th:
        t.join(60)
    for r in range(world):
        g = out[r]
        assert g[0] == [None] * 3 and g[1] == 4.5 and g[2] is False and g[4] == b"id-from-0"
        assert [d["rank"] for d in g[3]] == [0, 1, 2] and g[3][2]["blob"] == bytes([2]) * 200000


def test_polynomial_and_domain_host_mirror():
    import kzg_amd
    p = kzg_amd.Polynomial([3, 1] + [0] * 11)        # Polynomial::new trims the degree (src/polynomial.rs:83-105)
    assert p.degree == 1 and p.num_coeffs() == 2 and p.slice_coeffs() == [3, 1]
    assert kzg_amd.Polynomial([0, 0, 0]).degree == 0
    q = kzg_amd.Polynomial.new_from_coeffs([1, 2, 0, 0], 3)
    assert q.num_coeffs() == 4
    e = kzg_amd.EvaluationDomain.from_coeffs([1, 2, 3])   # zero-pads to 2^k (src/ft.rs:94-109)
    assert e.d == 4 and e.exp == 2 and len(e) == 4 and e.coeffs == [1, 2, 3, 0]
    assert kzg_amd.splitmix_scalar(1, 0) < kzg_amd.api.R_MODULUS


def test_host_only_helpers_shard_range_and_footprint():
    """kzg_shard_range / kzg_srs_footprint are pure host functions: usable (and tested) without a GPU."""
    import kzg_amd
    from kzg_amd.distributed import shard_range
    lib = kzg_amd.load()
    assert shard_range(1 << 24, 3, 8) == (3 << 21, 4 << 21)             # configs[4]: 2^21 terms per rank
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    lo, hi = ctypes.c_size_t(), ctypes.c_size_t()
    assert lib.kzg_shard_range(10, 4, 4, ctypes.byref(lo), ctypes.byref(hi)) == 3   # KZG_ERR_SHAPE: rank out of range
    nbytes = ctypes.c_size_t()
    assert lib.kzg_srs_footprint(1 << 20, 0, 0, ctypes.byref(nbytes)) == 0 and nbytes.value == (1 << 20) * (96 + 15 * 128)
    assert lib.kzg_srs_footprint(1 << 24, 0, 0, ctypes.byref(nbytes)) == 0 and nbytes.value == (1 << 24) * (96 + 13 * 128)   # 20-bit windows from 2^23 on
    assert lib.kzg_srs_footprint(1 << 24, 17, 0, ctypes.byref(nbytes)) == 0 and nbytes.value == (1 << 24) * (96 + 15 * 128)
    assert lib.kzg_srs_footprint(1 << 16, 0, 0, ctypes.byref(nbytes)) == 0 and nbytes.value == (1 << 16) * (96 + 20 * 128)  # c = 13
    assert lib.kzg_srs_footprint(1 << 20, 0, 3, ctypes.byref(nbytes)) == 0 and nbytes.value == (1 << 20) * (96 + 3 * 128)
    assert lib.kzg_srs_footprint(1 << 20, 99, 0, ctypes.byref(nbytes)) == 3
    # no GPU needed either: forming a group of zero devices, duplicate devices
    h = ctypes.c_void_p()
    arr = (ctypes.c_int * 2)(0, 0)
    assert lib.kzg_mctx_create(arr, 2, ctypes.byref(h)) == 3 and lib.kzg_mctx_create(arr, 0, ctypes.byref(h)) == 3


def test_single_node_rccl_env_sets_defaults_and_keeps_the_hosts_values(monkeypatch):
    """kzg_amd.distributed.single_node_rccl_env (what DeviceGroup calls before anything loads RCCL): the one-node knobs go into the
    process environment with setdefault semantics -- a value the host's operator exported is kept -- and KZG_RCCL_SINGLE_NODE_ENV=0
    switches the whole thing off.  (RCCL bootstrapping over a non-loopback interface cost the driver's box five minutes
    per communicator.)"""
    import os
    from kzg_amd.distributed import SINGLE_NODE_RCCL_ENV, single_node_rccl_env
    for k in SINGLE_NODE_RCCL_ENV:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.delenv("KZG_RCCL_SINGLE_NODE_ENV", raising=False)
    monkeypatch.setenv("NCCL_SOCKET_IFNAME", "eth7")
    done = single_node_rccl_env()
    assert os.environ["NCCL_SOCKET_IFNAME"] == "eth7" and "NCCL_SOCKET_IFNAME" not in done
    assert os.environ["NCCL_RAS_ENABLE"] == "0" and os.environ["NCCL_IB_DISABLE"] == "1" and os.environ["NCCL_NET_PLUGIN"] == "none"
    assert set(done) == set(SINGLE_NODE_RCCL_ENV) - {"NCCL_SOCKET_IFNAME"}
    for k in SINGLE_NODE_RCCL_ENV:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("KZG_RCCL_SINGLE_NODE_ENV", "0")
    assert single_node_rccl_env() == {} and "NCCL_RAS_ENABLE" not in os.environ
    # the C++ and Rust hosts set the same four variables
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hpp = open(os.path.join(root, "include", "kzg_mi355x.hpp")).read()
    rs = open(os.path.join(root, "integration", "mi355x.rs")).read()
    for k, v in SINGLE_NODE_RCCL_ENV.items():
        assert 'setenv("%s", "%s", 0)' % (k, v) in hpp and '("%s", "%s")' % (k, v) in rs, k


def test_bench_cpu_baseline_legs_on_a_small_sample():
    """tools/benchlib/cpu_pool.py without a GPU: the worker pool, the MSM legs at the requested sizes (both of the oracle's Pippengers for
    the small ones), the NTT and create_witness legs, the all-cores pass -- on a 2^10 sample whose "GPU side" is produced by the oracle's
    plain functions here, so every `matches_gpu` must come out true and a corrupted expected value must come out false."""
    import hashlib
    import random
    sys.path.insert(0, ROOT)
    from oracle import c_oracle as C
    from oracle import kzg_model as M
    from tools.benchlib.cpu_pool import CpuBaseline
    n, log_n, tau = 1 << 10, 10, 0x1234567
    rng = random.Random(9)
    coeffs = [rng.randrange(M.R) for _ in range(n)]
    sc = C.scalars_to_bytes(coeffs)
    pts = C.setup_g1(tau, n)
    x = rng.randrange(M.R)
    y = C.poly_eval(coeffs, x)
    G = C.g1_generator()
    ptau = C.poly_eval(coeffs, tau)
    gpu = {"msm": {10: C.msm_g1_raw(pts, sc, n), 6: C.msm_g1_raw(pts[:96 * 64], sc[:32 * 64], 64)},
           "ntt_sha256": hashlib.sha256(C.fft_bytes(sc, log_n)).hexdigest(),
           "witness": (x, y, C.g1_mul(G, (ptau - y) * pow(tau - x, -1, M.R) % M.R))}
    cpu = CpuBaseline()
    cpu.start()
    try:
        single, allc = cpu.run(pts, sc, n, log_n, gpu)
        assert single["kind"] == "port" and single["cores"] == 1 and single["unit"] == "commitments/s" and single["value"] > 0
        assert set(k for k in single if k.startswith(("msm_", "ntt_", "witness_"))) == {"msm_2e10", "msm_2e6", "ntt_2e10", "witness_2e10"}
        assert single["all_match_gpu"] is True and all(single[k]["matches_gpu"] for k in ("msm_2e10", "msm_2e6", "ntt_2e10", "witness_2e10"))
        assert single["msm_2e6"]["algorithm"].startswith("orc_msm_g1") and set(single["msm_2e6"]["seconds_by_algorithm"]) == {"orc_msm_g1_fast", "orc_msm_g1"}
        assert allc["cores"] >= 1 and allc["matches_gpu"] is True and allc["ntt_2e10"]["matches_gpu"] is True
        bad = dict(gpu, msm={10: gpu["msm"][10], 6: gpu["msm"][10]}, ntt_sha256="00" * 32)
        single, _ = cpu.run(pts, sc, n, log_n, bad)
        assert single["msm_2e6"]["matches_gpu"] is False and single["ntt_2e10"]["matches_gpu"] is False and single["all_match_gpu"] is False
    finally:
        cpu.close()
