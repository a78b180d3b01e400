"""ctypes loader for libkzg_mi355x.so.  There is no CPU fallback: if the library is missing or no
MI355X-class device is usable the import / context creation raises."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(HERE, "libkzg_mi355x.so")

KZG_OK = 0
KZG_ERR_POINT_NOT_ON_POLY = 1
KZG_ERR_DEGREE_TOO_LARGE = 2
KZG_ERR_SHAPE = 3
KZG_ERR_BAD_POINT = 4
KZG_ERR_HIP = -1
KZG_ERR_NO_DEVICE = -2
KZG_ERR_ALLOC = -3
KZG_ERR_INTERNAL = -4

FR_MONT = 0
FR_CANONICAL = 1
G1_AFFINE_MONT = 0
G1_JACOBIAN_MONT = 1
G1_ZCASH_UNCOMPRESSED = 2
G1_ZCASH_COMPRESSED = 3
POINT_BYTES = {0: 96, 1: 144, 2: 96, 3: 48}
G2_AFFINE_MONT, G2_JACOBIAN_MONT, G2_UNCOMPRESSED, G2_COMPRESSED = 0, 1, 2, 3
G2_POINT_BYTES = {0: 192, 1: 288, 2: 192, 3: 96}
IN_DEVICE = 1
OUT_DEVICE = 2

_lib = None

c_void_pp = ctypes.POINTER(ctypes.c_void_p)


def load(path=None):
    """Load the shared library.  `path` (tools/ab_libs.py only) or the environment variable KZG_AMD_LIBRARY (tests: the hooks
    build under torch.distributed.run) selects another build of the same ABI."""
    global _lib
    if _lib is not None:
        return _lib
    so = path or os.environ.get("KZG_AMD_LIBRARY") or SO_PATH
    if not os.path.exists(so):
        raise ImportError(
            f"{so} not found: build it with `python -m kzg_amd.build` (hipcc --offload-arch=gfx950). "
            "kzg_amd has no CPU fallback.")
    L = ctypes.CDLL(so)
    sz, i32, u32, u64, vp = ctypes.c_size_t, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_void_p
    sig = {
        "kzg_version": (ctypes.c_char_p, []),
        "kzg_init_hw_queues": (i32, [i32]),
        "kzg_device_count": (i32, []),
        "kzg_runtime_info": (i32, [ctypes.c_char_p, sz]),
        "kzg_ctx_create": (i32, [i32, c_void_pp]),
        "kzg_ctx_destroy": (None, [vp]),
        "kzg_last_error": (ctypes.c_char_p, [vp]),
        "kzg_sync": (i32, [vp]),
        "kzg_ctx_set_option": (i32, [vp, ctypes.c_char_p, ctypes.c_int64]),
        "kzg_ctx_info": (i32, [vp, ctypes.c_char_p, sz]),
        "kzg_srs_upload_g1": (i32, [vp, vp, sz, i32, c_void_pp]),
        "kzg_srs_setup_g1": (i32, [vp, vp, i32, sz, c_void_pp]),
        "kzg_srs_setup_g1_shard": (i32, [vp, vp, i32, sz, sz, c_void_pp]),
        "kzg_srs_setup_lagrange_g1": (i32, [vp, vp, i32, sz, c_void_pp]),
        "kzg_srs_lagrange_from_monomial_g1": (i32, [vp, vp, c_void_pp]),
        "kzg_srs_download_g1": (i32, [vp, vp, sz, sz, vp, i32]),
        "kzg_srs_len": (sz, [vp]),
        "kzg_srs_footprint": (i32, [sz, i32, i32, ctypes.POINTER(sz)]),
        "kzg_srs_table_rows": (i32, [vp]),
        "kzg_srs_free": (None, [vp, vp]),
        "kzg_srs_window_info": (i32, [vp, ctypes.POINTER(i32), ctypes.POINTER(i32)]),
        "kzg_msm_g1": (i32, [vp, vp, sz, vp, sz, i32, i32, vp, i32]),
        "kzg_msm_g1_batch": (i32, [vp, vp, sz, vp, sz, sz, i32, i32, vp, i32]),
        "kzg_g1_sum": (i32, [vp, vp, sz, i32, i32, vp, i32]),
        "kzg_g1_sum_batch": (i32, [vp, vp, sz, sz, i32, i32, vp, i32]),
        "kzg_mctx_create": (i32, [ctypes.POINTER(i32), i32, c_void_pp]),
        "kzg_mctx_unique_id": (i32, [vp]),
        "kzg_mctx_create_rank": (i32, [i32, i32, i32, vp, c_void_pp]),
        "kzg_mctx_destroy": (None, [vp]),
        "kzg_mctx_last_error": (ctypes.c_char_p, [vp]),
        "kzg_mctx_create_error": (ctypes.c_char_p, []),
        "kzg_mctx_world": (i32, [vp]),
        "kzg_mctx_local_count": (i32, [vp]),
        "kzg_mctx_rank": (i32, [vp, i32]),
        "kzg_mctx_ctx": (vp, [vp, i32]),
        "kzg_mctx_set_option": (i32, [vp, ctypes.c_char_p, ctypes.c_int64]),
        "kzg_shard_range": (i32, [sz, i32, i32, ctypes.POINTER(sz), ctypes.POINTER(sz)]),
        "kzg_srs_setup_g1_sharded": (i32, [vp, vp, i32, sz, c_void_pp]),
        "kzg_srs_upload_g1_sharded": (i32, [vp, vp, sz, i32, c_void_pp]),
        "kzg_msrs_len": (sz, [vp]),
        "kzg_msrs_shard": (vp, [vp, i32, ctypes.POINTER(sz)]),
        "kzg_msrs_free": (None, [vp, vp]),
        "kzg_commit_coeff_sharded": (i32, [vp, vp, vp, sz, i32, i32, vp, i32]),
        "kzg_commit_coeff_sharded_batch": (i32, [vp, vp, vp, sz, sz, i32, i32, vp, i32]),
        "kzg_witness_coeff_sharded": (i32, [vp, vp, vp, sz, vp, vp, i32, i32, vp, i32]),
        "kzg_witness_coeff_batched_sharded": (i32, [vp, vp, vp, sz, vp, vp, sz, i32, i32, vp, i32, vp, ctypes.POINTER(sz)]),
        "kzg_witness_eval_sharded": (i32, [vp, vp, vp, sz, sz, i32, i32, vp, i32]),
        "kzg_mctx_info": (i32, [vp, ctypes.c_char_p, sz]),
        "kzg_compute_omega": (i32, [sz, ctypes.POINTER(sz), ctypes.POINTER(u32), vp, i32]),
        "kzg_ntt_fr": (i32, [vp, vp, u32, i32, i32]),
        "kzg_coset_ntt_fr": (i32, [vp, vp, u32, i32, i32, i32]),
        "kzg_domain_z": (i32, [sz, vp, i32, vp]),
        "kzg_divide_by_z_on_coset": (i32, [vp, vp, u32, i32, i32]),
        "kzg_fr_vec_mul": (i32, [vp, vp, vp, sz, i32, i32]),
        "kzg_fr_vec_sub": (i32, [vp, vp, vp, sz, i32, i32]),
        "kzg_commit_coeff": (i32, [vp, vp, vp, sz, i32, i32, vp, i32]),
        "kzg_witness_coeff": (i32, [vp, vp, vp, sz, vp, vp, i32, i32, vp, i32]),
        "kzg_witness_coeff_many": (i32, [vp, vp, vp, sz, vp, vp, sz, i32, i32, vp, i32, ctypes.POINTER(i32)]),
        "kzg_witness_coeff_batched": (i32, [vp, vp, vp, sz, vp, vp, sz, i32, i32, vp, i32, vp, ctypes.POINTER(sz)]),
        "kzg_verify_poly_coeff": (i32, [vp, vp, vp, i32, vp, sz, i32, i32, ctypes.POINTER(i32)]),
        "kzg_commit_eval": (i32, [vp, vp, vp, sz, i32, i32, vp, i32]),
        "kzg_witness_eval_many": (i32, [vp, vp, vp, sz, ctypes.POINTER(sz), sz, i32, i32, vp, i32]),
        "kzg_witness_eval": (i32, [vp, vp, vp, sz, sz, i32, i32, vp, i32]),
        "kzg_verify_poly_eval": (i32, [vp, vp, vp, i32, vp, sz, i32, i32, ctypes.POINTER(i32)]),
        "kzg_srs_setup_g2": (i32, [vp, vp, i32, sz, c_void_pp]),
        "kzg_srs_setup_lagrange_g2": (i32, [vp, vp, i32, sz, c_void_pp]),
        "kzg_srs_lagrange_from_monomial_g2": (i32, [vp, vp, c_void_pp]),
        "kzg_srs_upload_g2": (i32, [vp, vp, sz, i32, c_void_pp]),
        "kzg_srs_download_g2": (i32, [vp, vp, sz, sz, vp, i32]),
        "kzg_srs_g2_len": (sz, [vp]),
        "kzg_srs_g2_free": (None, [vp, vp]),
        "kzg_msm_g2": (i32, [vp, vp, sz, vp, sz, i32, vp, i32]),
        "kzg_pairing_check": (i32, [vp, vp, i32, vp, i32, sz, sz, vp]),
        "kzg_verify_eval": (i32, [vp, vp, vp, vp, vp, i32, vp, vp, i32, sz, vp]),
        "kzg_verify_eval_batched": (i32, [vp, vp, vp, vp, sz, vp, sz, i32, vp, vp, i32, ctypes.POINTER(i32)]),
        "kzg_verify_eval_all": (i32, [vp, vp, vp, vp, vp, sz, i32, vp, vp, i32, ctypes.POINTER(i32)]),
        "kzg_poly_eval": (i32, [vp, vp, sz, vp, i32, i32, vp]),
        "kzg_quotient_linear": (i32, [vp, vp, sz, vp, vp, i32, i32, vp]),
        "kzg_quotient_eval": (i32, [vp, vp, sz, sz, i32, i32, vp]),
        "kzg_poly_mul": (i32, [vp, vp, sz, vp, sz, i32, i32, vp]),
        "kzg_dev_alloc": (i32, [vp, sz, c_void_pp]),
        "kzg_dev_free": (i32, [vp, vp]),
        "kzg_dev_upload": (i32, [vp, vp, vp, sz]),
        "kzg_dev_download": (i32, [vp, vp, vp, sz]),
        "kzg_fill_random_fr": (i32, [vp, vp, sz, u64, i32, i32]),
        "kzg_measure_mad_issue_rate": (i32, [vp, i32, ctypes.POINTER(ctypes.c_double)]),
        "kzg_prof_enable": (i32, [vp, i32]),
        "kzg_prof_reset": (i32, [vp]),
        "kzg_prof_get": (i32, [vp, ctypes.c_char_p, ctypes.POINTER(u64), ctypes.POINTER(ctypes.c_double)]),
        "kzg_prof_names": (i32, [vp, ctypes.c_char_p, sz]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)  # raises AttributeError if the export is missing
        f.restype = res
        f.argtypes = args
    L._kzg_signatures = sig
    # This module is the host: like a Rust host following INTEGRATION.md it asks for the hardware queues of the pipelined
    # paths before the first HIP call of the process (the library itself never changes the environment).  KZG_HW_QUEUES=0
    # leaves the runtime's default pool (the engine then narrows its pipeline to the queues it measures).
    if os.environ.get("KZG_HW_QUEUES", "1") != "0":
        L.kzg_init_hw_queues(0)
    _lib = L
    return L
