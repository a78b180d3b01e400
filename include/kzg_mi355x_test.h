/*
 * kzg_mi355x_test.h -- unit-test hooks: device arithmetic exercised directly.  NOT part of the product ABI: they exist only in
 * the -DKZG_TEST_HOOKS build of the library (kzg_amd/libkzg_mi355x_hooks.so, built by `python -m kzg_amd.build` next to the
 * product library and loaded by tests/ only).
 *
 * The hooks build also honours two environment variables (read when the library first needs RCCL):
 *   KZG_TEST_NO_RCCL=1        RCCL is "not installed": the load-failure path on a host that has it
 *   KZG_TEST_SHM_TRANSPORT=1  the eight RCCL entry points are replaced by a shared-memory stand-in (kzg_amd/csrc/test_transport.h)
 *                             and kzg_mctx_create accepts the same device several times: device groups of world size > 1 on a
 *                             one-GPU box (RCCL refuses two ranks per GPU).  tests/test_gpu_mgpu_world.py, tools/bench_shared_gpu.sh.
 */
#ifndef KZG_MI355X_TEST_H
#define KZG_MI355X_TEST_H
#include "kzg_mi355x.h"
#ifdef __cplusplus
extern "C" {
#endif
int kzg_test_fr_mul(kzg_ctx *ctx, const void *a, const void *b, size_t n, void *out);   /* Montgomery */
int kzg_test_fq_mul(kzg_ctx *ctx, const void *a, const void *b, size_t n, void *out);   /* Montgomery, 48 B */
int kzg_test_fr_inv(kzg_ctx *ctx, const void *a, size_t n, void *out);
int kzg_test_g1_add(kzg_ctx *ctx, const void *a, const void *b, size_t n, void *out);   /* affine mont 96 */
int kzg_test_g1_mul(kzg_ctx *ctx, const void *p, const void *k_canonical, size_t n, void *out);
/* pretend `srs` is resident on GPU `device` (the "SRS of another GPU" error of every MSM entry point, on a one-GPU box) */
int kzg_test_srs_set_device(struct kzg_srs *srs, int device);
/* the next sharded call of this group fails locally on local GPU 0 with `code` (status agreement across ranks, mgpu.hip) */
int kzg_test_mctx_inject_failure(struct kzg_mctx *m, int code);
/* the next growth of the group's exchange buffers fails on local GPU 0 (a rank-local allocation failure BEFORE the exchange: the
 * ranks agree on it through the status-only all-gather instead of leaving the others inside the data all-gather) */
int kzg_test_mctx_inject_alloc_failure(struct kzg_mctx *m);
/* the next exchange sits behind a spin kernel of `ms` milliseconds on local GPU 0 (a peer that arrives late or never: option
 * "gather_timeout_ms" turns it into KZG_ERR_INTERNAL and a dead group) */
int kzg_test_mctx_inject_stall(struct kzg_mctx *m, int ms);
#ifdef __cplusplus
}
#endif
#endif
