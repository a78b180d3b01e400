/*
 * kzg_mi355x.h -- C ABI of the MI355X-native (gfx950) KZG commit/open engine.
 *
 * This is the drop-in boundary for the hot path of proxima-one/kzg (crate kzg 0.8.0-beta.1).  The
 * reference has no FFI of its own (it is pure Rust calling blstrs); the entry points below are the
 * ones a `kzg-mi355x-sys` binding would splice in at the reference's call sites, cited per function
 * as (reference file:line).  INTEGRATION.md shows the Rust-side binding.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ exceptions or callbacks cross the ABI.
 *   - every call returns a kzg_status; 0 = ok, >0 = the reference's own error conditions,
 *     <0 = runtime failure (HIP).  kzg_last_error(ctx) gives a human-readable string.
 *   - a kzg_ctx is bound to one GPU and is thread-safe.  The reference's blocking calls -- kzg_commit_coeff,
 *     kzg_commit_eval, kzg_msm_g1, kzg_witness_coeff, kzg_witness_coeff_batched, kzg_witness_eval, kzg_verify_poly_coeff /
 *     _eval, kzg_ntt_fr, kzg_coset_ntt_fr, kzg_poly_mul, kzg_poly_eval, kzg_quotient_linear / _eval, kzg_fr_vec_mul / _sub,
 *     kzg_divide_by_z_on_coset (KZGProver / KZGProverEvalForm / KZGVerifier are Clone + &self, src/coeff_form.rs:37-124; an
 *     EvaluationDomain is the caller's own) -- run CONCURRENTLY on one ctx: each call leases one of the context's lanes
 *     (option "streams", default and maximum 16), submits its kernels there and waits for its own result only, so N host
 *     threads calling commit() / create_witness_batched() against one resident SRS fill the GPU like kzg_msm_g1_batch does.
 *     Every other call (the *_batch / *_many forms, SRS construction, the pairing verifier, options, profiling) takes the
 *     context exclusively (it waits for the leased lanes to drain).  kzg_dev_alloc / _upload / _download take no lock at all and
 *     are NOT ordered against calls in flight on other threads: a download sees the results of every call that has RETURNED;
 *     reading a buffer another thread's call is still writing is the caller's race.  A kzg_srs / kzg_srs_g2 is immutable after creation
 *     and may be used from any number of threads and from every kzg_ctx on the same device.  kzg_last_error returns the
 *     calling thread's own last failure.  There is NO CPU fallback: without a usable HIP device kzg_ctx_create fails
 *     with KZG_ERR_NO_DEVICE.
 *   - points entering through kzg_srs_upload_g1/g2, the verifier entry points and kzg_pairing_check are validated the way
 *     blstrs' G1Affine / G2Affine deserialisation validates them upstream, in EVERY format: coordinates < q, on the curve,
 *     in the r-torsion subgroup ([r]P == O); failure -> KZG_ERR_BAD_POINT.  Option "trusted_points" = 1 skips the subgroup
 *     test (never the on-curve test).  kzg_g1_sum[_batch] checks coordinates and the curve equation only.
 *   - scalars in KZG_FR_CANONICAL_LE_32 that are >= r are taken mod r (single host scalars such as x, y, tau must be < r:
 *     KZG_ERR_SHAPE).
 *   - `flags` says where buffers live: scalars/points in host memory (default) or already resident
 *     in this GPU's HBM (KZG_IN_DEVICE), result written to host (default) or device (KZG_OUT_DEVICE).
 *
 * Data formats (all little-endian unless "ZCASH")
 *   KZG_FR_MONT_LE_32        4 x u64 limbs of a*2^256 mod r      (= blst_fr = blstrs::Scalar in memory)
 *   KZG_FR_CANONICAL_LE_32   32 bytes of a < r                    (= Scalar::to_bytes_le)
 *   KZG_G1_AFFINE_MONT_96    x,y: 6 x u64 Montgomery limbs each; identity = all zero (= blst_p1_affine)
 *   KZG_G1_JACOBIAN_MONT_144 X,Y,Z Montgomery limbs; identity has Z = 0 (= blst_p1 = G1Projective)
 *   KZG_G1_ZCASH_UNCOMPRESSED_96 / KZG_G1_ZCASH_COMPRESSED_48   big-endian canonical with flag bits
 *                            (bit7 compressed, bit6 infinity, bit5 y-sign) = G1Affine::to_uncompressed
 *                            / to_compressed
 *   KZG_G2_AFFINE_MONT_192   x.c0,x.c1,y.c0,y.c1 Montgomery limbs; identity all zero (= blst_p2_affine)
 *   KZG_G2_JACOBIAN_MONT_288 X,Y,Z in Fq2; identity has Z = 0 (= blst_p2 = G2Projective)
 *   KZG_G2_ZCASH_UNCOMPRESSED_192 / KZG_G2_ZCASH_COMPRESSED_96  x.c1 || x.c0 [|| y.c1 || y.c0] big-endian
 *                            with the same flag bits (= G2Affine::to_uncompressed / to_compressed)
 */
#ifndef KZG_MI355X_H
#define KZG_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Limits (each returns KZG_ERR_SHAPE with a message in kzg_last_error; tests/test_gpu_validation.py):
 *   kzg_ntt_fr                              log_n <= 28                      (2^24: two LDS passes of <= 2^12 points; above: one
 *                                                                             more four-step level of <= 16 around them)
 *   kzg_coset_ntt_fr                        log_n <= 28;  kzg_poly_mul: products of up to 2^27 coefficients;
 *   kzg_witness_coeff_batched               polynomials of up to 2^26 coefficients (the SRS with its window tables is 118 GB there);
 *   kzg_verify_poly_eval                    as far as the monomial SRS goes (MSM limit below)
 *   kzg_witness_coeff_batched,
 *   kzg_verify_eval_batched                 k <= 16384 opening points         (the interpolation works on a k x k matrix: 8.6 GB there)
 *   kzg_srs_lagrange_from_monomial_g1       d <= 2^24;  _g2: d <= 1024
 *   MSM                                     table rows x points < 2^31       (the sorted entry is a 31-bit table index + sign);
 *                                           window_bits 18, 19 (option), and 20 with option sort_single_pass: windows x points < 2^27
 *   kzg_g1_sum_batch                        count <= 2^20, groups <= 2^24
 *   kzg_commit_coeff_sharded_batch          batch <= 2^20
 * SRS footprint in HBM: points x (96 + rows x 128) bytes, rows = windows = ceil(256 / c) with c chosen from the size
 * (c = 20, 13 windows from 2^23 points on; c = 17, 15 windows from 2^17 on; 13 / 10 / 8 bits below: 1.97 GiB at 2^20, 27.5 GiB at 2^24;
 * kzg_srs_footprint computes it).  A host that
 * keeps many SRSs resident can trade speed for memory with option "window_rows" = r < windows: only r table rows are kept and
 * every MSM takes ceil(windows / r) passes over its scalars plus a doubling chain of c x r x (passes - 1) doublings. */
typedef struct kzg_ctx kzg_ctx;
typedef struct kzg_srs kzg_srs;
typedef struct kzg_srs_g2 kzg_srs_g2; /* the G2 half of KZGParams (hs) or a G2 Lagrange basis */

typedef enum {
    KZG_OK = 0,
    KZG_ERR_POINT_NOT_ON_POLY = 1,  /* KZGError::PointNotOnPolynomial      (src/lib.rs:30-31) */
    KZG_ERR_DEGREE_TOO_LARGE = 2,   /* KZGError::PolynomialDegreeTooLarge  (src/lib.rs:34-35) */
    KZG_ERR_SHAPE = 3,              /* a condition on which the reference panics (slice OOB,
                                       assert!(d == evals.d), index out of range, ...) */
    KZG_ERR_BAD_POINT = 4,          /* input point failed to decode / not on the curve */
    KZG_ERR_HIP = -1,
    KZG_ERR_NO_DEVICE = -2,
    KZG_ERR_ALLOC = -3,
    KZG_ERR_INTERNAL = -4
} kzg_status;

enum { KZG_FR_MONT_LE_32 = 0, KZG_FR_CANONICAL_LE_32 = 1 };
enum {
    KZG_G1_AFFINE_MONT_96 = 0,
    KZG_G1_JACOBIAN_MONT_144 = 1,
    KZG_G1_ZCASH_UNCOMPRESSED_96 = 2,
    KZG_G1_ZCASH_COMPRESSED_48 = 3
};
enum {
    KZG_G2_AFFINE_MONT_192 = 0,
    KZG_G2_JACOBIAN_MONT_288 = 1,
    KZG_G2_ZCASH_UNCOMPRESSED_192 = 2,
    KZG_G2_ZCASH_COMPRESSED_96 = 3
};
enum { KZG_IN_DEVICE = 1, KZG_OUT_DEVICE = 2 };

/* ---- context ------------------------------------------------------------------------------- */
const char *kzg_version(void);
/* Hardware queues: the pipelined paths (kzg_msm_g1_batch, concurrent blocking callers) want one HIP hardware queue per stream
 * (lanes + 2); the HIP runtime sizes its pool from GPU_MAX_HW_QUEUES (default 4) at the FIRST HIP call of the process.  The
 * library never changes the host's environment by itself.  A host that wants the full pipeline either exports
 * GPU_MAX_HW_QUEUES (>= 18) itself or calls kzg_init_hw_queues(n) (n = 0: 24) BEFORE its first HIP call -- it sets the
 * variable unless the host already did -- or starts with KZG_SET_HW_QUEUES=1 in the environment (the library's load-time
 * constructor then makes that call).  Otherwise the engine measures the queues it has and narrows the pipeline to fit
 * (4 queues: 3 lanes + 1 accumulation stream; loss: INTEGRATION.md section 6). */
int kzg_init_hw_queues(int queues);
int kzg_device_count(void);  /* usable HIP devices (0 if none) */
/* "hip=<file of the HIP runtime this library is bound to> runtime_version=<n> driver_version=<n>": a process may hold two HIP
 * runtimes (PyTorch wheels bundle their own); which one the library runs on follows from the host's load order. */
int kzg_runtime_info(char *buf, size_t buflen);
/* device: HIP device ordinal.  Fails with KZG_ERR_NO_DEVICE if no gfx950-class device is usable. */
int kzg_ctx_create(int device, kzg_ctx **out);
void kzg_ctx_destroy(kzg_ctx *ctx);
const char *kzg_last_error(kzg_ctx *ctx);
int kzg_sync(kzg_ctx *ctx);
/* tunables: "window_bits" (0 = auto, 4..20), "window_rows" (0 = one table row per window; applies to SRSs created afterwards),
 * "trusted_points" (0 / 1), "streams" (1..16: batch pipelining depth, default 13), "accum_streams" (0..4), "accum_blocks[_batch]",
 * "sort_threads[_batch]", "ntt_vec_log" / "ntt_vec2_log" (tile widths of the NTT passes), "ntt_kernel" (1 = default: tile load / store fused
 * into the first / last stage pair; 0 = the round-5 three-phase passes, 2 = two butterflies per thread: A/B only), "ntt_three_from" (sizes from
 * 2^this on take three passes of <= 2^8 points, default 23; 0 = never), "hw_queues" (0 = measure), "tail_quads" (0 / 1: latency-mode tail kernels of a
 * lone MSM), "host_affine" (1 / 0: a lone result bound for host memory is converted to affine and serialised by the calling
 * thread -- the same field code compiled for the host -- instead of one GPU lane; same bytes, ~90 us less latency),
 * "sort_single_pass" (0 / 1: 17-bit windows sorted in one pass instead of two levels; A/B only),
 * "defer_tail" (1 / 0: kzg_msm_g1_batch enqueues a lane's tail kernels after the sort of the lane's next MSM; default 1),
 * "accum_blocks_small" / "accum_streams_small" / "small_entries" (MSMs of at most small_entries sorted entries -- windows x terms,
 *  default 2^21: up to 2^17 points -- inside a pipeline take a 160-block accumulation grid and are spread over 4 accumulation
 *  streams instead of 480 blocks on 2: +20 % at 2^16, +50 % at 2^14; accum_blocks_small = 0 switches the rule off),
 * "heavy_bins" (sort bins far above their share -- scalars that are bits, bytes, all equal -- sorted in slices by many blocks:
 *  0 = for the 64 MSMs after one that met such a bin (default; the extra kernels cost uniform scalars 0.8 %), 1 = always, 2 = never;
 *  setting it clears the history),
 * "naf_window" (0 / 18, applies to SRSs created afterwards: 18 = positional tables, 2^j P for every bit position j = 255 rows of
 * 128 B per point, scalars recoded in width-18 non-adjacent form -- 13.9 instead of 15 bucket additions per scalar for 17x the
 * table: 34 GB at 2^20; measured +3.7 % batched throughput at 2^20, nothing below 2^19, +1.3 ms on a lone commit: opt-in);
 * unknown keys -> KZG_ERR_SHAPE */
int kzg_ctx_set_option(kzg_ctx *ctx, const char *key, int64_t value);
/* "device=<d> lanes=<n> accum_streams=<m> hw_queues_found=<q> narrowed_from=<L>+<A>|none witness_cache_slots=<s>": the plan of the
 * batched / concurrent-caller pipeline (zeros before the first such call) and whether the process' pool of hardware queues forced it
 * below what was asked for -- another context, an RCCL communicator or the host's own streams hold queues too, and a narrowed
 * pipeline loses 10-25 % of its batched rate.  The narrowing is also reported once per context on stderr. */
int kzg_ctx_info(kzg_ctx *ctx, char *buf, size_t buflen);

/* ---- SRS (KZGParams.gs, src/lib.rs:14-19; lagrange_basis_g, src/eval_form.rs:40-46) --------- */
/* Upload n G1 points once; they stay resident in HBM in the engine's internal layout
 * (affine Montgomery + per-window precomputed multiples).  Caller keeps ownership of `pts`. */
int kzg_srs_upload_g1(kzg_ctx *ctx, const void *pts, size_t n, int pfmt, kzg_srs **out);
/* setup(s, n), G1 half (src/lib.rs:38-47): gs[i] = [s^i]G, generated on the GPU. */
int kzg_srs_setup_g1(kzg_ctx *ctx, const void *s, int sfmt, size_t n, kzg_srs **out);
/* One contiguous shard of the same SRS: gs[first .. first+n) = [s^(first+i)]G (multi-GPU sharding). */
int kzg_srs_setup_g1_shard(kzg_ctx *ctx, const void *s, int sfmt, size_t first, size_t n, kzg_srs **out);
/* Lagrange-basis SRS for a known secret: L_i(s) G, i < d, d a power of two.  Same group elements
 * as compute_lagrange_basis(&setup(s, d)).0 (src/eval_form.rs:254-280), in O(d) not O(d^3). */
int kzg_srs_setup_lagrange_g1(kzg_ctx *ctx, const void *s, int sfmt, size_t d, kzg_srs **out);
/* compute_lagrange_basis (src/eval_form.rs:254-280), G1 half, from the monomial SRS alone (no secret): the inverse
 * group-NTT of gs, L_i = (1/d) sum_j w^(-ij) gs[j] (radix-2 over G1 points, one 255-bit scalar multiplication per
 * butterfly: ~0.7 s at d = 2^20).  d = len(gs) must be a power of two (KZG_ERR_SHAPE otherwise, the reference's assert)
 * and at most 2^24. */
int kzg_srs_lagrange_from_monomial_g1(kzg_ctx *ctx, const kzg_srs *monomial, kzg_srs **out);
int kzg_srs_download_g1(kzg_ctx *ctx, const kzg_srs *srs, size_t offset, size_t n, void *out, int pfmt);
size_t kzg_srs_len(const kzg_srs *srs);
/* HBM bytes an SRS of n points occupies under the given options (0 = engine defaults); host-only helper */
int kzg_srs_footprint(size_t n, int window_bits, int window_rows, size_t *bytes);
void kzg_srs_free(kzg_ctx *ctx, kzg_srs *srs);

/* ---- MSM: G1Projective::multi_exp(&gs[offset..offset+n], scalars) --------------------------- */
/* (call sites src/coeff_form.rs:61,78,102; src/eval_form.rs:118,136).  n may be 0 (-> identity). */
int kzg_msm_g1(kzg_ctx *ctx, const kzg_srs *srs, size_t offset, const void *scalars, size_t n, int sfmt,
               int flags, void *out, int ofmt);
/* `batch` scalar vectors (each n scalars, contiguous, stride n*32 B) against one SRS; out gets
 * `batch` points.  Throughput mode: independent MSMs are pipelined on several HIP streams, two in flight per lane (option
 * "defer_tail"): the context's workspace grows to 2 x `streams` MSM workspaces (2 x 16 x ~0.25 GB at 2^20, 17-bit windows). */
int kzg_msm_g1_batch(kzg_ctx *ctx, const kzg_srs *srs, size_t offset, const void *scalars, size_t n,
                     size_t batch, int sfmt, int flags, void *out, int ofmt);
/* sum of `count` G1 points (multi-GPU combine of per-rank partial MSMs). points in pfmt. */
int kzg_g1_sum(kzg_ctx *ctx, const void *points, size_t count, int pfmt, int flags, void *out, int ofmt);
/* `groups` independent sums: out[g] = sum_i points[g*count + i] (batched multi-GPU combine). */
int kzg_g1_sum_batch(kzg_ctx *ctx, const void *points, size_t count, size_t groups, int pfmt, int flags, void *out,
                     int ofmt);

/* ---- multi-GPU: the SRS sharded over up to 8 GPUs, partial commitments combined over RCCL / xGMI ---------------------
 * (north star: "MSM shards the SRS across up to 8 GPUs with an RCCL all-reduce of partial bucket sums"; the seam is the
 * multi_exp call of KZGProver::commit, src/coeff_form.rs:59-64, and of create_witness, :66-81.)
 * A kzg_mctx is a group of `world` GPUs (ranks).  Rank r holds the contiguous SRS range kzg_shard_range(n, r, world) resident
 * and reduces the matching scalar slice of every polynomial to ONE partial point on its GPU; the 144-byte Jacobian partials are
 * exchanged with one ncclAllGather and every rank adds them locally (EC addition is not an RCCL reduction op, so the
 * "all-reduce" is all-gather + local sum).  Two ways to form the group:
 *   - one host process driving n GPUs: kzg_mctx_create(devices, n)   (ncclCommInitAll; one host thread per device per call);
 *   - one process per GPU:  rank 0 calls kzg_mctx_unique_id, the host distributes the 128 bytes by any means (MPI, a TCP
 *     store, torch.distributed), every rank calls kzg_mctx_create_rank(device, rank, world, id)   (ncclCommInitRank).
 * RCCL (librccl.so.1) is loaded on first use; a group of one GPU needs no RCCL unless option "always_gather" is set.
 * Streams: all contexts of one device in a process -- plain ones and a group's -- take their streams from ONE pool (lane i of every
 * context is the same HIP stream), 13 lanes + 4 accumulation streams (+ one exchange stream per device group): the runtime multiplexes a process' streams onto one pool of
 * hardware queues (24 after kzg_init_hw_queues) of which an RCCL communicator needs about six, and streams that share a queue
 * serialise.  A plain prover context and a device group alive in one process therefore cost each other nothing (group path 471
 * against 473 commitments/s alone; both committing at once 512-517 in total).  kzg_ctx_info / kzg_mctx_info report each context's plan
 * and say so (also once on stderr) when the pool was short and a pipeline had to be narrowed.
 * Every entry point below is collective in the one-process-per-GPU mode: all ranks call it with the same arguments.
 * Failures stay collective too: a rank whose local phase fails still enters the exchange, its status travels with its
 * partials, and EVERY rank returns that error (no rank is left waiting inside the all-gather).  A rank-local resource failure
 * BEFORE the exchange (growing the exchange / quotient buffers: the only allocations a call makes, and only when a call is larger
 * than any the ranks have agreed on before) is agreed on through a status-only all-gather over buffers that exist since the group
 * was formed; after such a failure the failing call may simply be repeated.  What
 * cannot be agreed on -- a peer process that died, a hung GPU, a rank that could not create its communicator -- is bounded by a
 * deadline: the wait behind every exchange polls for at most "gather_timeout_ms" (kzg_mctx_set_option, default 60000, 0 = wait
 * for ever); when it expires the communicators are aborted (ncclCommAbort), the call returns KZG_ERR_INTERNAL and the group is
 * DEAD -- every later call on it fails at once with KZG_ERR_INTERNAL; destroy it and form a new one (kzg_mctx_destroy of a dead
 * group never synchronises on a stream an aborted collective still holds: such a context is left behind, with a line on stderr).
 * Communicator FORMATION is bounded the same way: ncclGetUniqueId / ncclCommInitRank / ncclCommInitAll / ncclCommDestroy run on
 * a helper thread and the caller waits at most "comm_timeout_ms" (kzg_mctx_set_option for a group that forms its communicator
 * lazily; the environment variable KZG_COMM_TIMEOUT_MS for the ones formed inside kzg_mctx_create*; default 60000, 0 = wait for
 * ever).  RCCL bootstraps over TCP on an interface of its own choosing even on one node, and a host whose interface swallows
 * packets stalls it for minutes.  On expiry the call returns KZG_ERR_INTERNAL with the phase timings in its message
 * (kzg_mctx_last_error, or kzg_mctx_create_error when the group was never handed out), the group is dead, and -- because the
 * abandoned call may hold RCCL's locks for ever -- no further communicator is formed in this process.  A one-node host should
 * export NCCL_SOCKET_IFNAME=lo, NCCL_RAS_ENABLE=0 and NCCL_IB_DISABLE=1 before the first RCCL call (INTEGRATION.md section 5b;
 * kzg_amd.api.DeviceGroup does).  KZG_DEBUG=1 prints every phase with its duration on stderr. */
typedef struct kzg_mctx kzg_mctx;
typedef struct kzg_msrs kzg_msrs;   /* an SRS sharded contiguously over the group */
enum { KZG_UNIQUE_ID_BYTES = 128 };
int kzg_mctx_create(const int *devices, int n, kzg_mctx **out);
int kzg_mctx_unique_id(void *id_out);
int kzg_mctx_create_rank(int device, int rank, int world, const void *unique_id, kzg_mctx **out);
void kzg_mctx_destroy(kzg_mctx *m);
const char *kzg_mctx_last_error(kzg_mctx *m);
/* why the calling thread's last kzg_mctx_create / kzg_mctx_create_rank / kzg_mctx_unique_id failed (no group exists to ask);
 * "" when it did not.  Valid until the thread's next call of one of the three. */
const char *kzg_mctx_create_error(void);
/* "rccl=<file> version=<n> hip=<runtime file> world=.. local=.. mode=.. formation_ms=<f> phases_ms=[load=.. uid=.. init=..
 * first_exchange=.. destroy=..] comm_timeout_ms=.. gather_timeout_ms=.. dead=0|1": which RCCL the group adopted (the copy the
 * process already holds, else librccl.so.1 from the library path), the HIP runtime it is bound to, and what forming the
 * communicator cost (wall-clock ms per phase: loading RCCL, ncclGetUniqueId, ncclCommInit*, the first exchange -- RCCL loads its
 * kernels there --; -1 = has not happened).  The library refuses an RCCL that
 * is bound to a different HIP runtime than itself (streams and device pointers cross the boundary): KZG_ERR_INTERNAL. */
int kzg_mctx_info(kzg_mctx *m, char *buf, size_t buflen);
int kzg_mctx_world(const kzg_mctx *m);        /* ranks in the group */
int kzg_mctx_local_count(const kzg_mctx *m);  /* GPUs this process drives (n, or 1 in the per-process mode) */
int kzg_mctx_rank(const kzg_mctx *m, int local_index);          /* global rank of a local GPU */
kzg_ctx *kzg_mctx_ctx(kzg_mctx *m, int local_index);            /* its single-GPU context (NTT, scans, device memory) */
/* options: "always_gather" (run the collective even in a group of one), "gather_timeout_ms", "comm_timeout_ms" (above), plus every
 * kzg_ctx_set_option key (applied to all) */
int kzg_mctx_set_option(kzg_mctx *m, const char *key, int64_t value);
/* rank r of `world` holds terms [lo, hi) of n: the first n % world ranks get one extra.  Host-only helper. */
int kzg_shard_range(size_t n, int rank, int world, size_t *lo, size_t *hi);
/* setup(s, n), G1 half, sharded: rank r generates gs[lo_r .. hi_r) on its GPU (src/lib.rs:38-47). */
int kzg_srs_setup_g1_sharded(kzg_mctx *m, const void *s, int sfmt, size_t n, kzg_msrs **out);
/* KZGParams.gs supplied by the caller: `pts` is the WHOLE vector (n points, host); every rank uploads its own range. */
int kzg_srs_upload_g1_sharded(kzg_mctx *m, const void *pts, size_t n, int pfmt, kzg_msrs **out);
size_t kzg_msrs_len(const kzg_msrs *srs);
const kzg_srs *kzg_msrs_shard(const kzg_msrs *srs, int local_index, size_t *first);  /* the resident shard and its first index */
void kzg_msrs_free(kzg_mctx *m, kzg_msrs *srs);
/* KZGProver::commit over the group (src/coeff_form.rs:59-64).  coeffs: the whole polynomial (n scalars, host), every rank
 * reads its slice; with KZG_IN_DEVICE `coeffs` is instead an array of kzg_mctx_local_count() device pointers, entry i
 * pointing at local GPU i's slice ([hi - lo] scalars, resident in that GPU's HBM).  out: one point (host), on every rank.
 * KZG_ERR_SHAPE if n > kzg_msrs_len (the slice index panic).  (This is the sharded multi_exp: with a Lagrange-basis SRS
 * uploaded through kzg_srs_upload_g1_sharded it is KZGProverEvalForm::commit, src/eval_form.rs:114-122, as well.) */
int kzg_commit_coeff_sharded(kzg_mctx *m, const kzg_msrs *srs, const void *coeffs, size_t n, int sfmt, int flags, void *out,
                             int ofmt);
/* `batch` polynomials of n coefficients each (host: contiguous, stride n * 32 B; KZG_IN_DEVICE: per local GPU a
 * [batch][hi - lo] array of slices).  One all-gather of batch x 144 B per rank for the whole batch.  out: batch points. */
int kzg_commit_coeff_sharded_batch(kzg_mctx *m, const kzg_msrs *srs, const void *coeffs, size_t n, size_t batch, int sfmt,
                                   int flags, void *out, int ofmt);
/* KZGProver::create_witness over the group (src/coeff_form.rs:66-81): every rank computes the quotient (p - y)/(X - x) on its
 * GPU (a replicated O(n) scan) and reduces its own slice of it.  coeffs: the whole polynomial in host memory, or with
 * KZG_IN_DEVICE an array of kzg_mctx_local_count() device pointers, entry i = the WHOLE polynomial resident on local GPU i.
 * KZG_ERR_POINT_NOT_ON_POLY iff p(x) != y. */
int kzg_witness_coeff_sharded(kzg_mctx *m, const kzg_msrs *srs, const void *coeffs, size_t n, const void *x, const void *y,
                              int sfmt, int flags, void *out, int ofmt);
/* KZGProver::create_witness_batched over the group (src/coeff_form.rs:83-111): interpolant and quotient (p - I)/Z replicated on
 * every GPU (NTTs do not shard), the quotient MSM sharded like the commit.  Arguments and errors as kzg_witness_coeff_batched;
 * coeffs as for kzg_witness_coeff_sharded.  out_w: one point (host), out_r: the interpolant, on every rank. */
int kzg_witness_coeff_batched_sharded(kzg_mctx *m, const kzg_msrs *srs, const void *coeffs, size_t n, const void *xs,
                                      const void *ys, size_t k, int sfmt, int flags, void *out_w, int ofmt, void *out_r,
                                      size_t *out_r_len);

/* KZGProverEvalForm::create_witness over the group (src/eval_form.rs:124-140): `lagrange` is the Lagrange-basis SRS sharded over
 * the group (kzg_srs_upload_g1_sharded); every rank computes div_by_omega_i of (evals - evals[index]) on its GPU (replicated O(d)
 * pass) and reduces its slice of the quotient.  evals: d scalars in host memory, or with KZG_IN_DEVICE an array of
 * kzg_mctx_local_count() device pointers (the whole vector resident on each local GPU).  KZG_ERR_SHAPE if index >= d, d is not a
 * power of two or d > kzg_msrs_len (the reference's panics). */
int kzg_witness_eval_sharded(kzg_mctx *m, const kzg_msrs *lagrange, const void *evals, size_t d, size_t index, int sfmt, int flags,
                             void *out, int ofmt);

/* ---- NTT: EvaluationDomain::fft / ifft (src/ft.rs:111-140; best_fft :274-288) --------------- */
/* EvaluationDomain::compute_omega (src/ft.rs:55-76): m = next pow2 >= d, exp = log2 m, omega.
 * Returns KZG_ERR_DEGREE_TOO_LARGE if exp >= 32.  Host-only helper; omega written in sfmt. */
int kzg_compute_omega(size_t d, size_t *m, uint32_t *exp, void *omega, int sfmt);
/* In place, natural order in and out: out[i] = sum_j a[j] w^(ij); inverse uses w^-1 and scales by
 * d^-1.  The transform is linear, so data may be in either Fr format (it is preserved). */
int kzg_ntt_fr(kzg_ctx *ctx, void *data, uint32_t log_n, int inverse, int flags);
/* coset_fft / icoset_fft (src/ft.rs:168-178): distribute_powers(g = 7) then fft; ifft then g^-i. */
int kzg_coset_ntt_fr(kzg_ctx *ctx, void *data, uint32_t log_n, int inverse, int sfmt, int flags);

/* EvaluationDomain::z (src/ft.rs:182-187): tau^d - 1.  Host-only helper. */
int kzg_domain_z(size_t d, const void *tau, int sfmt, void *out);
/* EvaluationDomain::divide_by_z_on_coset (src/ft.rs:192-217): data[i] *= 1 / (7^d - 1), d = 2^log_n. */
int kzg_divide_by_z_on_coset(kzg_ctx *ctx, void *data, uint32_t log_n, int sfmt, int flags);
/* EvaluationDomain::mul_assign / sub_assign (src/ft.rs:220-271): a[i] *= b[i] / a[i] -= b[i], n elements each
 * (KZG_IN_DEVICE: both resident). */
int kzg_fr_vec_mul(kzg_ctx *ctx, void *a, const void *b, size_t n, int sfmt, int flags);
int kzg_fr_vec_sub(kzg_ctx *ctx, void *a, const void *b, size_t n, int sfmt, int flags);

/* ---- coefficient form (src/coeff_form.rs) ---------------------------------------------------- */
/* KZGProver::commit (:59-64).  coeffs[0..n) = polynomial.slice_coeffs(); KZG_ERR_SHAPE if n > SRS. */
int kzg_commit_coeff(kzg_ctx *ctx, const kzg_srs *srs, const void *coeffs, size_t n, int sfmt, int flags,
                     void *out, int ofmt);
/* KZGProver::create_witness (:66-81): [(p - y)/(X - x)]_1; KZG_ERR_POINT_NOT_ON_POLY iff p(x) != y.
 * x, y are host scalars in sfmt.  n = polynomial.num_coeffs() >= 1. */
int kzg_witness_coeff(kzg_ctx *ctx, const kzg_srs *srs, const void *coeffs, size_t n, const void *x,
                      const void *y, int sfmt, int flags, void *out, int ofmt);
/* Throughput form of the above ("256 independent create_witness calls sharing one SRS"): `count`
 * openings (xs[j], ys[j]) of ONE polynomial, pipelined like kzg_msm_g1_batch -- quotient scan and MSM of opening j on lane
 * j % streams.  out: count points in ofmt.  status (optional, count ints): 0 or KZG_ERR_POINT_NOT_ON_POLY per opening, the
 * call then returns KZG_OK; without it the call returns KZG_ERR_POINT_NOT_ON_POLY if any opening is off the polynomial (every
 * witness is still written).  Not a reference method: equivalent to calling create_witness count times. */
int kzg_witness_coeff_many(kzg_ctx *ctx, const kzg_srs *srs, const void *coeffs, size_t n, const void *xs, const void *ys,
                           size_t count, int sfmt, int flags, void *out, int ofmt, int *status);
/* KZGProver::create_witness_batched (:83-111): w = [(p - I)/Z]_1 and r = I (the interpolant through
 * (xs, ys)).  xs, ys: k host scalars.  out_r receives *out_r_len scalars (host, sfmt): k normally,
 * 2 for k == 1 (the reference returns X + (y - x), src/polynomial.rs:244-247).
 * KZG_ERR_POINT_NOT_ON_POLY iff some p(xs[i]) != ys[i].
 * Opening-point sets are remembered per context: what depends on xs alone (Z, the barycentric weights, the coset shift, 1 / Z on the
 * coset: N x 32 bytes) is kept in one of "witness_cache_slots" equal slots (kzg_ctx_set_option, default 16, 0 = off; the pool is
 * allocated by the first call, sized for that call), so that opening many polynomials at ONE point set -- the way batched openings are
 * used -- pays for the point set once.  Results are identical with and without the cache.  kzg_prof_get(ctx, "point_set_cache",
 * &hits, &misses_as_double) reports its use. */
int kzg_witness_coeff_batched(kzg_ctx *ctx, const kzg_srs *srs, const void *coeffs, size_t n,
                              const void *xs, const void *ys, size_t k, int sfmt, int flags, void *out_w,
                              int ofmt, void *out_r, size_t *out_r_len);
/* KZGVerifier::verify_poly (:119-124): *ok = (commit(coeffs) == commitment). */
int kzg_verify_poly_coeff(kzg_ctx *ctx, const kzg_srs *srs, const void *commitment, int pfmt,
                          const void *coeffs, size_t n, int sfmt, int flags, int *ok);

/* ---- evaluation form (src/eval_form.rs) ------------------------------------------------------ */
/* KZGProverEvalForm::commit (:114-122): MSM against the Lagrange SRS.  KZG_ERR_SHAPE unless
 * d == kzg_srs_len(lagrange) (assert!(self.d == evals.d)). */
int kzg_commit_eval(kzg_ctx *ctx, const kzg_srs *lagrange, const void *evals, size_t d, int sfmt, int flags,
                    void *out, int ofmt);
/* KZGProverEvalForm::create_witness (:124-140) incl. div_by_omega_i (:58-84).  KZG_ERR_SHAPE if
 * i >= d (the reference panics on the index). */
int kzg_witness_eval(kzg_ctx *ctx, const kzg_srs *lagrange, const void *evals, size_t d, size_t i, int sfmt,
                     int flags, void *out, int ofmt);
/* Throughput form: `count` openings (indices[j]) of ONE evaluation vector, pipelined like kzg_msm_g1_batch.  out: count points.
 * Not a reference method: equivalent to calling KZGProverEvalForm::create_witness count times. */
int kzg_witness_eval_many(kzg_ctx *ctx, const kzg_srs *lagrange, const void *evals, size_t d, const size_t *indices,
                          size_t count, int sfmt, int flags, void *out, int ofmt);
/* KZGVerifierEvalForm::verify_poly (:162-171): ifft then monomial MSM, compare. */
int kzg_verify_poly_eval(kzg_ctx *ctx, const kzg_srs *monomial, const void *commitment, int pfmt,
                         const void *evals, size_t d, int sfmt, int flags, int *ok);

/* ---- verifier: G2 parameters and pairing checks (src/coeff_form.rs:126-182, src/eval_form.rs:173-217) ----
 * The reference runs these on the CPU through blstrs' pairing(); here one GPU thread evaluates one check
 * (shared Miller loop over both pairs + one final exponentiation), so `count` openings verify in one launch.
 * Host-resident inputs only; commitments / witnesses in an affine G1 format (KZGCommitment = G1Affine). */
/* setup(), G2 half (src/lib.rs:48-52): hs[i] = [s^i] H for i < n. */
int kzg_srs_setup_g2(kzg_ctx *ctx, const void *s, int sfmt, size_t n, kzg_srs_g2 **out);
/* [L_i(s)] H over the size-d domain (lagrange_basis_h of KZGVerifierEvalForm::new, src/eval_form.rs:150). */
int kzg_srs_setup_lagrange_g2(kzg_ctx *ctx, const void *s, int sfmt, size_t d, kzg_srs_g2 **out);
/* compute_lagrange_basis, G2 half, from hs alone (src/eval_form.rs:254-280); d = len(hs) a power of two <= 1024. */
int kzg_srs_lagrange_from_monomial_g2(kzg_ctx *ctx, const kzg_srs_g2 *hs, kzg_srs_g2 **out);
/* KZGParams.hs supplied by the caller (Vec<G2Projective> = KZG_G2_JACOBIAN_MONT_288, or any G2 format);
 * KZG_ERR_BAD_POINT if a point does not decode / is not on the twist. */
int kzg_srs_upload_g2(kzg_ctx *ctx, const void *pts, size_t n, int pfmt, kzg_srs_g2 **out);
int kzg_srs_download_g2(kzg_ctx *ctx, const kzg_srs_g2 *srs, size_t offset, size_t n, void *out, int pfmt);
size_t kzg_srs_g2_len(const kzg_srs_g2 *srs);
void kzg_srs_g2_free(kzg_ctx *ctx, kzg_srs_g2 *srs);
/* G2Projective::multi_exp (call site src/coeff_form.rs:156): sum scalars[i] * srs[offset+i]; meant for the
 * verifier's small n (one thread per term). */
int kzg_msm_g2(kzg_ctx *ctx, const kzg_srs_g2 *srs, size_t offset, const void *scalars, size_t n, int sfmt, void *out,
               int ofmt);
/* ok[c] = (prod_{i < pairs_per_check} e(g1[c*ppc + i], g2[c*ppc + i]) == 1), 1 <= pairs_per_check <= 4. */
int kzg_pairing_check(kzg_ctx *ctx, const void *g1_points, int pfmt1, const void *g2_points, int pfmt2,
                      size_t pairs_per_check, size_t checks, uint8_t *ok);
/* KZGVerifier::verify_eval (src/coeff_form.rs:126-142) for `count` independent (x, y, commitment, witness)
 * tuples: ok[c] = e(w_c, hs[1] - [x_c]hs[0]) == e(C_c - [y_c]gs[0], hs[0]).  KZGVerifierEvalForm::verify_eval
 * (src/eval_form.rs:173-190) is the same check at x = omega^i.  KZG_ERR_SHAPE if hs has fewer than 2 points. */
int kzg_verify_eval(kzg_ctx *ctx, const kzg_srs *gs, const kzg_srs_g2 *hs, const void *xs, const void *ys, int sfmt,
                    const void *commitments, const void *witnesses, int pfmt, size_t count, uint8_t *ok);
/* KZGVerifier::verify_eval_batched (src/coeff_form.rs:144-182): z = prod (X - xs[i]), hz = MSM(hs, z),
 * gr = MSM(gs, r) with r = witness.r (r_len = num_coeffs), *ok = e(w, hz) == e(C - gr, hs[0]).
 * KZG_ERR_SHAPE if k + 1 > len(hs) or r_len > len(gs) (slice index panics), k = 0, or k > 16384. */
int kzg_verify_eval_batched(kzg_ctx *ctx, const kzg_srs *gs, const kzg_srs_g2 *hs, const void *xs, size_t k,
                            const void *r_coeffs, size_t r_len, int sfmt, const void *commitment, const void *witness,
                            int pfmt, int *ok);
/* KZGVerifierEvalForm::verify_eval_all (src/eval_form.rs:192-217) exactly as written there. */
int kzg_verify_eval_all(kzg_ctx *ctx, const kzg_srs *lagrange_g, const kzg_srs_g2 *lagrange_h, const kzg_srs_g2 *hs,
                        const void *ys, size_t ys_len, int sfmt, const void *commitment, const void *witness, int pfmt,
                        int *ok);

/* ---- Fr polynomial helpers on the path (device) ---------------------------------------------- */
/* Polynomial::eval (src/polynomial.rs:156-165) at one point. */
int kzg_poly_eval(kzg_ctx *ctx, const void *coeffs, size_t n, const void *x, int sfmt, int flags, void *y_out);
/* quotient of create_witness without the MSM: q = (p - y)/(X - x), n-1 scalars, in place allowed.
 * Returns KZG_ERR_POINT_NOT_ON_POLY if the remainder is non-zero. */
int kzg_quotient_linear(kzg_ctx *ctx, const void *coeffs, size_t n, const void *x, const void *y, int sfmt,
                        int flags, void *q_out);
/* div_by_omega_i (src/eval_form.rs:58-84) applied to (evals - evals[i]). */
int kzg_quotient_eval(kzg_ctx *ctx, const void *evals, size_t d, size_t i, int sfmt, int flags, void *q_out);

/* Polynomial::fft_mul / best_mul (src/polynomial.rs:167-191): out = a * b, na + nb - 1 coefficients, by three
 * NTTs of size next_pow2(na + nb) and a pointwise product (EvaluationDomain::mul_assign, src/ft.rs:220-244).
 * The product is unique, so it equals the reference's naive Mul (:473-487) as well.  Host or device buffers. */
int kzg_poly_mul(kzg_ctx *ctx, const void *a, size_t na, const void *b, size_t nb, int sfmt, int flags, void *out);

/* ---- device memory + measurement ------------------------------------------------------------- */
int kzg_dev_alloc(kzg_ctx *ctx, size_t bytes, void **out);
int kzg_dev_free(kzg_ctx *ctx, void *p);
int kzg_dev_upload(kzg_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int kzg_dev_download(kzg_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
/* Fill n Fr elements with a counter-based generator (SplitMix64 of (seed, i), reduced mod r;
 * `u64_valued` != 0 mirrors the reference benches' u64-valued coefficients,
 * benches/commit_coeff_form.rs:16-21).  Output canonical or Montgomery per sfmt. */
int kzg_fill_random_fr(kzg_ctx *ctx, void *dst_dev, size_t n, uint64_t seed, int u64_valued, int sfmt);
/* Per-kernel HIP-event timing on the stream the kernels run on (for bench.py's roofline line).  on = 1: every kernel; on = 2: the
 * bucket-accumulation kernel only (two events per launch of all ~14 kernels of an MSM cost 1.4 % of the batched rate, one pair per
 * MSM does not); 0: off. */
int kzg_prof_enable(kzg_ctx *ctx, int on);
int kzg_prof_reset(kzg_ctx *ctx);
int kzg_prof_get(kzg_ctx *ctx, const char *kernel, uint64_t *launches, double *total_ms);
/* comma-separated list of kernel names seen so far */
int kzg_prof_names(kzg_ctx *ctx, char *buf, size_t buflen);
/* The v_mad_i64_i32 issue rate of THIS device, measured now (~30 ms at 8 waves per SIMD: 8 independent accumulator chains per
 * lane), in 10^12 lane-multiply-adds per second: the roofline peak bench.py prices the bucket-accumulation kernel against
 * (devices of one pool differ by several percent, and the sustained clock under this load is not the nominal one). */
int kzg_measure_mad_issue_rate(kzg_ctx *ctx, int waves_per_simd, double *tera_lane_mads_per_s);
/* number of window bits and windows the engine chose for this SRS (for G1-adds accounting) */
int kzg_srs_window_info(const kzg_srs *srs, int *window_bits, int *windows);
int kzg_srs_table_rows(const kzg_srs *srs);   /* resident table rows (= windows unless option window_rows) */

#ifdef __cplusplus
}
#endif
#endif /* KZG_MI355X_H */
