// gfft.hip -- compute_lagrange_basis (src/eval_form.rs:254-280), G1 half, from the monomial SRS alone (no secret):
// lagrange_basis_g[i] = commit(l_i) with l_i(X) = (1/d) sum_j w^(-ij) X^j, i.e. the INVERSE DFT of gs over the group:
//     L = iNTT_G1(gs),   L_i = (1/d) sum_j w^(-ij) gs[j].
// The reference builds every l_i by d - 1 polynomial multiplications and commits to it (O(d^3) field work, infeasible beyond a
// few hundred points); a real ceremony SRS has no tau, so the eval-form path at 2^20 needs this transform.
//
// Radix-2 decimation in frequency over G1 points held in the MSM's XYZZ / 30-bit form, natural order in, bit-reversed out (the
// permutation is folded into the final conversion).  A butterfly is (x, y) -> (x + y, [w^e](x - y)); the twiddle is a full
// 255-bit scalar and the point is variable, so each butterfly costs one variable-base scalar multiplication: fixed 4-bit
// windows (uniform control flow across a wave; a NAF's sparsity is lost to divergence), the 15 multiples of x - y in a
// per-thread slot of an HBM scratch table.  The last stage has trivial twiddles and carries the 1/d scaling instead.
// Work: (d/2)(log d - 1) + d scalar multiplications of ~240 additions each: ~0.7 s at d = 2^20.  Untimed input generation.
#include "common.h"

namespace kzg {

constexpr int GF_W = 4;                     // window bits
constexpr int GF_TAB = (1 << GF_W) - 1;     // multiples 1 .. 15
constexpr size_t GF_CHUNK = (size_t)1 << 19;  // butterflies per launch (bounds the scratch table: 2^19 x 15 x 224 B = 1.76 GB)

__device__ __forceinline__ MsmPoint neg_point(MsmPoint p) {
    p.y = neg30(p.y);
    return p;
}

// [k]D for a canonical 256-bit scalar k (8 x u32, little endian); tab: this thread's 15-entry slot
__device__ __forceinline__ MsmPoint scalar_mul_w4(const MsmPoint &D, const uint32_t k[8], MsmPoint *tab) {
    tab[0] = D;  // tab[i] = (i + 1) D
    for (int i = 2; i <= GF_TAB; i++) {
        MsmPoint t;
        if (i & 1) t = g1_add30(tab[i - 2], D);
        else t = g1_dbl30(tab[i / 2 - 1]);
        tab[i - 1] = t;
    }
    MsmPoint acc = MsmPoint::infinity();
    for (int nib = 63; nib >= 0; nib--) {
#pragma nounroll
        for (int j = 0; j < GF_W; j++) acc = g1_dbl30(acc);
        uint32_t limb = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) limb = (q == (nib >> 3)) ? k[q] : limb;
        const uint32_t dgt = (limb >> (4 * (nib & 7))) & 15u;
        if (dgt) acc = g1_add30(acc, tab[dgt - 1]);
    }
    return acc;
}

// one DIF stage over butterflies [t0, t0 + count): half-size m; tw[e] = w^e (canonical), e < d / 2.
// last stage (m == 1): both outputs are multiplied by `scale` (canonical 1/d) instead of a twiddle.
__global__ __launch_bounds__(256) void k_gfft_stage(MsmPoint *P, size_t d, size_t m, const Fr *tw, Fr scale, size_t t0, size_t count,
                                                    MsmPoint *table) {
    const size_t lt = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lt >= count) return;
    const size_t t = t0 + lt;
    const size_t j = t / m, i = t % m;
    const size_t i0 = j * 2 * m + i, i1 = i0 + m;
    const MsmPoint x = P[i0], y = P[i1];
    MsmPoint s = g1_add30(x, y);
    MsmPoint D = g1_add30(x, neg_point(y));
    MsmPoint *tab = table + lt * GF_TAB;
    uint32_t k[8];
    if (m == 1) {
#pragma unroll
        for (int q = 0; q < 8; q++) k[q] = scale.v[q];
        P[i0] = scalar_mul_w4(s, k, tab);
        P[i1] = scalar_mul_w4(D, k, tab);
        return;
    }
    const Fr w = tw[i * (d / (2 * m))];
#pragma unroll
    for (int q = 0; q < 8; q++) k[q] = w.v[q];
    P[i0] = s;
    P[i1] = scalar_mul_w4(D, k, tab);
}

// d == 1: L_0 = gs[0]; otherwise the bit-reversed read-out into the canonical XYZZ form
__global__ __launch_bounds__(256) void k_gfft_finish(const MsmPoint *P, size_t d, uint32_t bits, G1Xyzz *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d) return;
    size_t src = bits ? (size_t)(__brevll((unsigned long long)i) >> (64 - bits)) : 0;
    out[i] = g1_xyzz_from30(P[src]);
}

__global__ __launch_bounds__(256) void k_affine_to_msm_points(const G1Affine *in, MsmPoint *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = g1_from_affine30(g1_affine_to30(in[i]), false);
}

__global__ __launch_bounds__(256) void k_from_mont_table(Fr *v, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = from_mont(v[i]);
}

}  // namespace kzg

using namespace kzg;

extern "C" int kzg_srs_lagrange_from_monomial_g1(kzg_ctx *ctx, const kzg_srs *mono, kzg_srs **out) {
    if (!ctx || !mono || !out) return KZG_ERR_SHAPE;
    kzg::Guard g(ctx);
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t d = mono->n;
    if (d == 0 || (d & (d - 1))) return fail(ctx, KZG_ERR_SHAPE, "assert!(d & (d - 1) == 0) (src/eval_form.rs:255-256)");
    const uint32_t exp = (uint32_t)ilog2_ceil(d);
    if (exp >= FR_TWO_ADICITY) return fail(ctx, KZG_ERR_DEGREE_TOO_LARGE, "domain too large");
    if (exp > 24) return fail(ctx, KZG_ERR_SHAPE, "compute_lagrange_basis on the GPU is limited to d <= 2^24 (documented limit)");
    hipStream_t st = ctx->lanes[0].stream;
    kzg_srs *s = nullptr;
    KZG_TRY(srs_alloc(ctx, d, &s));
    MsmPoint *P = nullptr, *table = nullptr;
    G1Xyzz *rows = nullptr;
    Fr *tw = nullptr;
    int rc = KZG_OK;
    const size_t half = d > 1 ? d / 2 : 1;
    const size_t chunk = half < GF_CHUNK ? half : GF_CHUNK;
    if (hipMalloc((void **)&P, d * sizeof(MsmPoint)) != hipSuccess || hipMalloc((void **)&rows, d * sizeof(G1Xyzz)) != hipSuccess ||
        hipMalloc((void **)&table, chunk * GF_TAB * sizeof(MsmPoint)) != hipSuccess || hipMalloc((void **)&tw, half * sizeof(Fr)) != hipSuccess)
        rc = fail(ctx, KZG_ERR_ALLOC, "hipMalloc(group-FFT workspace)");
    if (rc == KZG_OK) {
        const Fr omega_inv = inv(host_omega(exp));
        const Fr dinv = from_mont(inv(from_u64<FrParams>((uint64_t)d)));  // canonical
        KZG_LAUNCH(ctx, st, "k_affine_to_msm_points", k_affine_to_msm_points, (unsigned)((d + 255) / 256), 256, 0, mono->table, P, d);
        rc = pow_table(ctx, st, omega_inv, Fr::one(), half, tw);
        if (rc == KZG_OK) KZG_LAUNCH(ctx, st, "k_from_mont_table", k_from_mont_table, (unsigned)((half + 255) / 256), 256, 0, tw, half);
        for (size_t m = d / 2; m >= 1 && rc == KZG_OK; m /= 2) {
            for (size_t t0 = 0; t0 < d / 2; t0 += chunk) {
                size_t cnt = d / 2 - t0 < chunk ? d / 2 - t0 : chunk;
                KZG_LAUNCH(ctx, st, "k_gfft_stage", k_gfft_stage, (unsigned)((cnt + 255) / 256), 256, 0, P, d, m, tw, dinv, t0, cnt, table);
            }
        }
        if (rc == KZG_OK) {
            KZG_LAUNCH(ctx, st, "k_gfft_finish", k_gfft_finish, (unsigned)((d + 255) / 256), 256, 0, P, d, exp, rows);
            rc = srs_finish_from_xyzz(ctx, s, rows);
        }
    }
    hipStreamSynchronize(st);
    if (hipGetLastError() != hipSuccess && rc == KZG_OK) rc = fail(ctx, KZG_ERR_HIP, "group FFT kernels failed");
    if (P) hipFree(P);
    if (rows) hipFree(rows);
    if (table) hipFree(table);
    if (tw) hipFree(tw);
    if (rc != KZG_OK) {
        kzg_srs_free(nullptr, s);
        return rc;
    }
    *out = s;
    return KZG_OK;
}
