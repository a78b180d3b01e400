// ntt.hip -- radix-2 NTT / iNTT over Fr, natural order in and out, in place.
//
// Semantics of EvaluationDomain::fft / ifft (src/ft.rs:111-140) and best_fft / serial_fft
// (src/ft.rs:274-333): out[i] = sum_j a[j] w^(ij), w = root_of_unity^(2^(32-exp)) per compute_omega
// (src/ft.rs:55-76); the inverse uses w^-1 and scales by d^-1.  The reference's bit-reversal +
// log n in-place passes are a CPU schedule; on the MI355X the transform is a four-step
// decomposition n = n1 * n2 whose sub-transforms (<= 2^12 points, 128 KiB) run entirely in the CU's
// 160 KiB LDS:
//   pass 1: for every column j2, an n1-point NTT over stride-n2 elements, times w^(j2*k1)
//           (two-level twiddle table), written to scratch in the same layout;
//   pass 2: for every row k1, an n2-point NTT over contiguous elements, written transposed
//           (index k1 + n1*k2) back into the caller's buffer -- natural order, no bit-reversal pass.
// Each block takes VEC adjacent columns / rows so every global access is a >= 64..128 B segment.
// Data may be canonical or Montgomery: the butterflies are linear and the twiddles are Montgomery
// constants, so mont_mul(a, w) preserves whichever form a is in.
#include "common.h"

namespace kzg {

struct NttPlan {
    uint32_t log_n = 0, k1 = 0, k2 = 0;
    int inverse = 0;
    Fr *tw1 = nullptr;   // w_{n1}^i, i < n1/2 (or w_n^i for the single-tile case)
    Fr *tw2 = nullptr;   // w_{n2}^i, i < n2/2
    Fr *tw_lo = nullptr; // w_n^i, i < 2^lo_bits
    Fr *tw_hi = nullptr; // w_n^(i << lo_bits)
    uint32_t lo_bits = 0;
    Fr scale;            // d^-1 for the inverse, one otherwise (Montgomery)
};

Fr host_omega(uint32_t exp) {
    // Scalar::root_of_unity().pow_vartime(&[1 << (Scalar::S - exp)])   (src/ft.rs:73)
    return pow_u64(fr_root_of_unity(), 1ull << (FR_TWO_ADICITY - exp));
}

__global__ __launch_bounds__(256) void k_pow_table(Fr base, Fr scale, size_t count, Fr *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    out[i] = mul(scale, pow_u64(base, (uint64_t)i));
}

int pow_table(kzg_ctx *ctx, hipStream_t stream, const Fr &base_mont, const Fr &scale_mont, size_t count, Fr *d_out) {
    if (!count) return KZG_OK;
    KZG_LAUNCH(ctx, stream, "k_pow_table", k_pow_table, (unsigned)((count + 255) / 256), 256, 0, base_mont, scale_mont,
               count, d_out);
    return KZG_OK;
}

extern __shared__ __attribute__((aligned(16))) Fr lds_fr[];

__device__ __forceinline__ uint32_t bitrev(uint32_t x, uint32_t bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

// radix-2 DIT stages over `vec` independent 2^t-point vectors held in LDS as lds[v << t | pos],
// input already in bit-reversed position order.
__device__ __forceinline__ void lds_ntt_stages(Fr *lds, uint32_t t, uint32_t vec, const Fr *tw) {
    const uint32_t half = 1u << (t - 1);
    const uint32_t total = vec << (t - 1);
    for (uint32_t s = 0; s < t; s++) {
        const uint32_t m = 1u << s;
        for (uint32_t b = threadIdx.x; b < total; b += blockDim.x) {
            uint32_t v = b >> (t - 1);
            uint32_t i = b & (half - 1);
            uint32_t j = i & (m - 1);
            uint32_t p0 = (v << t) | (((i >> s) << (s + 1)) | j);
            uint32_t p1 = p0 + m;
            Fr u = lds[p0];
            Fr w = lds[p1];
            if (s != 0) w = mul(w, tw[j << (t - 1 - s)]);  // stage 0 twiddle is 1
            lds[p0] = add(u, w);
            lds[p1] = sub(u, w);
        }
        __syncthreads();
    }
}

// Whole transform in one tile (log_n <= 12).
__global__ __launch_bounds__(1024) void k_ntt_single(Fr *data, uint32_t t, const Fr *tw, Fr scale, int do_scale) {
    const uint32_t n = 1u << t;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) lds_fr[bitrev(i, t)] = data[i];
    __syncthreads();
    if (t) lds_ntt_stages(lds_fr, t, 1, tw);
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        Fr v = lds_fr[i];
        if (do_scale) v = mul(v, scale);
        data[i] = v;
    }
}

// pass 1: columns j2 = blockIdx.x*vec .. +vec-1; element (j1, j2) at in[j1*n2 + j2]
__global__ __launch_bounds__(1024) void k_ntt_pass1(const Fr *in, Fr *out, uint32_t k1, uint32_t k2, uint32_t vec_log,
                                                    const Fr *tw1, const Fr *tw_lo, const Fr *tw_hi, uint32_t lo_bits) {
    const uint32_t n1 = 1u << k1, vec = 1u << vec_log;
    const uint32_t j2_0 = blockIdx.x << vec_log;
    const uint32_t total = n1 << vec_log;
    for (uint32_t e = threadIdx.x; e < total; e += blockDim.x) {
        uint32_t v = e & (vec - 1), j1 = e >> vec_log;  // consecutive threads -> consecutive columns
        lds_fr[(v << k1) | bitrev(j1, k1)] = in[((size_t)j1 << k2) + j2_0 + v];
    }
    __syncthreads();
    lds_ntt_stages(lds_fr, k1, vec, tw1);
    const uint32_t lo_mask = (1u << lo_bits) - 1;
    for (uint32_t e = threadIdx.x; e < total; e += blockDim.x) {
        uint32_t v = e & (vec - 1), kk1 = e >> vec_log;
        uint32_t j2 = j2_0 + v;
        uint64_t ex = (uint64_t)j2 * kk1;  // < n
        Fr val = lds_fr[(v << k1) | kk1];
        Fr w = mul(tw_hi[ex >> lo_bits], tw_lo[ex & lo_mask]);
        out[((size_t)kk1 << k2) + j2] = mul(val, w);
    }
}

// pass 2: rows k1 = blockIdx.x*vec .. +vec-1; row k1 contiguous at in[k1*n2 ..]; out[k1 + n1*k2]
__global__ __launch_bounds__(1024) void k_ntt_pass2(const Fr *in, Fr *out, uint32_t k1, uint32_t k2, uint32_t vec_log,
                                                    const Fr *tw2, Fr scale, int do_scale) {
    const uint32_t n2 = 1u << k2, vec = 1u << vec_log;
    const uint32_t r0 = blockIdx.x << vec_log;
    const uint32_t total = n2 << vec_log;
    for (uint32_t e = threadIdx.x; e < total; e += blockDim.x) {
        uint32_t j2 = e & (n2 - 1), v = e >> k2;  // consecutive threads -> consecutive row elements
        lds_fr[(v << k2) | bitrev(j2, k2)] = in[((size_t)(r0 + v) << k2) + j2];
    }
    __syncthreads();
    lds_ntt_stages(lds_fr, k2, vec, tw2);
    for (uint32_t e = threadIdx.x; e < total; e += blockDim.x) {
        uint32_t v = e & (vec - 1), kk2 = e >> vec_log;  // consecutive threads -> consecutive k1
        Fr val = lds_fr[(v << k2) | kk2];
        if (do_scale) val = mul(val, scale);
        out[((size_t)kk2 << k1) + r0 + v] = val;
    }
}

static bool g_ntt_attr = false;

static int ntt_plan(kzg_ctx *ctx, hipStream_t st, uint32_t log_n, int inverse, NttPlan **out) {
    uint32_t key = log_n * 2 + (inverse ? 1 : 0);
    auto it = ctx->ntt_plans.find(key);
    if (it != ctx->ntt_plans.end()) {
        *out = it->second;
        return KZG_OK;
    }
    if (!g_ntt_attr) {
        KZG_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)k_ntt_single, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096));
        KZG_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)k_ntt_pass1, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096));
        KZG_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)k_ntt_pass2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096));
        g_ntt_attr = true;
    }
    NttPlan *p = new NttPlan();
    p->log_n = log_n;
    p->inverse = inverse;
    Fr w = host_omega(log_n);
    if (inverse) w = inv(w);
    size_t n = (size_t)1 << log_n;
    p->scale = inverse ? inv(from_u64<FrParams>((uint64_t)n)) : Fr::one();
    if (log_n <= 12) {
        p->k1 = log_n;
        p->k2 = 0;
        size_t cnt = log_n ? (n >> 1) : 1;
        KZG_HIP_CHECK(ctx, hipMalloc((void **)&p->tw1, cnt * sizeof(Fr)));
        KZG_TRY(pow_table(ctx, st, w, Fr::one(), cnt, p->tw1));
    } else {
        p->k1 = (log_n + 1) / 2;
        p->k2 = log_n - p->k1;
        size_t n1 = (size_t)1 << p->k1, n2 = (size_t)1 << p->k2;
        Fr w1 = pow_u64(w, (uint64_t)n2);  // w_{n1}
        Fr w2 = pow_u64(w, (uint64_t)n1);  // w_{n2}
        p->lo_bits = p->k1;
        size_t nlo = (size_t)1 << p->lo_bits, nhi = (size_t)1 << (log_n - p->lo_bits);
        KZG_HIP_CHECK(ctx, hipMalloc((void **)&p->tw1, (n1 >> 1) * sizeof(Fr)));
        KZG_HIP_CHECK(ctx, hipMalloc((void **)&p->tw2, (n2 >> 1) * sizeof(Fr)));
        KZG_HIP_CHECK(ctx, hipMalloc((void **)&p->tw_lo, nlo * sizeof(Fr)));
        KZG_HIP_CHECK(ctx, hipMalloc((void **)&p->tw_hi, nhi * sizeof(Fr)));
        KZG_TRY(pow_table(ctx, st, w1, Fr::one(), n1 >> 1, p->tw1));
        KZG_TRY(pow_table(ctx, st, w2, Fr::one(), n2 >> 1, p->tw2));
        KZG_TRY(pow_table(ctx, st, w, Fr::one(), nlo, p->tw_lo));
        KZG_TRY(pow_table(ctx, st, pow_u64(w, (uint64_t)nlo), Fr::one(), nhi, p->tw_hi));
    }
    ctx->ntt_plans[key] = p;
    *out = p;
    return KZG_OK;
}

int ntt_run(kzg_ctx *ctx, int lane, Fr *d_data, uint32_t log_n, int inverse) {
    if (log_n >= FR_TWO_ADICITY) return fail(ctx, KZG_ERR_DEGREE_TOO_LARGE, "polynomial degree too large");
    if (log_n > 24) return fail(ctx, KZG_ERR_SHAPE, "NTT sizes above 2^24 are not implemented yet");
    hipStream_t st = ctx->lanes[lane].stream;
    NttPlan *p = nullptr;
    KZG_TRY(ntt_plan(ctx, st, log_n, inverse, &p));
    if (log_n <= 12) {
        size_t n = (size_t)1 << log_n;
        unsigned threads = n >= 2048 ? 1024 : (n >= 128 ? (unsigned)(n / 2) : 64);
        KZG_LAUNCH(ctx, st, "k_ntt_single", k_ntt_single, 1, threads, n * sizeof(Fr), d_data, log_n, p->tw1, p->scale,
                   inverse);
        return KZG_OK;
    }
    size_t n = (size_t)1 << log_n;
    Fr *scratch = (Fr *)lane_alloc(ctx, lane, n * sizeof(Fr));
    if (!scratch) return fail(ctx, KZG_ERR_ALLOC, "NTT workspace not reserved");
    // vec adjacent columns/rows per block: as many as fit 128 KiB of LDS, at most 4 (128-B segments)
    uint32_t vec1 = 12 - p->k1 < 2 ? 12 - p->k1 : 2;
    uint32_t vec2 = 12 - p->k2 < 2 ? 12 - p->k2 : 2;
    size_t lds1 = ((size_t)1 << (p->k1 + vec1)) * sizeof(Fr), lds2 = ((size_t)1 << (p->k2 + vec2)) * sizeof(Fr);
    unsigned g1 = 1u << (p->k2 - vec1), g2 = 1u << (p->k1 - vec2);
    KZG_LAUNCH(ctx, st, "k_ntt_pass1", k_ntt_pass1, g1, 1024, lds1, d_data, scratch, p->k1, p->k2, vec1, p->tw1, p->tw_lo,
               p->tw_hi, p->lo_bits);
    KZG_LAUNCH(ctx, st, "k_ntt_pass2", k_ntt_pass2, g2, 1024, lds2, scratch, d_data, p->k1, p->k2, vec2, p->tw2, p->scale,
               inverse);
    return KZG_OK;
}

}  // namespace kzg
