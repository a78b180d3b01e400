// poly.hip -- Fr vector kernels that feed the witness MSMs without leaving the device:
//   * (p - y)/(X - x): the reference's long_division by a linear factor (src/polynomial.rs:193-227 as
//     called from src/coeff_form.rs:71) is the Horner recurrence q_{i-1} = a_i + x q_i, restated as a
//     three-kernel blocked suffix scan (block partials, carry scan, apply) -- 96 B/coefficient of HBM
//     traffic instead of an n-step serial chain.  The remainder p(x) - y falls out of the carry scan.
//   * div_by_omega_i (src/eval_form.rs:58-84) in closed form with per-domain tables
//     w^t and 1/(w^t - 1) (one batch inversion per domain, not one inversion per element).
//   * batch inversion, format conversion, Polynomial::eval (src/polynomial.rs:156-165), and the
//     counter-based synthetic input generator used by bench.py.
#include "common.h"

namespace kzg {

// ---------------------------------------------------------------------------------------------
// format conversion and batch inversion
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fr_convert(Fr *data, size_t n, int to_m) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    data[i] = to_m ? to_mont(data[i]) : from_mont(data[i]);
}

int fr_convert(kzg_ctx *ctx, hipStream_t stream, Fr *d_data, size_t n, int to_m) {
    if (!n) return KZG_OK;
    KZG_LAUNCH(ctx, stream, "k_fr_convert", k_fr_convert, (unsigned)((n + 255) / 256), 256, 0, d_data, n, to_m);
    return KZG_OK;
}

constexpr int BI_K = 16;

// out[i] = 1/in[i] (Montgomery), zeros map to zero.  in != out.  Thread t owns the BI_K elements t, t + T, t + 2T, ... (T threads:
// every load and store of a wave is one contiguous 2 KiB run) and inverts their product once (Montgomery's trick).
__global__ __launch_bounds__(256) void k_batch_inverse(const Fr *in, Fr *out, size_t n, size_t T) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T || t >= n) return;
    Fr prod = Fr::one();
    size_t last = t;
    for (size_t i = t; i < n; i += T) {
        out[i] = prod;
        Fr v = in[i];
        if (!v.is_zero()) prod = mul(prod, v);
        last = i;
    }
    Fr iv = inv(prod);
    for (size_t i = last;; i -= T) {
        Fr v = in[i];
        if (v.is_zero()) {
            out[i] = Fr::zero();
        } else {
            Fr r = mul(iv, out[i]);
            iv = mul(iv, v);
            out[i] = r;
        }
        if (i < T) break;  // i == t
    }
}

// a[i] *= 1 / c[i]: the same two sweeps with the quotient formed in the second one (no separate pass over the inverses); a zero
// divisor sets flag bit 0 and zeroes the quotient
__global__ __launch_bounds__(256) void k_batch_inverse_mul(const Fr *c, Fr *a, Fr *tmp, size_t n, size_t T, int *flag) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T || t >= n) return;
    Fr prod = Fr::one();
    size_t last = t;
    bool zero = false;
    for (size_t i = t; i < n; i += T) {
        tmp[i] = prod;
        Fr v = c[i];
        if (!v.is_zero()) prod = mul(prod, v);
        else zero = true;
        last = i;
    }
    if (zero) atomicOr(flag, 1);
    Fr iv = inv(prod);
    for (size_t i = last;; i -= T) {
        Fr v = c[i];
        if (v.is_zero()) {
            a[i] = Fr::zero();
        } else {
            a[i] = mul(a[i], mul(iv, tmp[i]));
            iv = mul(iv, v);
        }
        if (i < T) break;  // i == t
    }
}

// (elements per inversion: 16, 32 and 64 measure the same under 16 concurrent callers -- 390-394 calls/s, profiles/r04_bim_ab.txt --
// and 16 is the fastest alone: 0.133 ms against 0.205 at 32)
#ifndef KZG_BIM_K
#define KZG_BIM_K 16
#endif
constexpr int BIM_K = KZG_BIM_K;
int batch_inverse_mul(kzg_ctx *ctx, hipStream_t stream, const Fr *d_c, Fr *d_a, Fr *d_tmp, size_t n, int *d_flag) {
    if (!n) return KZG_OK;
    size_t T = (n + BIM_K - 1) / BIM_K;
    T = (T + 255) / 256 * 256;
    KZG_LAUNCH(ctx, stream, "k_batch_inverse_mul", k_batch_inverse_mul, (unsigned)(T / 256), 256, 0, d_c, d_a, d_tmp, n, T, d_flag);
    return KZG_OK;
}

int batch_inverse(kzg_ctx *ctx, hipStream_t stream, const Fr *d_in, Fr *d_out, size_t n) {
    if (!n) return KZG_OK;
    size_t T = (n + BI_K - 1) / BI_K;
    T = (T + 255) / 256 * 256;
    KZG_LAUNCH(ctx, stream, "k_batch_inverse", k_batch_inverse, (unsigned)(T / 256), 256, 0, d_in, d_out, n, T);
    return KZG_OK;
}

// ---------------------------------------------------------------------------------------------
// Horner as a blocked suffix scan
// ---------------------------------------------------------------------------------------------
constexpr int HT = 256;          // threads per block
constexpr int HE = 8;            // coefficients per thread
constexpr int HB = HT * HE;      // coefficients per block

__device__ __forceinline__ Fr horner8(const Fr a[HE], const Fr &x) {
    Fr v = a[HE - 1];
#pragma unroll
    for (int k = HE - 2; k >= 0; k--) v = add(mul(v, x), a[k]);
    return v;
}

// The block's 2048 coefficients (64 KiB) move between global memory and the threads' registers through LDS: a thread owns EIGHT
// CONSECUTIVE coefficients (256 bytes), so direct loads / stores put the 64 lanes of an instruction 256 bytes apart -- every
// instruction touched 64 lines for 16 bytes each (k_quotient_apply moved 64 MB in 59 us, k_horner_partials 32 MB in 38 us: 1.1 and
// 0.9 TB/s).  Now lane l of an instruction moves the 16-byte chunk l of a 4 KiB run (whole lines), and the
// thread's own 16 chunks sit in LDS at chunk index 17 t + j: the 272-byte stride spreads 16 lanes of a ds_read_b128 over all 64 banks.
constexpr int H_CHUNKS = HB * 2;                         // 16-byte chunks per block
constexpr int H_LDS_BYTES = (H_CHUNKS + HT) * 16;        // one pad chunk per thread segment
typedef uint32_t __attribute__((ext_vector_type(4))) hchunk;

__device__ __forceinline__ uint32_t h_slot(uint32_t c) { return c + (c >> 4); }

// a[k] = coeffs[block start + 8 t + k] (zero beyond n), through the LDS tile `lds` (H_LDS_BYTES)
__device__ __forceinline__ void load8_tile(const Fr *coeffs, size_t n, hchunk *lds, Fr a[HE]) {
    const int t = threadIdx.x;
    const size_t c0 = (size_t)blockIdx.x * H_CHUNKS, cn = n * 2;     // chunk indices; the array holds cn chunks
    const hchunk *g = (const hchunk *)coeffs;
    hchunk v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const size_t c = c0 + (size_t)(t + j * HT);
        v[j] = c < cn ? g[c] : hchunk{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int j = 0; j < 16; j++) lds[h_slot(t + j * HT)] = v[j];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < HE; k++) {
        const hchunk lo = lds[17 * t + 2 * k], hi = lds[17 * t + 2 * k + 1];
        a[k].v[0] = lo.x; a[k].v[1] = lo.y; a[k].v[2] = lo.z; a[k].v[3] = lo.w;
        a[k].v[4] = hi.x; a[k].v[5] = hi.y; a[k].v[6] = hi.z; a[k].v[7] = hi.w;
    }
}

// x8^(2^k), k < 8: the multipliers of the eight steps of a 256-thread tree / scan.  Computed once on the host -- every thread used to
// square its own copy eight times (8 of a thread's 23-31 multiplications).
struct HornerPowers {
    Fr p[8];
};

// S[b] = sum_{i in block b} a_i x^(i - start_b)
__global__ __launch_bounds__(HT) void k_horner_partials(const Fr *coeffs, size_t n, Fr x, HornerPowers pw, Fr *S) {
    extern __shared__ __attribute__((aligned(16))) hchunk h_lds[];
    int t = threadIdx.x;
    Fr a[HE];
    load8_tile(coeffs, n, h_lds, a);
    Fr v = horner8(a, x);
    __syncthreads();                 // the tile is consumed: its memory carries the tree
    Fr *sh = (Fr *)h_lds;
    sh[t] = v;
    __syncthreads();
#pragma unroll
    for (int st = 0; st < 8; st++) {
        const int off = 1 << st;
        bool active = (t & (2 * off - 1)) == 0;
        if (active) v = add(v, mul(sh[t + off], pw.p[st]));
        __syncthreads();
        if (active) sh[t] = v;
        __syncthreads();
    }
    if (t == 0) S[blockIdx.x] = v;
}

// H[b] = sum_{b' > b} S[b'] X^(b'-b-1) with X = x^HB; px = S[0] + X H[0] = p(x).  Single block.  The multipliers of the ten scan
// steps, (X^g)^(2^k), come from the host: every thread used to raise X to the g-th power by a 64-step square-and-multiply loop of its
// own and to square the result ten times -- 64 us for a kernel that moves 32 KB (profiles/r06_prof_witness_coeff.txt).
constexpr int HS_T = 1024;
struct ScanPowers {
    Fr p[10];
};
__global__ __launch_bounds__(HS_T) void k_horner_scan(const Fr *S, uint32_t nblk, Fr X, ScanPowers pw, Fr *H, Fr *px) {
    __shared__ Fr sh[HS_T];
    int t = threadIdx.x;
    uint32_t g = (nblk + HS_T - 1) / HS_T;  // blocks per thread
    uint32_t b0 = t * g;
    // segment value v_t = sum_k S[b0+k] X^k
    Fr v = Fr::zero();
    for (uint32_t k = g; k-- > 0;) {
        Fr s = (b0 + k < nblk) ? S[b0 + k] : Fr::zero();
        v = g == 1 ? s : add(mul(v, X), s);
    }
    // inclusive suffix scan A_t = v_t + M A_{t+1}, M = X^g
    sh[t] = v;
    __syncthreads();
#pragma unroll
    for (int st = 0; st < 10; st++) {
        const int off = 1 << st;
        Fr o = (t + off < HS_T) ? sh[t + off] : Fr::zero();
        __syncthreads();
        v = add(v, mul(o, pw.p[st]));
        sh[t] = v;
        __syncthreads();
    }
    Fr carry = (t + 1 < HS_T) ? sh[t + 1] : Fr::zero();
    for (uint32_t k = g; k-- > 0;) {
        if (b0 + k < nblk) {
            H[b0 + k] = carry;
            carry = add(S[b0 + k], mul(carry, X));
        }
    }
    if (t == 0) *px = carry;
}

// q_i = sum_{j > i} a_j x^(j-i-1) for i < n - 1.  coeffs may alias q (a block reads its whole tile before it writes).
__global__ __launch_bounds__(HT) void k_quotient_apply(const Fr *coeffs, size_t n, Fr x, HornerPowers pw, const Fr *H, Fr *q) {
    extern __shared__ __attribute__((aligned(16))) hchunk h_lds[];
    int t = threadIdx.x;
    Fr a[HE];
    load8_tile(coeffs, n, h_lds, a);
    Fr v = horner8(a, x);
    const Fr Hb = H[blockIdx.x];
    if (t == HT - 1) v = add(v, mul(pw.p[0], Hb));  // fold the carry from higher blocks in
    __syncthreads();                 // every thread has read its coefficients: the tile's memory carries the scan
    Fr *sh = (Fr *)h_lds;
    sh[t] = v;
    __syncthreads();
#pragma unroll
    for (int st = 0; st < 8; st++) {
        const int off = 1 << st;
        Fr o = (t + off < HT) ? sh[t + off] : Fr::zero();
        __syncthreads();
        v = add(v, mul(o, pw.p[st]));
        sh[t] = v;
        __syncthreads();
    }
    Fr carry = (t + 1 < HT) ? sh[t + 1] : Hb;
    __syncthreads();
    // the eight quotient coefficients of this thread go back through the tile: whole-line stores
#pragma unroll
    for (int k = HE - 1; k >= 0; k--) {
        h_lds[17 * t + 2 * k] = hchunk{carry.v[0], carry.v[1], carry.v[2], carry.v[3]};
        h_lds[17 * t + 2 * k + 1] = hchunk{carry.v[4], carry.v[5], carry.v[6], carry.v[7]};
        carry = add(a[k], mul(carry, x));
    }
    __syncthreads();
    const size_t c0 = (size_t)blockIdx.x * H_CHUNKS, cq = (n - 1) * 2;     // q has n - 1 coefficients
    hchunk *g = (hchunk *)q;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const size_t c = c0 + (size_t)(t + j * HT);
        if (c < cq) g[c] = h_lds[h_slot(t + j * HT)];
    }
}

static int horner_common(kzg_ctx *ctx, int lane, const Fr *d_coeffs, size_t n, const Fr &x, Fr **S, Fr **H, Fr **px,
                         uint32_t *nblk_out, HornerPowers *pw_out) {
    hipStream_t st = ctx->lanes[lane].stream;
    uint32_t nblk = (uint32_t)((n + HB - 1) / HB);
    *S = (Fr *)lane_alloc(ctx, lane, (size_t)nblk * sizeof(Fr));
    *H = (Fr *)lane_alloc(ctx, lane, (size_t)nblk * sizeof(Fr));
    *px = (Fr *)lane_alloc(ctx, lane, sizeof(Fr));
    if (!*S || !*H || !*px) return fail(ctx, KZG_ERR_ALLOC, "Horner workspace not reserved");
    Fr x8 = pow_u64(x, HE);
    Fr X = pow_u64(x, HB);
    HornerPowers pw;
    pw.p[0] = x8;
    for (int k = 1; k < 8; k++) pw.p[k] = sqr(pw.p[k - 1]);
    if (!ctx->attr_horner_set) {
        KZG_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)k_horner_partials, hipFuncAttributeMaxDynamicSharedMemorySize, H_LDS_BYTES));
        KZG_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)k_quotient_apply, hipFuncAttributeMaxDynamicSharedMemorySize, H_LDS_BYTES));
        ctx->attr_horner_set = true;
    }
    KZG_LAUNCH(ctx, st, "k_horner_partials", k_horner_partials, nblk, HT, H_LDS_BYTES, d_coeffs, n, x, pw, *S);
    ScanPowers spw;
    spw.p[0] = pow_u64(X, (uint64_t)((nblk + HS_T - 1) / HS_T));
    for (int k = 1; k < 10; k++) spw.p[k] = sqr(spw.p[k - 1]);
    KZG_LAUNCH(ctx, st, "k_horner_scan", k_horner_scan, 1, HS_T, 0, *S, nblk, X, spw, *H, *px);
    *nblk_out = nblk;
    *pw_out = pw;
    return KZG_OK;
}

int poly_eval_run(kzg_ctx *ctx, int lane, const Fr *d_coeffs, size_t n, const Fr &x_mont, Fr *d_y_out) {
    hipStream_t st = ctx->lanes[lane].stream;
    if (n == 0) return fail(ctx, KZG_ERR_SHAPE, "empty polynomial");
    Fr *S, *H, *px;
    HornerPowers pw;
    uint32_t nblk;
    KZG_TRY(horner_common(ctx, lane, d_coeffs, n, x_mont, &S, &H, &px, &nblk, &pw));
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(d_y_out, px, sizeof(Fr), hipMemcpyDeviceToDevice, st));
    return KZG_OK;
}

int quotient_linear_run(kzg_ctx *ctx, int lane, const Fr *d_coeffs, size_t n, const Fr &x_mont, Fr *d_q_out,
                        Fr *d_px_out) {
    hipStream_t st = ctx->lanes[lane].stream;
    if (n == 0) return fail(ctx, KZG_ERR_SHAPE, "empty polynomial");
    Fr *S, *H, *px;
    HornerPowers pw;
    uint32_t nblk;
    KZG_TRY(horner_common(ctx, lane, d_coeffs, n, x_mont, &S, &H, &px, &nblk, &pw));
    if (n > 1) KZG_LAUNCH(ctx, st, "k_quotient_apply", k_quotient_apply, nblk, HT, H_LDS_BYTES, d_coeffs, n, x_mont, pw, H, d_q_out);
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(d_px_out, px, sizeof(Fr), hipMemcpyDeviceToDevice, st));
    return KZG_OK;
}

// ---------------------------------------------------------------------------------------------
// eval-form quotient (div_by_omega_i)
// ---------------------------------------------------------------------------------------------
struct EvalDomainTables {
    size_t d = 0;
    Fr *pw = nullptr;    // w^t
    Fr *inv1 = nullptr;  // 1/(w^t - 1), inv1[0] = 0
};

void eval_tabs_free(kzg_ctx *ctx) {
    for (auto &kv : ctx->eval_tabs) {
        if (kv.second->pw) hipFree(kv.second->pw);
        if (kv.second->inv1) hipFree(kv.second->inv1);
        delete kv.second;
    }
    ctx->eval_tabs.clear();
}

__global__ __launch_bounds__(256) void k_sub_one(const Fr *in, Fr *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = sub(in[i], Fr::one());
}

static int eval_tables(kzg_ctx *ctx, hipStream_t st, uint32_t log_d, EvalDomainTables **out) {
    // leased lanes (concurrent kzg_witness_eval callers) share the cache: one builder at a time, and a table is published
    // only after the stream that built it has been synchronised
    std::lock_guard<std::mutex> clk(ctx->cache_mu);
    auto it = ctx->eval_tabs.find(log_d);
    if (it != ctx->eval_tabs.end()) {
        *out = it->second;
        return KZG_OK;
    }
    size_t d = (size_t)1 << log_d;
    EvalDomainTables *t = new EvalDomainTables();
    t->d = d;
    Fr *tmp = nullptr;
    int rc = KZG_OK;
    if (hipMalloc((void **)&t->pw, d * sizeof(Fr)) != hipSuccess || hipMalloc((void **)&t->inv1, d * sizeof(Fr)) != hipSuccess ||
        hipMalloc((void **)&tmp, d * sizeof(Fr)) != hipSuccess)
        rc = fail(ctx, KZG_ERR_ALLOC, "hipMalloc(evaluation-domain tables)");
    if (rc == KZG_OK) rc = pow_table(ctx, st, host_omega(log_d), Fr::one(), d, t->pw);
    if (rc == KZG_OK) {
        KZG_LAUNCH(ctx, st, "k_sub_one", k_sub_one, (unsigned)((d + 255) / 256), 256, 0, t->pw, tmp, d);
        rc = batch_inverse(ctx, st, tmp, t->inv1, d);
    }
    if (rc == KZG_OK && hipStreamSynchronize(st) != hipSuccess) rc = fail(ctx, KZG_ERR_HIP, "evaluation-domain tables");
    if (tmp) hipFree(tmp);
    if (rc != KZG_OK) {  // nothing half-built stays behind
        if (t->pw) hipFree(t->pw);
        if (t->inv1) hipFree(t->inv1);
        delete t;
        return rc;
    }
    ctx->eval_tabs[log_d] = t;
    *out = t;
    return KZG_OK;
}

__device__ __forceinline__ Fr block_sum_fr(Fr v, Fr *sh) {
    int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int off = blockDim.x >> 1; off > 0; off >>= 1) {
        if (t < off) sh[t] = add(sh[t], sh[t + off]);
        __syncthreads();
    }
    Fr r = sh[0];
    __syncthreads();
    return r;
}

// q_j = (f_j - y) / (w^j - w^m) for j != m; partial[b] = sum_{j in block} q_j w^((j-m) mod d)
__global__ __launch_bounds__(256) void k_eval_quotient(const Fr *evals, size_t d, size_t m, Fr wm_inv, const Fr *pw,
                                                       const Fr *inv1, Fr *q, Fr *partial) {
    __shared__ Fr sh[256];
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fr contrib = Fr::zero();
    if (j < d && j != m) {
        Fr y = evals[m];
        size_t t = (j + d - m) & (d - 1);
        Fr qj = mul(sub(evals[j], y), mul(wm_inv, inv1[t]));
        q[j] = qj;
        contrib = mul(qj, pw[t]);
    }
    Fr s = block_sum_fr(contrib, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// q_m = - sum partial
__global__ __launch_bounds__(256) void k_eval_quotient_fix(const Fr *partial, uint32_t nblk, size_t m, Fr *q) {
    __shared__ Fr sh[256];
    Fr acc = Fr::zero();
    for (uint32_t b = threadIdx.x; b < nblk; b += blockDim.x) acc = add(acc, partial[b]);
    Fr s = block_sum_fr(acc, sh);
    if (threadIdx.x == 0) q[m] = neg(s);
}

// builds (first use) the per-domain tables and waits for them: callers that run quotient_eval_run on several lanes at once
int eval_tables_ready(kzg_ctx *ctx, int lane, uint32_t log_d) {
    EvalDomainTables *tab = nullptr;
    KZG_TRY(eval_tables(ctx, ctx->lanes[lane].stream, log_d, &tab));
    KZG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->lanes[lane].stream));
    return KZG_OK;
}

int quotient_eval_run(kzg_ctx *ctx, int lane, const Fr *d_evals, uint32_t log_d, size_t m, int sfmt, Fr *d_q_out) {
    (void)sfmt;  // the map f -> q is linear with Montgomery-form constants: either input form is preserved
    hipStream_t st = ctx->lanes[lane].stream;
    size_t d = (size_t)1 << log_d;
    if (m >= d) return fail(ctx, KZG_ERR_SHAPE, "evaluation index out of range (reference: index panic)");
    EvalDomainTables *tab = nullptr;
    KZG_TRY(eval_tables(ctx, st, log_d, &tab));
    uint32_t nblk = (uint32_t)((d + 255) / 256);
    Fr *partial = (Fr *)lane_alloc(ctx, lane, (size_t)nblk * sizeof(Fr));
    if (!partial) return fail(ctx, KZG_ERR_ALLOC, "eval quotient workspace not reserved");
    Fr omega = host_omega(log_d);
    Fr wm_inv = inv(pow_u64(omega, (uint64_t)m));
    KZG_LAUNCH(ctx, st, "k_eval_quotient", k_eval_quotient, nblk, 256, 0, d_evals, d, m, wm_inv, tab->pw, tab->inv1,
               d_q_out, partial);
    KZG_LAUNCH(ctx, st, "k_eval_quotient_fix", k_eval_quotient_fix, 1, 256, 0, partial, nblk, m, d_q_out);
    return KZG_OK;
}

// ---------------------------------------------------------------------------------------------
// synthetic inputs: element i = (SplitMix64(seed + 4i + k))_{k<4} as a 256-bit LE integer, mod r
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void k_fill_random(Fr *dst, size_t n, uint64_t seed, int u64_valued, int sfmt) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr v = Fr::zero();
    const int limbs = u64_valued ? 1 : 4;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint64_t w = k < limbs ? splitmix64(seed + 4ull * i + (uint64_t)k) : 0ull;
        v.v[2 * k] = (uint32_t)w;
        v.v[2 * k + 1] = (uint32_t)(w >> 32);
    }
    Fr m = mul(v, Fr::r2());  // v < 2^256: Montgomery form of (v mod r)
    dst[i] = (sfmt == KZG_FR_MONT_LE_32) ? m : from_mont(m);
}

}  // namespace kzg

using namespace kzg;

extern "C" int kzg_fill_random_fr(kzg_ctx *ctx, void *dst_dev, size_t n, uint64_t seed, int u64_valued, int sfmt) {
    if (!ctx || (!dst_dev && n)) return KZG_ERR_SHAPE;
    kzg::Guard g(ctx);
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (!n) return KZG_OK;
    hipStream_t st = ctx->lanes[0].stream;
    KZG_LAUNCH(ctx, st, "k_fill_random", k_fill_random, (unsigned)((n + 255) / 256), 256, 0, (Fr *)dst_dev, n, seed,
               u64_valued, sfmt);
    KZG_HIP_CHECK(ctx, hipStreamSynchronize(st));
    return KZG_OK;
}
