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
#include "fr29.h"

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

// Every multiplication of the three kernels is by a constant of the launch (x, a power of x), so it is the NTT's Shoup product on
// 9 x 29-bit limbs (fr29.h: 143 multiply-adds and 37 shifts / masks, no carry folds) instead of the saturated Montgomery product
// (128 multiply-adds + 156 carry folds + moves: ~380 instructions, out of line).  The data keeps whatever form it has (the product
// with the plain integer x is the product of the residues), sums stay unreduced between products (a product accepts any value
// below 2^261 with limbs below 1.5 * 2^30 and returns one below 2r), and what leaves a kernel is canonical.
struct ShoupConst {
    Fr29 w, wp;                   // the constant as a plain residue and floor(w 2^261 / r)
};
static ShoupConst shoup_const(const Fr &c_mont) {
    ShoupConst c;
    fr29_shoup_from_twiddle(fr29_twiddle_from_mont(c_mont), c.w, c.wp);
    return c;
}
KZG_HD Fr29 mul_const(const Fr29 &v, const ShoupConst &c) { return mulshoup29(v, c.w, c.wp); }
// nine-limb values of a block in LDS, limb-major (conflict-free 32-bit accesses)
template <int T>
__device__ __forceinline__ void sh_put(uint32_t *sh, int t, const Fr29 &v) {
#pragma unroll
    for (int i = 0; i < R29_N; i++) sh[i * T + t] = v.v[i];
}
template <int T>
__device__ __forceinline__ Fr29 sh_get(const uint32_t *sh, int t) {
    Fr29 v;
#pragma unroll
    for (int i = 0; i < R29_N; i++) v.v[i] = sh[i * T + t];
    return v;
}

// a_0 + x (a_1 + x (... a_7)): limbs below 2^30, value below 2^256 + 2r; not normalised.  (Splitting it into an even and an odd chain
// by x^2 issued as one interleaved stream -- four products deep instead of seven -- changed nothing: DESIGN.md section 3.4.)
KZG_HD Fr29 horner8(const Fr a[HE], const ShoupConst &x) {
    Fr29 v = fr29_unpack(a[HE - 1]);
#pragma unroll
    for (int k = HE - 2; k >= 0; k--) v = fr29_add_lazy(mul_const(v, x), fr29_unpack(a[k]));
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

// the constants of a launch: x, and x^8 squared k times (k < 8) -- the multipliers of the block tree / scan steps.  They come from
// the host: every thread used to square its own copy eight times.
struct HornerConsts {
    ShoupConst x;
    ShoupConst p[8];
};

// S[b] = sum_{i in block b} a_i x^(i - start_b)
__global__ __launch_bounds__(HT) void k_horner_partials(const Fr *coeffs, size_t n, HornerConsts c, Fr *S) {
    extern __shared__ __attribute__((aligned(16))) hchunk h_lds[];
    int t = threadIdx.x;
    Fr a[HE];
    load8_tile(coeffs, n, h_lds, a);
    Fr29 v = fr29_normalize(horner8(a, c.x));
    __syncthreads();                 // the tile is consumed: its memory carries the tree
    uint32_t *sh = (uint32_t *)h_lds;
    sh_put<HT>(sh, t, v);
    __syncthreads();
#pragma unroll
    for (int st = 0; st < 8; st++) {     // v grows by less than 2r per step: below 21 r at the end
        const int off = 1 << st;
        bool active = (t & (2 * off - 1)) == 0;
        if (active) v = fr29_normalize(fr29_add_lazy(v, mul_const(sh_get<HT>(sh, t + off), c.p[st])));
        __syncthreads();
        if (active) sh_put<HT>(sh, t, v);
        __syncthreads();
    }
    if (t == 0) S[blockIdx.x] = fr29_canonical(v);
}

// H[b] = sum_{b' > b} S[b'] X^(b'-b-1) with X = x^HB; px = S[0] + X H[0] = p(x).  Single block of T = blockDim.x threads, T the
// power of two that covers nblk, 64 <= T <= 512.  The block runs on ONE CU, so what counts is the number of products a SIMD goes
// through in sequence: 2 g + log2 T with g = nblk / T sums per thread, times half the waves per SIMD (a lone wave's dependent
// multiply-add chain leaves room for a second one; from there on waves share the issue slots).  Measured, same box
// (profiles/r06_ab_horner.txt): 512 partial sums of a 2^20 polynomial -- 1024 threads (half of them multiplying zeros) 0.037 ms,
// 512 threads 0.0206, 256 threads 0.0195; 8192 of a 2^24 polynomial -- 1024 threads 0.0606, 512 threads 0.0584, 256 threads 0.0684 ms.  The
// multipliers of the scan steps, (X^g)^(2^k), come from the host: every thread used to raise X to the g-th power by a 64-step
// square-and-multiply loop of its own and to square the result ten times -- 64 us for a kernel that moves 32 KB
// (profiles/r06_prof_witness_coeff.txt).
constexpr int HS_T = 512;
struct ScanConsts {
    ShoupConst X;
    ShoupConst p[9];
};
__global__ __launch_bounds__(HS_T) void k_horner_scan(const Fr *S, uint32_t nblk, ScanConsts c, Fr *H, Fr *px) {
    __shared__ uint32_t sh[R29_N * HS_T];
    const int t = threadIdx.x, T = blockDim.x;
    uint32_t g = (nblk + T - 1) / T;  // blocks per thread
    uint32_t b0 = t * g;
    // segment value v_t = sum_k S[b0+k] X^k
    Fr29 v = fr29_zero();
    for (uint32_t k = g; k-- > 0;) {
        Fr29 s = (b0 + k < nblk) ? fr29_unpack(S[b0 + k]) : fr29_zero();
        v = g == 1 ? s : fr29_add_lazy(mul_const(v, c.X), s);
    }
    v = fr29_normalize(v);
    // inclusive suffix scan A_t = v_t + M A_{t+1}, M = X^g; v grows by less than 2r per step: below 25 r at the end
#pragma unroll
    for (int i = 0; i < R29_N; i++) sh[i * T + t] = v.v[i];
    __syncthreads();
#pragma unroll
    for (int st = 0; st < 9; st++) {
        const int off = 1 << st;
        if (off < T) {               // uniform
            Fr29 o = fr29_zero();
            if (t + off < T) {
#pragma unroll
                for (int i = 0; i < R29_N; i++) o.v[i] = sh[i * T + t + off];
            }
            __syncthreads();
            v = fr29_normalize(fr29_add_lazy(v, mul_const(o, c.p[st])));
#pragma unroll
            for (int i = 0; i < R29_N; i++) sh[i * T + t] = v.v[i];
            __syncthreads();
        }
    }
    Fr29 carry = fr29_zero();
    if (t + 1 < T) {
#pragma unroll
        for (int i = 0; i < R29_N; i++) carry.v[i] = sh[i * T + t + 1];
    }
    for (uint32_t k = g; k-- > 0;) {
        if (b0 + k < nblk) {
            H[b0 + k] = fr29_canonical(carry);
            carry = fr29_normalize(fr29_add_lazy(fr29_unpack(S[b0 + k]), mul_const(carry, c.X)));
        }
    }
    if (t == 0) *px = fr29_canonical(carry);
}

// q_i = sum_{j > i} a_j x^(j-i-1) for i < n - 1.  coeffs may alias q (a block reads its whole tile before it writes).
__global__ __launch_bounds__(HT) void k_quotient_apply(const Fr *coeffs, size_t n, HornerConsts c, const Fr *H, Fr *q) {
    extern __shared__ __attribute__((aligned(16))) hchunk h_lds[];
    int t = threadIdx.x;
    Fr a[HE];
    load8_tile(coeffs, n, h_lds, a);
    Fr29 v = horner8(a, c.x);
    const Fr29 Hb = fr29_unpack(H[blockIdx.x]);
    if (t == HT - 1) v = fr29_add_lazy(v, mul_const(Hb, c.p[0]));  // fold the carry from higher blocks in
    v = fr29_normalize(v);
    __syncthreads();                 // every thread has read its coefficients: the tile's memory carries the scan
    uint32_t *sh = (uint32_t *)h_lds;
    sh_put<HT>(sh, t, v);
    __syncthreads();
#pragma unroll
    for (int st = 0; st < 8; st++) {     // v grows by less than 2r per step: below 23 r at the end
        const int off = 1 << st;
        Fr29 o = (t + off < HT) ? sh_get<HT>(sh, t + off) : fr29_zero();
        __syncthreads();
        v = fr29_normalize(fr29_add_lazy(v, mul_const(o, c.p[st])));
        sh_put<HT>(sh, t, v);
        __syncthreads();
    }
    Fr29 carry = (t + 1 < HT) ? sh_get<HT>(sh, t + 1) : Hb;
    __syncthreads();
    // the eight quotient coefficients of this thread go back through the tile: whole-line stores.  q_(8t+7) is the carry; each of the
    // others is a coefficient (canonical: the caller's scalars are) plus a Shoup product made canonical -- one modular addition, and
    // the next product starts from the sum's limbs.
    Fr out = fr29_canonical(carry);
#pragma unroll
    for (int k = HE - 1; k >= 0; k--) {
        h_lds[17 * t + 2 * k] = hchunk{out.v[0], out.v[1], out.v[2], out.v[3]};
        h_lds[17 * t + 2 * k + 1] = hchunk{out.v[4], out.v[5], out.v[6], out.v[7]};
        if (k) {
            out = add(a[k], fr29_pack_canonical(mul_const(carry, c.x)));
            carry = fr29_unpack(out);
        }
    }
    __syncthreads();
    const size_t c0 = (size_t)blockIdx.x * H_CHUNKS, cq = (n - 1) * 2;     // q has n - 1 coefficients
    hchunk *g = (hchunk *)q;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const size_t cc = c0 + (size_t)(t + j * HT);
        if (cc < cq) g[cc] = h_lds[h_slot(t + j * HT)];
    }
}

static int horner_common(kzg_ctx *ctx, int lane, const Fr *d_coeffs, size_t n, const Fr &x, Fr **S, Fr **H, Fr **px,
                         uint32_t *nblk_out, HornerConsts *hc_out) {
    hipStream_t st = ctx->lanes[lane].stream;
    uint32_t nblk = (uint32_t)((n + HB - 1) / HB);
    *S = (Fr *)lane_alloc(ctx, lane, (size_t)nblk * sizeof(Fr));
    *H = (Fr *)lane_alloc(ctx, lane, (size_t)nblk * sizeof(Fr));
    *px = (Fr *)lane_alloc(ctx, lane, sizeof(Fr));
    if (!*S || !*H || !*px) return fail(ctx, KZG_ERR_ALLOC, "Horner workspace not reserved");
    Fr pk = pow_u64(x, HE);
    HornerConsts hc;
    hc.x = shoup_const(x);
    for (int k = 0; k < 8; k++) {
        hc.p[k] = shoup_const(pk);
        pk = sqr(pk);
    }
    const Fr X = pk;                 // x^(8 * 2^8) = x^HB
    if (!ctx->attr_horner_set) {
        KZG_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)k_horner_partials, hipFuncAttributeMaxDynamicSharedMemorySize, H_LDS_BYTES));
        KZG_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)k_quotient_apply, hipFuncAttributeMaxDynamicSharedMemorySize, H_LDS_BYTES));
        ctx->attr_horner_set = true;
    }
    KZG_LAUNCH(ctx, st, "k_horner_partials", k_horner_partials, nblk, HT, H_LDS_BYTES, d_coeffs, n, hc, *S);
    uint32_t T = 64;                 // scan width: the power of two that covers the partial sums
    while (T < (uint32_t)HS_T && T < nblk) T *= 2;
    ScanConsts sc;
    sc.X = shoup_const(X);
    pk = pow_u64(X, (uint64_t)((nblk + T - 1) / T));
    for (int k = 0; k < 9; k++) {
        sc.p[k] = shoup_const(pk);
        pk = sqr(pk);
    }
    KZG_LAUNCH(ctx, st, "k_horner_scan", k_horner_scan, 1, T, 0, *S, nblk, sc, *H, *px);
    *nblk_out = nblk;
    *hc_out = hc;
    return KZG_OK;
}

int poly_eval_run(kzg_ctx *ctx, int lane, const Fr *d_coeffs, size_t n, const Fr &x_mont, Fr *d_y_out) {
    hipStream_t st = ctx->lanes[lane].stream;
    if (n == 0) return fail(ctx, KZG_ERR_SHAPE, "empty polynomial");
    Fr *S, *H, *px;
    HornerConsts hc;
    uint32_t nblk;
    KZG_TRY(horner_common(ctx, lane, d_coeffs, n, x_mont, &S, &H, &px, &nblk, &hc));
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(d_y_out, px, sizeof(Fr), hipMemcpyDeviceToDevice, st));
    return KZG_OK;
}

int quotient_linear_run(kzg_ctx *ctx, int lane, const Fr *d_coeffs, size_t n, const Fr &x_mont, Fr *d_q_out,
                        Fr *d_px_out) {
    hipStream_t st = ctx->lanes[lane].stream;
    if (n == 0) return fail(ctx, KZG_ERR_SHAPE, "empty polynomial");
    Fr *S, *H, *px;
    HornerConsts hc;
    uint32_t nblk;
    KZG_TRY(horner_common(ctx, lane, d_coeffs, n, x_mont, &S, &H, &px, &nblk, &hc));
    if (n > 1) KZG_LAUNCH(ctx, st, "k_quotient_apply", k_quotient_apply, nblk, HT, H_LDS_BYTES, d_coeffs, n, hc, H, d_q_out);
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(d_px_out, px, sizeof(Fr), hipMemcpyDeviceToDevice, st));
    return KZG_OK;
}

// ---------------------------------------------------------------------------------------------
// eval-form quotient (div_by_omega_i)
// ---------------------------------------------------------------------------------------------
struct EvalDomainTables {
    size_t d = 0;
    Fr *inv1 = nullptr;  // 1/(w^t - 1), inv1[0] = 0.  (w^t itself is not kept: w^t / (w^t - 1) = 1 + inv1[t], see k_eval_quotient)
};

void eval_tabs_free(kzg_ctx *ctx) {
    for (auto &kv : ctx->eval_tabs) {
        if (kv.second->inv1) (void)hipFree(kv.second->inv1);
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
    Fr *pw = nullptr, *tmp = nullptr;        // w^t and w^t - 1: needed while the table is built
    int rc = KZG_OK;
    if (hipMalloc((void **)&pw, d * sizeof(Fr)) != hipSuccess || hipMalloc((void **)&t->inv1, d * sizeof(Fr)) != hipSuccess ||
        hipMalloc((void **)&tmp, d * sizeof(Fr)) != hipSuccess)
        rc = fail(ctx, KZG_ERR_ALLOC, "hipMalloc(evaluation-domain tables)");
    if (rc == KZG_OK) rc = pow_table(ctx, st, host_omega(log_d), Fr::one(), d, pw);
    if (rc == KZG_OK) {
        KZG_LAUNCH(ctx, st, "k_sub_one", k_sub_one, (unsigned)((d + 255) / 256), 256, 0, pw, tmp, d);
        rc = batch_inverse(ctx, st, tmp, t->inv1, d);
    }
    if (rc == KZG_OK && hipStreamSynchronize(st) != hipSuccess) rc = fail(ctx, KZG_ERR_HIP, "evaluation-domain tables");
    if (tmp) (void)hipFree(tmp);
    if (pw) (void)hipFree(pw);
    if (rc != KZG_OK) {  // nothing half-built stays behind
        if (t->inv1) (void)hipFree(t->inv1);
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
__global__ __launch_bounds__(256) void k_eval_quotient(const Fr *evals, size_t d, size_t m, Fr wm_inv, const Fr *inv1, Fr *q,
                                                       Fr *partial) {
    __shared__ Fr sh[256];
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fr contrib = Fr::zero();
    if (j < d && j != m) {
        Fr y = evals[m];
        size_t t = (j + d - m) & (d - 1);
        // q_j = g_j / (w^t - 1) with g_j = (f_j - y) w^-m, and its term of q_m: q_j w^t = g_j (1 + 1 / (w^t - 1)) = g_j + q_j --
        // two products per element and no read of the w^t table (it used to be three and 128 B per element)
        Fr g = mul(sub(evals[j], y), wm_inv);
        Fr qj = mul(g, inv1[t]);
        q[j] = qj;
        contrib = add(g, qj);
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
    Fr wm_inv = pow_u64(omega, (uint64_t)((d - m) & (d - 1)));  // w^-m = w^(d-m): no field inversion on the host (~25 us per call)
    KZG_LAUNCH(ctx, st, "k_eval_quotient", k_eval_quotient, nblk, 256, 0, d_evals, d, m, wm_inv, tab->inv1, d_q_out, partial);
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
