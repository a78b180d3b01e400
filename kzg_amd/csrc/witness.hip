// witness.hip -- KZGProver::create_witness_batched (src/coeff_form.rs:83-111), coset NTTs
// (src/ft.rs:142-178) and compute_lagrange_basis from the monomial SRS (src/eval_form.rs:254-280).
//
// create_witness_batched returns r = I (the interpolant of the k points) and w = [(p - I)/Z]_1 with
// Z = prod (X - x_i).  The reference builds a sub-product tree and runs an (n-k) x (k+1) schoolbook
// long_division whose outer loop is sequential (2.7e8 dependent mul-subs at n = 2^20, k = 256).  The
// same polynomials are obtained here by GPU-shaped algorithms (results are unique, so parity holds):
//   * Z: k sequential multiplications by (X - x_i), each a k-wide parallel update (one workgroup);
//   * I: barycentric form  I = sum_i y_i / Z'(x_i) * Z/(X - x_i)  -- one thread per point for
//     Z'(x_i) = prod_{j != i}(x_i - x_j) and for the synthetic division Z/(X - x_i), one batch
//     inversion, then a column sum;
//   * (p - I)/Z: evaluate p, I, Z on the coset g*H (|H| = N >= n), divide pointwise (one batch
//     inversion), interpolate back: three forward coset NTTs + one inverse.  The division is exact iff
//     the top k coefficients of the interpolated quotient vanish, which is the reference's
//     `Some(remainder) => Err(PointNotOnPolynomial)` test.
#include "common.h"

namespace kzg {

// ---------------------------------------------------------------------------------------------
// distribute_powers: v[i] *= g^i   (src/ft.rs:142-166)
// ---------------------------------------------------------------------------------------------
// Thread t scales elements t, t + T, t + 2T, ... (T = number of threads: coalesced) with a running g^(t + jT).  g^t comes from
// two cached 1024-entry tables (g^t = g^(1024 (t >> 10)) g^(t & 1023): one multiplication instead of a 30-multiplication
// square-and-multiply), so few elements per thread suffice and the kernel fills the chip: 2^20 elements in ~20 us (the
// 32-elements-per-thread version it replaces ran 512 waves on 1024 SIMDs for 70-100 us).
constexpr int DP_E = 4;            // elements per thread
constexpr size_t DP_TAB = 1024;    // table entries: T <= DP_TAB^2 threads
__global__ __launch_bounds__(256) void k_distribute_powers(Fr *data, size_t n, const Fr *lo_tab, const Fr *hi_tab, Fr gT, size_t T) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T || t >= n) return;
    Fr u = mul(hi_tab[t >> 10], lo_tab[t & (DP_TAB - 1)]);
    for (size_t i = t; i < n; i += T) {
        data[i] = mul(data[i], u);
        u = mul(u, gT);
    }
}

// The numerator of create_witness_batched on its way to the coset: A[i] = (p[i] - I[i]) * g^i for i < n (I has k < n coefficients,
// p is converted to Montgomery form on the fly when `to_m`), zero for n <= i < N.  One pass instead of three (zero-padded load,
// prefix subtraction, distribute_powers); same thread / element mapping as k_distribute_powers.
__global__ __launch_bounds__(256) void k_numerator_on_coset(const Fr *p, size_t n, const Fr *I, size_t k, Fr *A, size_t N, int to_m,
                                                            const Fr *lo_tab, const Fr *hi_tab, Fr gT, size_t T) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T || t >= N) return;
    const bool scale = lo_tab != nullptr;  // nullptr: the shift is 1 (plain domain), nothing to multiply by
    Fr u = scale ? mul(hi_tab[t >> 10], lo_tab[t & (DP_TAB - 1)]) : Fr::one();
    for (size_t i = t; i < N; i += T) {
        if (i < n) {
            Fr v = p[i];
            if (to_m) v = to_mont(v);
            if (i < k) v = sub(v, I[i]);
            if (scale) {
                v = mul(v, u);
                u = mul(u, gT);
            }
            A[i] = v;
        } else {
            A[i] = Fr::zero();
        }
    }
}

// device tables for the coset generator g, built once per context and generator (7, its inverse, and the rare 7^k of
// kzg_witness_coeff_batched when opening points lie on the coset)
static int coset_tables(kzg_ctx *ctx, hipStream_t st, const Fr &g, const Fr **lo_tab, const Fr **hi_tab) {
    std::array<uint32_t, 8> key;
    for (int i = 0; i < 8; i++) key[i] = g.v[i];
    std::lock_guard<std::mutex> clk(ctx->cache_mu);  // leased lanes share the cache: one builder at a time
    for (auto &ct : ctx->coset_tabs)
        if (ct.first == key) {
            *lo_tab = (const Fr *)ct.second;
            *hi_tab = *lo_tab + DP_TAB;
            return KZG_OK;
        }
    Fr *d = nullptr;
    KZG_HIP_CHECK(ctx, hipMalloc((void **)&d, 2 * DP_TAB * sizeof(Fr)));
    int rc = pow_table(ctx, st, g, Fr::one(), DP_TAB, d);
    if (rc == KZG_OK) rc = pow_table(ctx, st, pow_u64(g, (uint64_t)DP_TAB), Fr::one(), DP_TAB, d + DP_TAB);
    // published only once built: a later call may run on another stream and would read the tables unsynchronised
    if (rc == KZG_OK && hipStreamSynchronize(st) != hipSuccess) rc = fail(ctx, KZG_ERR_HIP, "coset tables");
    if (rc != KZG_OK) {
        hipFree(d);
        return rc;
    }
    ctx->coset_tabs.emplace_back(key, (void *)d);
    *lo_tab = d;
    *hi_tab = d + DP_TAB;
    return KZG_OK;
}

static int numerator_on_coset(kzg_ctx *ctx, hipStream_t st, const Fr *p, size_t n, const Fr *I, size_t k, Fr *A, size_t N, int to_m,
                              const Fr &g) {
    size_t T = (N + DP_E - 1) / DP_E;
    T = (T + 255) / 256 * 256;
    if (T > DP_TAB * DP_TAB) T = DP_TAB * DP_TAB;
    const Fr *lo_tab = nullptr, *hi_tab = nullptr;
    if (!(g == Fr::one())) KZG_TRY(coset_tables(ctx, st, g, &lo_tab, &hi_tab));
    KZG_LAUNCH(ctx, st, "k_numerator_on_coset", k_numerator_on_coset, (unsigned)(T / 256), 256, 0, p, n, I, k, A, N, to_m, lo_tab, hi_tab,
               pow_u64(g, (uint64_t)T), T);
    return KZG_OK;
}

static int distribute_powers(kzg_ctx *ctx, hipStream_t st, Fr *d, size_t n, const Fr &g) {
    if (!n) return KZG_OK;
    size_t T = (n + DP_E - 1) / DP_E;
    T = (T + 255) / 256 * 256;
    if (T > DP_TAB * DP_TAB) T = DP_TAB * DP_TAB;  // more elements per thread above 2^22
    const Fr *lo_tab = nullptr, *hi_tab = nullptr;
    KZG_TRY(coset_tables(ctx, st, g, &lo_tab, &hi_tab));
    KZG_LAUNCH(ctx, st, "k_distribute_powers", k_distribute_powers, (unsigned)(T / 256), 256, 0, d, n, lo_tab, hi_tab,
               pow_u64(g, (uint64_t)T), T);
    return KZG_OK;
}

// coset_fft: distribute_powers(g) then fft; icoset_fft: ifft then distribute_powers(g^-1).
// nnz (forward only): the caller knows that d[nnz ..) is zero -- a zero-padded short polynomial -- so only d[0, nnz) is scaled
static int coset_ntt_run(kzg_ctx *ctx, int lane, Fr *d, uint32_t log_n, int inverse, const Fr &g, size_t nnz = (size_t)-1) {
    hipStream_t st = ctx->lanes[lane].stream;
    size_t n = (size_t)1 << log_n;
    const bool unit = g == Fr::one();  // the "coset" 1 * H: the plain transform (create_witness_batched's first choice of a shift)
    if (!inverse) {
        if (!unit) KZG_TRY(distribute_powers(ctx, st, d, nnz < n ? nnz : n, g));
        return ntt_run(ctx, lane, d, log_n, 0, nnz);
    }
    KZG_TRY(ntt_run(ctx, lane, d, log_n, 1));
    return unit ? (int)KZG_OK : distribute_powers(ctx, st, d, n, inv(g));
}

// ---------------------------------------------------------------------------------------------
// interpolation through k points (k <= 16384: the k x k matrix of scaled quotients Z / (X - x_i) is 8.6 GB there)
// ---------------------------------------------------------------------------------------------
// Z = prod_i (X - x_i): z has k+1 coefficients.  One block of 1024 threads, ping-pong in global memory.
__global__ __launch_bounds__(1024) void k_vanishing_poly(const Fr *xs, uint32_t k, Fr *z0, Fr *z1) {
    // z0 = 1
    for (uint32_t j = threadIdx.x; j <= k; j += blockDim.x) {
        z0[j] = (j == 0) ? Fr::one() : Fr::zero();
        z1[j] = Fr::zero();
    }
    __syncthreads();
    Fr *cur = z0, *nxt = z1;
    for (uint32_t i = 0; i < k; i++) {
        Fr x = xs[i];
        // (cur of degree i) * (X - x): nxt[j] = cur[j-1] - x cur[j]
        for (uint32_t j = threadIdx.x; j <= i + 1; j += blockDim.x) {
            Fr lo = (j <= i) ? mul(x, cur[j]) : Fr::zero();
            Fr hi = (j >= 1) ? cur[j - 1] : Fr::zero();
            nxt[j] = sub(hi, lo);
        }
        __threadfence_block();
        __syncthreads();
        Fr *t = cur;
        cur = nxt;
        nxt = t;
    }
    if (cur != z0) {
        for (uint32_t j = threadIdx.x; j <= k; j += blockDim.x) z0[j] = cur[j];
    }
}

// flag |= 1 if some x_i lies on the coset g*H_N, i.e. (x_i / g)^N == 1
__global__ __launch_bounds__(256) void k_any_on_coset(const Fr *xs, size_t k, Fr ginv, uint32_t log_N, int *flag) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    Fr t = mul(xs[i], ginv);
    for (uint32_t s = 0; s < log_N; s++) t = sqr(t);
    if (t == Fr::one()) atomicOr(flag, 1);
}

int vanishing_poly_run(kzg_ctx *ctx, hipStream_t st, const Fr *d_xs_mont, size_t k, Fr *d_z, Fr *d_tmp) {
    KZG_LAUNCH(ctx, st, "k_vanishing_poly", k_vanishing_poly, 1, 1024, 0, d_xs_mont, (uint32_t)k, d_z, d_tmp);
    return KZG_OK;
}

// den[i] = Z'(x_i) = prod_{j != i} (x_i - x_j)
__global__ __launch_bounds__(256) void k_bary_den(const Fr *xs, uint32_t k, Fr *den) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    Fr xi = xs[i];
    Fr acc = Fr::one();
    for (uint32_t j = 0; j < k; j++) {
        if (j == i) continue;
        acc = mul(acc, sub(xi, xs[j]));
    }
    den[i] = acc;
}

// row i of the k x k matrix: c_i * Z/(X - x_i), c_i = y_i * den_inv_i (synthetic division from the top)
__global__ __launch_bounds__(256) void k_bary_rows(const Fr *xs, const Fr *ys, const Fr *den_inv, const Fr *z, uint32_t k,
                                                   Fr *rows) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    Fr xi = xs[i];
    Fr c = mul(ys[i], den_inv[i]);
    Fr carry = z[k];  // leading coefficient (1)
    for (uint32_t j = k; j-- > 0;) {
        rows[(size_t)j * k + i] = mul(c, carry);  // coefficient j of Z/(X - x_i), stored column-major
        carry = add(z[j], mul(xi, carry));
    }
}

// I_j = sum_i rows[j][i]   (ys given: sum_i ys[i] rows[j][i] -- rows = a point set's kept basis matrix)
__global__ __launch_bounds__(256) void k_bary_colsum(const Fr *rows, uint32_t k, Fr *out, const Fr *ys) {
    __shared__ Fr sh[256];
    uint32_t j = blockIdx.x;
    Fr acc = Fr::zero();
    for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) acc = add(acc, ys ? mul(ys[i], rows[(size_t)j * k + i]) : rows[(size_t)j * k + i]);
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] = add(sh[threadIdx.x], sh[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[j] = sh[0];
}

// ---- the same interpolation building blocks with logarithmic depth (k sequential Fr multiplies are ~0.3 ms at k = 256) ----
// out[b] = prod_{j < k, j != skip} (pt_b - xs[j]).  mode 0: pt_b = pts[b], nothing skipped (Z evaluated on a 2^m-th-roots domain);
// mode 1: pt_b = xs[b], j = b skipped (the barycentric denominators Z'(x_b)).  One block per b, tree product through LDS.
__global__ __launch_bounds__(256) void k_prod_diff(const Fr *pts, const Fr *xs, uint32_t k, int mode, Fr *out) {
    __shared__ Fr sh[256];
    const uint32_t b = blockIdx.x;
    const Fr p = mode ? xs[b] : pts[b];
    Fr acc = Fr::one();
    for (uint32_t j = threadIdx.x; j < k; j += blockDim.x)
        if (!(mode && j == b)) acc = mul(acc, sub(p, xs[j]));
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] = mul(sh[threadIdx.x], sh[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[b] = sh[0];
}

__global__ __launch_bounds__(256) void k_inverse_each(const Fr *in, Fr *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i].is_zero() ? Fr::zero() : inv(in[i]);
}

// rows[j * k + i] = c_i q_{i,j}, q_i = Z / (X - x_i) (q_{i,j} = sum_{m > j} z_m x_i^(m-j-1)), c_i = y_i / Z'(x_i): the synthetic
// division of k_bary_rows cut into chunks of CH coefficients.  One block per i, one thread per chunk c of the index range [0, k]:
// local Horner value of the chunk (CH steps), suffix Horner over the chunks above (<= (k+1)/CH steps, X = x_i^CH) = the carry
// entering the chunk, then the chunk's CH quotient coefficients: depth ~3 sqrt(k) instead of k.
__global__ __launch_bounds__(256) void k_bary_rows_chunked(const Fr *xs, const Fr *ys, const Fr *den_inv, const Fr *z, uint32_t k,
                                                           uint32_t CH, uint32_t nch, Fr *rows) {
    __shared__ Fr B[256];
    const uint32_t i = blockIdx.x, c = threadIdx.x;
    const Fr xi = xs[i];
    const Fr ci = ys ? mul(ys[i], den_inv[i]) : den_inv[i];  // (no ys: the basis matrix of a point set, kept by PointSetCache)
    const uint32_t lo = c * CH, hi = lo + CH < k + 1 ? lo + CH : k + 1;  // coefficients z[lo, hi) of this chunk
    Fr bv = Fr::zero();
    if (c < nch)
        for (uint32_t m = hi; m-- > lo;) bv = add(mul(bv, xi), z[m]);
    B[c] = bv;
    __syncthreads();
    if (c >= nch) return;
    const Fr X = pow_u64(xi, (uint64_t)CH);
    Fr carry = Fr::zero();  // U_c = sum_{c' > c} B_c' X^(c'-c-1): the value of the coefficients above this chunk at x_i
    for (uint32_t cc = nch; cc-- > c + 1;) carry = add(mul(carry, X), B[cc]);
    // quotient coefficients j in [lo, hi) /\ [0, k): q_{hi-1} = carry, q_{j-1} = z_j + x_i q_j
    for (uint32_t j = hi; j-- > lo;) {
        if (j < k) rows[(size_t)j * k + i] = mul(ci, carry);
        carry = add(z[j], mul(xi, carry));
    }
}

// flag |= 1 if any of v[0..n) is zero
__global__ __launch_bounds__(256) void k_any_zero(const Fr *v, size_t n, int *flag) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && v[i].is_zero()) atomicOr(flag, 1);
}
// flag |= 2 if any of v[0..n) is non-zero
__global__ __launch_bounds__(256) void k_any_nonzero(const Fr *v, size_t n, int *flag) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !v[i].is_zero()) atomicOr(flag, 2);
}

// flag |= 2 if a and b differ anywhere
__global__ __launch_bounds__(256) void k_any_diff(const Fr *a, const Fr *b, size_t n, int *flag) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && a[i] != b[i]) atomicOr(flag, 2);
}

// a[i] *= cinv[i]
__global__ __launch_bounds__(256) void k_mul_inplace(Fr *a, const Fr *cinv, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = mul(a[i], cinv[i]);
}
// a[j] -= b[j], j < k   (p - I in coefficient form: I has k coefficients)
__global__ __launch_bounds__(256) void k_sub_prefix(Fr *a, const Fr *b, size_t k) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < k) a[i] = sub(a[i], b[i]);
}

// dst[0..n) = src[0..m) zero-extended, converted to Montgomery form if `to_m`
__global__ __launch_bounds__(256) void k_load_padded(const Fr *src, size_t m, Fr *dst, size_t n, int to_m) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr v = i < m ? src[i] : Fr::zero();
    dst[i] = (to_m && i < m) ? to_mont(v) : v;
}

// p'[0..n2): p' = p - (X + (y - x)) for the k == 1 quirk (src/polynomial.rs:244-247)
__global__ __launch_bounds__(256) void k_sub_linear(const Fr *src, size_t n, Fr *dst, size_t n2, Fr c0) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    Fr v = i < n ? src[i] : Fr::zero();
    if (i == 0) v = sub(v, c0);
    if (i == 1) v = sub(v, Fr::one());
    dst[i] = v;
}

static inline unsigned gridfor(size_t n, unsigned b = 256) { return (unsigned)((n + b - 1) / b); }

}  // namespace kzg

using namespace kzg;

static int hscalar(kzg_ctx *ctx, const void *s, int sfmt, Fr *mont) {
    Fr v;
    memcpy(v.v, s, 32);
    if (sfmt == KZG_FR_CANONICAL_LE_32) {
        if (!is_canonical(v)) return fail(ctx, KZG_ERR_SHAPE, "scalar not canonical (>= r)");
        v = to_mont(v);
    } else if (sfmt != KZG_FR_MONT_LE_32) {
        return fail(ctx, KZG_ERR_SHAPE, "unknown scalar format");
    }
    *mont = v;
    return KZG_OK;
}

// a[i] *= b[i]  (Montgomery product; with b in Montgomery form the result keeps a's form)
__global__ __launch_bounds__(256) void k_mul_assign(Fr *a, const Fr *b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = mul(a[i], b[i]);
}

extern "C" int kzg_poly_mul(kzg_ctx *ctx, const void *a, size_t na, const void *b, size_t nb, int sfmt, int flags, void *out) {
    // Polynomial::fft_mul (src/polynomial.rs:167-183)
    if (!ctx || !a || !b || !out || na == 0 || nb == 0) return KZG_ERR_SHAPE;
    kzg::Lease ls;
    KZG_TRY(lease_lane(ctx, &ls));
    const int lane = ls.lane;
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (sfmt != KZG_FR_MONT_LE_32 && sfmt != KZG_FR_CANONICAL_LE_32) return fail(ctx, KZG_ERR_SHAPE, "unknown scalar format");
    uint32_t log_n = (uint32_t)ilog2_ceil(na + nb);   // from_coeffs(resize(n + k)) rounds up to 2^exp
    if (log_n >= FR_TWO_ADICITY) return fail(ctx, KZG_ERR_DEGREE_TOO_LARGE, "polynomial degree too large");
    if (log_n > 27) return fail(ctx, KZG_ERR_SHAPE, "kzg_poly_mul: products of up to 2^27 coefficients");
    size_t N = (size_t)1 << log_n, nout = na + nb - 1;
    KZG_TRY(lane_reserve(ctx, lane, 4 * N * 32 + 3 * ntt_workspace_bytes(log_n) + (1 << 20)));
    hipStream_t st = ctx->lanes[lane].stream;
    Fr *A = (Fr *)lane_alloc(ctx, lane, N * 32), *Bv = (Fr *)lane_alloc(ctx, lane, N * 32), *ia = (Fr *)lane_alloc(ctx, lane, N * 32),
       *ib = (Fr *)lane_alloc(ctx, lane, N * 32);
    if (!A || !Bv || !ia || !ib) return fail(ctx, KZG_ERR_ALLOC, "workspace");
    hipMemcpyKind kind = (flags & KZG_IN_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(ia, a, na * 32, kind, st));
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(ib, b, nb * 32, kind, st));
    // a keeps its form; b goes to Montgomery form so that the pointwise Montgomery product preserves a's form
    // a short operand (a low-degree factor times a long polynomial) takes the transform's short-input path: no zero padding, no
    // column pass
    const bool sa = ntt_short_input_ok(log_n, na), sb = ntt_short_input_ok(log_n, nb);
    KZG_LAUNCH(ctx, st, "k_load_padded", k_load_padded, gridfor(sa ? na : N), 256, 0, ia, na, A, sa ? na : N, 0);
    KZG_LAUNCH(ctx, st, "k_load_padded", k_load_padded, gridfor(sb ? nb : N), 256, 0, ib, nb, Bv, sb ? nb : N, sfmt == KZG_FR_CANONICAL_LE_32);
    KZG_TRY(ntt_run(ctx, lane, A, log_n, 0, sa ? na : (size_t)-1));
    KZG_TRY(ntt_run(ctx, lane, Bv, log_n, 0, sb ? nb : (size_t)-1));
    KZG_LAUNCH(ctx, st, "k_mul_assign", k_mul_assign, gridfor(N), 256, 0, A, Bv, N);
    KZG_TRY(ntt_run(ctx, lane, A, log_n, 1));
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(out, A, nout * 32, (flags & KZG_OUT_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
    KZG_HIP_CHECK(ctx, hipStreamSynchronize(st));
    if (ctx->prof) prof_collect(ctx);
    return KZG_OK;
}

extern "C" int kzg_coset_ntt_fr(kzg_ctx *ctx, void *data, uint32_t log_n, int inverse, int sfmt, int flags) {
    if (!ctx || !data) return KZG_ERR_SHAPE;
    kzg::Lease ls;
    KZG_TRY(lease_lane(ctx, &ls));
    const int lane = ls.lane;
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    (void)sfmt;  // linear map with Montgomery constants: the data's form is preserved
    if (log_n >= FR_TWO_ADICITY) return fail(ctx, KZG_ERR_DEGREE_TOO_LARGE, "polynomial degree too large");
    if (log_n > 28) return fail(ctx, KZG_ERR_SHAPE, "coset NTT sizes above 2^28 are not supported (as kzg_ntt_fr)");
    size_t n = (size_t)1 << log_n;
    KZG_TRY(lane_reserve(ctx, lane, ((flags & KZG_IN_DEVICE) ? 0 : n * 32) + ntt_workspace_bytes(log_n) + 65536));
    hipStream_t st = ctx->lanes[lane].stream;
    Fr *d = (flags & KZG_IN_DEVICE) ? (Fr *)data : (Fr *)lane_alloc(ctx, lane, n * 32);
    if (!d) return fail(ctx, KZG_ERR_ALLOC, "workspace");
    if (!(flags & KZG_IN_DEVICE)) KZG_HIP_CHECK(ctx, hipMemcpyAsync(d, data, n * 32, hipMemcpyHostToDevice, st));
    KZG_TRY(coset_ntt_run(ctx, lane, d, log_n, inverse, from_u64<FrParams>(FR_MULT_GENERATOR)));
    if (!(flags & KZG_IN_DEVICE)) KZG_HIP_CHECK(ctx, hipMemcpyAsync(data, d, n * 32, hipMemcpyDeviceToHost, st));
    KZG_HIP_CHECK(ctx, hipStreamSynchronize(st));
    if (ctx->prof) prof_collect(ctx);
    return KZG_OK;
}

extern "C" int kzg_witness_coeff_batched(kzg_ctx *ctx, const kzg_srs *srs, const void *coeffs, size_t n, const void *xs,
                                         const void *ys, size_t k, int sfmt, int flags, void *out_w, int ofmt,
                                         void *out_r, size_t *out_r_len) {
    if (!ctx || !srs || !coeffs || !xs || !ys || !out_w || !out_r || !out_r_len || n == 0) return KZG_ERR_SHAPE;
    WitnessSink sink{srs, 0, srs->n, srs->n, nullptr};
    return witness_coeff_batched_run(ctx, sink, coeffs, n, xs, ys, k, sfmt, flags, out_w, ofmt, out_r, out_r_len);
}

// ---------------------------------------------------------------------------------------------
// Opening-point sets.  Everything create_witness_batched derives from the opening POINTS alone -- Z = prod (X - x_i), the
// barycentric weights 1 / Z'(x_i), the coset shift on which Z has no root, and 1 / Z on that coset (N values: the batch inversion of
// the division) -- is the same for every polynomial opened at those points, which is how batched openings are used (one point set,
// many polynomials).  A call that finds its point set here skips the coset test with its host round trip, the two product trees,
// the small iNTT, the k inversions, Z's transform and the batch inversion: ~0.35 of the ~1.1 ms a k = 256 opening at 2^20 costs
// over a commit (profiles/r05_prof_witness.txt).  Entries are published by the call that built them after its stream is
// synchronised, are immutable, and are pinned (users) while a call reads them; eviction is LRU over unpinned slots.
// ---------------------------------------------------------------------------------------------
namespace kzg {
struct PointSetEntry {
    std::vector<uint8_t> key;  // the k x 32 bytes of xs as the caller passed them
    int sfmt = 0;
    uint32_t log_N = 0;
    size_t k = 0;
    Fr gsh;                    // the coset shift
    uint8_t *slot = nullptr;   // dx[k] | z0[k + 1] | deni[k] | zinv[N] | basis[k x k] (k <= POINT_SET_BASIS_MAX)   (Montgomery form)
    size_t N = 0;
    int users = 0;
    uint64_t stamp = 0;
    bool valid = false;
    Fr *dx() const { return (Fr *)slot; }
    Fr *z0() const { return dx() + k; }
    Fr *deni() const { return z0() + k + 1; }
    Fr *zinv() const { return deni() + k; }
    Fr *basis() const { return zinv() + N; }  // rows[j * k + i] = coefficient j of L_i = Z / ((X - x_i) Z'(x_i)): I = sum_i y_i L_i
};
constexpr size_t POINT_SET_BASIS_MAX = 1024;  // 32 MB of basis matrix at most
struct PointSetCache {
    uint8_t *pool = nullptr;
    size_t slot_bytes = 0;
    std::vector<PointSetEntry> slots;
    uint64_t clock = 0, hits = 0, misses = 0;
};
static size_t point_set_bytes(size_t k, size_t N) {
    return ((3 * k + 1 + N + (k <= POINT_SET_BASIS_MAX ? k * k : 0)) * 32 + 255) & ~(size_t)255;
}

// a published entry for (xs, sfmt, log_N), pinned; nullptr on a miss
static PointSetEntry *point_set_lookup(kzg_ctx *ctx, const void *xs, size_t k, int sfmt, uint32_t log_N) {
    std::lock_guard<std::mutex> lk(ctx->cache_mu);
    PointSetCache *c = ctx->point_sets;
    if (!c) return nullptr;
    for (auto &e : c->slots)
        if (e.valid && e.k == k && e.sfmt == sfmt && e.log_N == log_N && !memcmp(e.key.data(), xs, k * 32)) {
            e.users++;
            e.stamp = ++c->clock;
            c->hits++;
            return &e;
        }
    c->misses++;
    return nullptr;
}
static void point_set_release(kzg_ctx *ctx, PointSetEntry *e) {
    if (!e) return;
    std::lock_guard<std::mutex> lk(ctx->cache_mu);
    e->users--;
}
// A slot for a new entry (pinned, not yet valid), or nullptr: cache off, every slot pinned, or the pool cannot be allocated.  The pool
// comes into being here, sized for the first entry (`slots` equal slots), and is rebuilt with larger slots when a larger entry comes.
static PointSetEntry *point_set_reserve(kzg_ctx *ctx, size_t k, size_t N) {
    std::lock_guard<std::mutex> lk(ctx->cache_mu);
    if (ctx->opt_witness_cache_slots <= 0) return nullptr;
    PointSetCache *c = ctx->point_sets;
    const size_t need = point_set_bytes(k, N);
    if (!c) {
        if (need * (size_t)ctx->opt_witness_cache_slots > ((size_t)8 << 30)) return nullptr;  // (2^24-coefficient polynomials: not cached)
        c = new PointSetCache();
        if (hipMalloc((void **)&c->pool, need * (size_t)ctx->opt_witness_cache_slots) != hipSuccess) {
            hipGetLastError();
            delete c;
            ctx->opt_witness_cache_slots = 0;
            return nullptr;
        }
        c->slot_bytes = need;
        c->slots.resize(ctx->opt_witness_cache_slots);
        for (size_t i = 0; i < c->slots.size(); i++) c->slots[i].slot = c->pool + i * need;
        ctx->point_sets = c;
    }
    if (need > c->slot_bytes) {
        // a larger point set than the pool was sized for (a context that opened small polynomials first): the pool is rebuilt with
        // larger slots -- once per growth, and only while nobody reads an entry (hipFree waits for the device: a one-time stall)
        for (auto &e : c->slots)
            if (e.users) return nullptr;
        if (need * c->slots.size() > ((size_t)8 << 30)) return nullptr;
        uint8_t *np = nullptr;
        if (hipMalloc((void **)&np, need * c->slots.size()) != hipSuccess) {
            hipGetLastError();
            return nullptr;
        }
        hipFree(c->pool);
        c->pool = np;
        c->slot_bytes = need;
        for (size_t i = 0; i < c->slots.size(); i++) {
            c->slots[i] = PointSetEntry();
            c->slots[i].slot = c->pool + i * need;
        }
    }
    PointSetEntry *best = nullptr;
    for (auto &e : c->slots) {
        if (e.users) continue;
        if (!e.valid) {
            best = &e;
            break;
        }
        if (!best || e.stamp < best->stamp) best = &e;
    }
    if (!best) return nullptr;
    best->valid = false;
    best->users = 1;
    best->k = k;  // the slot's layout (dx | z0 | deni | zinv | basis) is the new occupant's from here on
    best->N = N;
    return best;
}
static void point_set_publish(kzg_ctx *ctx, PointSetEntry *e, const void *xs, size_t k, int sfmt, uint32_t log_N, const Fr &gsh, bool ok) {
    std::lock_guard<std::mutex> lk(ctx->cache_mu);
    e->users--;
    if (!ok) return;
    e->key.assign((const uint8_t *)xs, (const uint8_t *)xs + k * 32);
    e->k = k;
    e->sfmt = sfmt;
    e->log_N = log_N;
    e->gsh = gsh;
    e->stamp = ++ctx->point_sets->clock;
    e->valid = true;
}
void point_set_stats(kzg_ctx *ctx, uint64_t *hits, uint64_t *misses) {
    std::lock_guard<std::mutex> lk(ctx->cache_mu);
    *hits = ctx->point_sets ? ctx->point_sets->hits : 0;
    *misses = ctx->point_sets ? ctx->point_sets->misses : 0;
}
void point_sets_free(kzg_ctx *ctx) {
    if (!ctx->point_sets) return;
    if (ctx->point_sets->pool) hipFree(ctx->point_sets->pool);
    delete ctx->point_sets;
    ctx->point_sets = nullptr;
}
}  // namespace kzg

// KZGProver::create_witness_batched (src/coeff_form.rs:83-111).  `sink` says where the quotient MSM goes: the whole SRS of this
// GPU and the witness to the caller (above), or one shard [first, first + len) of a group's SRS and the 144-byte partial into
// the group's exchange buffer (mgpu.hip: the quotient is replicated on every GPU, the MSM is sharded -- SURVEY 8e).
int kzg::witness_coeff_batched_run(kzg_ctx *ctx, const WitnessSink &sink, const void *coeffs, size_t n, const void *xs,
                                   const void *ys, size_t k, int sfmt, int flags, void *out_w, int ofmt, void *out_r,
                                   size_t *out_r_len) {
    // a leased lane, like commit / create_witness: interpolation, the coset-NTT division and the quotient MSM on the lane's stream
    // and arena, the accumulation kernel on the shared FIFO streams once other calls are in flight (KZGProver is Clone + &self:
    // N threads call create_witness_batched at once, src/coeff_form.rs:83-111)
    kzg::Lease ls;
    KZG_TRY(lease_lane(ctx, &ls));
    const int lane = ls.lane;
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const kzg_srs *srs = sink.srs;
    auto sink_len = [&](size_t cnt) { return cnt <= sink.first ? (size_t)0 : (cnt - sink.first < sink.len ? cnt - sink.first : sink.len); };
    auto sink_msm = [&](Fr *q, size_t cnt, MsmPoint **res) {
        const size_t len = sink_len(cnt);
        return lease_msm(ctx, ls, srs, 0, len ? q + sink.first : q, len, KZG_FR_MONT_LE_32, res);
    };
    auto sink_finish = [&](const MsmPoint *res) {
        if (!sink.d_partial) return finish_point(ctx, lane, res, out_w, ofmt, flags);
        KZG_TRY(emit_point(ctx, lane, res, sink.d_partial, KZG_G1_JACOBIAN_MONT_144));
        KZG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->lanes[lane].stream));
        return (int)KZG_OK;
    };
    if (sfmt != KZG_FR_MONT_LE_32 && sfmt != KZG_FR_CANONICAL_LE_32) return fail(ctx, KZG_ERR_SHAPE, "unknown scalar format");
    if (!point_format_bytes(ofmt)) return fail(ctx, KZG_ERR_SHAPE, "unknown G1 output format");
    if (k == 0) return fail(ctx, KZG_ERR_SHAPE, "no opening points (the reference recurses without bound on an empty slice)");
    if (k > 16384) return fail(ctx, KZG_ERR_SHAPE, "create_witness_batched supports at most 16384 opening points (the interpolation works on a k x k matrix)");
    const int to_m = sfmt == KZG_FR_CANONICAL_LE_32;
    hipStream_t st = ctx->lanes[lane].stream;

    // ---- k == 1: the reference's interpolant is X + (y - x) (src/polynomial.rs:244-247) -----------
    if (k == 1) {
        Fr xm, ym;
        KZG_TRY(hscalar(ctx, xs, sfmt, &xm));
        KZG_TRY(hscalar(ctx, ys, sfmt, &ym));
        size_t n2 = n < 2 ? 2 : n;
        if (n2 - 1 > sink.total) return fail(ctx, KZG_ERR_SHAPE, "quotient longer than the SRS (reference: slice index panic)");
        KZG_TRY(lane_reserve(ctx, lane, msm_workspace_bytes(srs, sink_len(n2 - 1)) + 4 * n2 * 32 + (n2 / 2048 + 8) * 64 + 65536));
        Fr *p = (Fr *)lane_alloc(ctx, lane, n2 * 32), *pin = (Fr *)lane_alloc(ctx, lane, n2 * 32), *q = (Fr *)lane_alloc(ctx, lane, n2 * 32);
        Fr *dpx = (Fr *)lane_alloc(ctx, lane, 256);
        if (!p || !pin || !q || !dpx) return fail(ctx, KZG_ERR_ALLOC, "workspace");
        KZG_HIP_CHECK(ctx, hipMemcpyAsync(pin, coeffs, n * 32, (flags & KZG_IN_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
        KZG_LAUNCH(ctx, st, "k_load_padded", k_load_padded, gridfor(n2), 256, 0, pin, n, p, n2, to_m);
        KZG_LAUNCH(ctx, st, "k_sub_linear", k_sub_linear, gridfor(n2), 256, 0, p, n2, p, n2, sub(ym, xm));
        KZG_TRY(quotient_linear_run(ctx, lane, p, n2, xm, q, dpx));
        // (to the lane's pinned staging buffer: a copy into pageable host memory would make the host wait for the stream here, before
        // the MSM is enqueued -- see kzg_witness_coeff)
        KZG_TRY(lane_pinned(ctx, lane, 4096));
        KZG_HIP_CHECK(ctx, hipMemcpyAsync(ctx->lanes[lane].pinned + 1024, dpx, 32, hipMemcpyDeviceToHost, st));
        MsmPoint *res = nullptr;
        KZG_TRY(sink_msm(q, n2 - 1, &res));
        KZG_TRY(sink_finish(res));
        Fr px;
        memcpy(px.v, ctx->lanes[lane].pinned + 1024, 32);
        if (ctx->prof) prof_collect(ctx);
        if (!px.is_zero()) return fail(ctx, KZG_ERR_POINT_NOT_ON_POLY, "point not on polynomial!");
        Fr r0 = sub(ym, xm), r1 = Fr::one();
        if (to_m) {
            r0 = from_mont(r0);
            r1 = from_mont(r1);
        }
        memcpy(out_r, r0.v, 32);
        memcpy((uint8_t *)out_r + 32, r1.v, 32);
        *out_r_len = 2;
        return KZG_OK;
    }

    // ---- k >= 2 -----------------------------------------------------------------------------------
    const bool small_poly = k >= n;  // deg I = k-1 >= deg p: quotient is zero iff I == p
    uint32_t log_N = (uint32_t)ilog2_ceil(n > k + 1 ? n : k + 1);
    if (log_N >= FR_TWO_ADICITY) return fail(ctx, KZG_ERR_DEGREE_TOO_LARGE, "polynomial degree too large");
    if (log_N > 26) return fail(ctx, KZG_ERR_SHAPE, "create_witness_batched: polynomials of up to 2^26 coefficients (the SRS with its window tables is 118 GB there)");
    size_t N = (size_t)1 << log_N;
    size_t nq = small_poly ? 0 : n - k;
    if (nq > sink.total) return fail(ctx, KZG_ERR_SHAPE, "quotient longer than the SRS (reference: slice index panic)");
    size_t need = msm_workspace_bytes(srs, sink_len(nq) ? sink_len(nq) : 1) + 5 * N * 32 + 4 * ntt_workspace_bytes(log_N) + (size_t)k * k * 32 +
                  64 * (k + 2) * 32 + (4 << 20);
    KZG_TRY(lane_reserve(ctx, lane, need));
    Fr *dx = (Fr *)lane_alloc(ctx, lane, k * 32), *dy = (Fr *)lane_alloc(ctx, lane, k * 32);
    Fr *z0 = (Fr *)lane_alloc(ctx, lane, (k + 1) * 32), *z1 = (Fr *)lane_alloc(ctx, lane, (k + 1) * 32);
    Fr *den = (Fr *)lane_alloc(ctx, lane, k * 32), *deni = (Fr *)lane_alloc(ctx, lane, k * 32);
    Fr *rows = (Fr *)lane_alloc(ctx, lane, (size_t)k * k * 32), *I = (Fr *)lane_alloc(ctx, lane, k * 32);
    Fr *A = (Fr *)lane_alloc(ctx, lane, N * 32), *Bv = (Fr *)lane_alloc(ctx, lane, N * 32), *Cv = (Fr *)lane_alloc(ctx, lane, N * 32);
    Fr *Ci = (Fr *)lane_alloc(ctx, lane, N * 32), *pin = (Fr *)lane_alloc(ctx, lane, N * 32);
    int *flag = (int *)lane_alloc(ctx, lane, 256);
    if (!dx || !dy || !z0 || !z1 || !den || !deni || !rows || !I || !A || !Bv || !Cv || !Ci || !pin || !flag)
        return fail(ctx, KZG_ERR_ALLOC, "workspace");
    KZG_HIP_CHECK(ctx, hipMemsetAsync(flag, 0, sizeof(int), st));
    // the point set: known (everything that depends on xs alone is taken from the cache), or new and to be kept, or neither
    struct PointSetGuard {
        kzg_ctx *ctx;
        PointSetEntry *hit = nullptr, *fill = nullptr;
        const void *xs = nullptr;
        size_t k = 0;
        int sfmt = 0;
        uint32_t log_N = 0;
        Fr gsh;
        bool ok = false;
        hipStream_t stream = nullptr;
        ~PointSetGuard() {
            if (hit) point_set_release(ctx, hit);
            // An early return (a failed launch, a failed check) leaves `ok` false with copies and kernels that write the reserved slot
            // possibly still in flight on this lane's stream: the slot must not be handed to another lane before they have drained
            // (a second filler's entry could otherwise be overwritten by the stale writes).
            if (fill && !ok && stream) hipStreamSynchronize(stream);
            if (fill) point_set_publish(ctx, fill, xs, k, sfmt, log_N, gsh, ok);
        }
    } ps{ctx};
    ps.stream = st;
    ps.xs = xs, ps.k = k, ps.sfmt = sfmt, ps.log_N = log_N;
    if (!small_poly) {
        ps.hit = point_set_lookup(ctx, xs, k, sfmt, log_N);
        if (!ps.hit) ps.fill = point_set_reserve(ctx, k, N);
    }
    if (ps.hit) {
        dx = ps.hit->dx();
        z0 = ps.hit->z0();
        deni = ps.hit->deni();
    } else {
        KZG_HIP_CHECK(ctx, hipMemcpyAsync(dx, xs, k * 32, hipMemcpyHostToDevice, st));
        if (to_m) KZG_TRY(fr_convert(ctx, st, dx, k, 1));
    }
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(dy, ys, k * 32, hipMemcpyHostToDevice, st));
    if (to_m) KZG_TRY(fr_convert(ctx, st, dy, k, 1));
    // pick a shift g on whose coset g*H Z has no root: g = 1 first -- H itself, no scaling passes at all (two passes over N elements
    // saved whenever no opening point is an N-th root of unity) --, then 7, 7^2, ...  An opening point inside g*H (x = 1 or a power of
    // omega for H, x = 7 for 7*H) would make Z vanish there; the cosets 7^j*H are pairwise distinct, so at most k + 1 candidates fail.
    // This is the one host round trip of the call, so it comes first, while the stream holds nothing but the upload of the points.
    Fr g1 = from_u64<FrParams>(FR_MULT_GENERATOR), gsh = Fr::one();
    if (ps.hit) gsh = ps.hit->gsh;
    if (!small_poly && !ps.hit) {
        int *cflag = (int *)lane_alloc(ctx, lane, 256);
        if (!cflag) return fail(ctx, KZG_ERR_ALLOC, "workspace");
        for (size_t attempt = 0;; attempt++) {
            int on = 0;
            KZG_HIP_CHECK(ctx, hipMemsetAsync(cflag, 0, sizeof(int), st));
            KZG_LAUNCH(ctx, st, "k_any_on_coset", k_any_on_coset, gridfor(k), 256, 0, dx, k, inv(gsh), log_N, cflag);
            KZG_HIP_CHECK(ctx, hipMemcpyAsync(&on, cflag, sizeof(int), hipMemcpyDeviceToHost, st));
            KZG_HIP_CHECK(ctx, hipStreamSynchronize(st));
            if (!on) break;
            if (attempt > k + 1) return fail(ctx, KZG_ERR_INTERNAL, "no usable coset shift");
            gsh = mul(gsh, g1);
        }
    }
    // interpolant
    // Z = prod (X - x_i): its values on the M-th roots of unity (M = 2^m > k), one tree product per value, then one small iNTT;
    // Z'(x_i) the same way; the interpolant by chunked synthetic divisions.  Everything here is log- or sqrt-depth in k.
    if (!ps.hit) {
        const uint32_t log_M = (uint32_t)ilog2_ceil(k + 1);
        const size_t Mz = (size_t)1 << log_M;
        Fr *zev = (Fr *)lane_alloc(ctx, lane, Mz * 32), *wpow = (Fr *)lane_alloc(ctx, lane, Mz * 32);
        if (!zev || !wpow) return fail(ctx, KZG_ERR_ALLOC, "workspace");
        KZG_TRY(pow_table(ctx, st, host_omega(log_M), Fr::one(), Mz, wpow));
        KZG_LAUNCH(ctx, st, "k_prod_diff", k_prod_diff, (unsigned)Mz, 256, 0, wpow, dx, (uint32_t)k, 0, zev);
        KZG_TRY(ntt_run(ctx, lane, zev, log_M, 1));
        KZG_HIP_CHECK(ctx, hipMemcpyAsync(z0, zev, (k + 1) * 32, hipMemcpyDeviceToDevice, st));
    }
    if (!ps.hit) {
        KZG_LAUNCH(ctx, st, "k_prod_diff", k_prod_diff, (unsigned)k, 256, 0, dx, dx, (uint32_t)k, 1, den);
        KZG_LAUNCH(ctx, st, "k_any_zero", k_any_zero, gridfor(k), 256, 0, den, k, flag);  // duplicate x_i
        KZG_LAUNCH(ctx, st, "k_inverse_each", k_inverse_each, gridfor(k), 256, 0, den, deni, k);
        if (ps.fill) {  // (a point set with duplicates fails the call and is not published)
            KZG_HIP_CHECK(ctx, hipMemcpyAsync(ps.fill->dx(), dx, k * 32, hipMemcpyDeviceToDevice, st));
            KZG_HIP_CHECK(ctx, hipMemcpyAsync(ps.fill->z0(), z0, (k + 1) * 32, hipMemcpyDeviceToDevice, st));
            KZG_HIP_CHECK(ctx, hipMemcpyAsync(ps.fill->deni(), deni, k * 32, hipMemcpyDeviceToDevice, st));
        }
    }
    if (ps.hit && k <= POINT_SET_BASIS_MAX) {
        KZG_LAUNCH(ctx, st, "k_bary_colsum", k_bary_colsum, (unsigned)k, 256, 0, ps.hit->basis(), (uint32_t)k, I, dy);  // I = sum_i y_i L_i
    } else {
        uint32_t CH = 1;
        while ((uint64_t)CH * CH * 4 <= k) CH *= 2;             // CH = 2^floor(log2(k) / 2): 16 at k = 256, 64 at 4096
        const uint32_t nch = (uint32_t)((k + 1 + CH - 1) / CH);  // <= 256 for k <= 16384 (k = 16383: CH = 64, nch = 256)
        if (nch > 256) return fail(ctx, KZG_ERR_INTERNAL, "interpolation chunk count exceeds the kernel's LDS array (B[256])");
        const unsigned bt = nch <= 64 ? 64 : (nch <= 128 ? 128 : 256);
        KZG_LAUNCH(ctx, st, "k_bary_rows", k_bary_rows_chunked, (unsigned)k, bt, 0, dx, dy, deni, z0, (uint32_t)k, CH, nch, rows);
        KZG_LAUNCH(ctx, st, "k_bary_colsum", k_bary_colsum, (unsigned)k, 256, 0, rows, (uint32_t)k, I, (const Fr *)nullptr);
        if (ps.fill && k <= POINT_SET_BASIS_MAX)  // the basis matrix of the point set, for the calls that find it
            KZG_LAUNCH(ctx, st, "k_bary_rows", k_bary_rows_chunked, (unsigned)k, bt, 0, dx, (const Fr *)nullptr, deni, z0, (uint32_t)k, CH, nch,
                       ps.fill->basis());
    }
    // numerator and divisor on the coset g*H
    const Fr *psrc = (const Fr *)coeffs;  // device-resident coefficients are read where they are
    if (!(flags & KZG_IN_DEVICE)) {
        KZG_HIP_CHECK(ctx, hipMemcpyAsync(pin, coeffs, n * 32, hipMemcpyHostToDevice, st));
        psrc = pin;
    }
    MsmPoint *res = nullptr;
    int hflag = 0;
    if (small_poly) {
        KZG_LAUNCH(ctx, st, "k_load_padded", k_load_padded, gridfor(N), 256, 0, psrc, n, A, N, to_m);
        KZG_LAUNCH(ctx, st, "k_load_padded", k_load_padded, gridfor(N), 256, 0, I, k, Bv, N, 0);
        // deg I <= k-1 and deg p <= n-1 <= k-1: the quotient is zero and the division is exact iff p == I
        KZG_LAUNCH(ctx, st, "k_any_diff", k_any_diff, gridfor(N), 256, 0, A, Bv, N, flag);
        KZG_TRY(sink_msm(A, 0, &res));  // identity
    } else {
        // Z on the coset: its k + 1 coefficients times g^j, then a transform whose input is short -- when the k + 1 values fit the
        // first row of the four-step matrix the column pass is skipped and nothing beyond them is read (no zero padding)
        if (!ps.hit) {
            if (ntt_short_input_ok(log_N, k + 1)) KZG_HIP_CHECK(ctx, hipMemcpyAsync(Cv, z0, (k + 1) * 32, hipMemcpyDeviceToDevice, st));
            else KZG_LAUNCH(ctx, st, "k_load_padded", k_load_padded, gridfor(N), 256, 0, z0, k + 1, Cv, N, 0);
            KZG_TRY(coset_ntt_run(ctx, lane, Cv, log_N, 0, gsh, k + 1));
        }
        // (p - I) g^i in one pass over p, then its transform; the pointwise division by Z's values (one batch inversion; a zero
        // among them would be an opening point on the coset, excluded above) inside the inversion's second sweep -- or, for a known
        // point set, one multiplication by the kept 1 / Z
        KZG_TRY(numerator_on_coset(ctx, st, psrc, n, I, k, A, N, to_m, gsh));
        KZG_TRY(ntt_run(ctx, lane, A, log_N, 0));
        if (ps.hit) {
            KZG_LAUNCH(ctx, st, "k_mul_inplace", k_mul_inplace, gridfor(N), 256, 0, A, ps.hit->zinv(), N);
        } else if (ps.fill) {
            KZG_TRY(batch_inverse(ctx, st, Cv, ps.fill->zinv(), N));  // the inverses are kept: one more pass than the fused division
            KZG_LAUNCH(ctx, st, "k_mul_inplace", k_mul_inplace, gridfor(N), 256, 0, A, ps.fill->zinv(), N);
        } else {
            KZG_TRY(batch_inverse_mul(ctx, st, Cv, A, Ci, N, flag));
        }
        KZG_TRY(coset_ntt_run(ctx, lane, A, log_N, 1, gsh));
        // exact division <=> deg q <= N-1-k <=> the top k coefficients vanish
        KZG_LAUNCH(ctx, st, "k_any_nonzero", k_any_nonzero, gridfor(k), 256, 0, A + (N - k), k, flag);
        KZG_TRY(sink_msm(A, nq, &res));
    }
    // flag and interpolant leave through the lane's pinned staging buffer (flag at +1024, coefficients from +4096): copies into the
    // caller's pageable memory would block the host until the MSM in front of them has finished, and the conversion kernel and the
    // second copy would only be enqueued then
    KZG_TRY(lane_pinned(ctx, lane, 8192 + k * 32));
    char *pin_host = ctx->lanes[lane].pinned;
    if (to_m) KZG_TRY(fr_convert(ctx, st, I, k, 0));
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(pin_host + 1024, flag, sizeof(int), hipMemcpyDeviceToHost, st));
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(pin_host + 4096, I, k * 32, hipMemcpyDeviceToHost, st));
    KZG_TRY(sink_finish(res));
    memcpy(&hflag, pin_host + 1024, sizeof(int));
    memcpy(out_r, pin_host + 4096, k * 32);
    ps.gsh = gsh;
    ps.ok = !(hflag & 1);  // the stream is synchronised: the slot's contents are complete (a wrong y leaves the point set usable)
    if (ctx->prof) prof_collect(ctx);
    if (hflag & 1) {
        // a zero denominator: duplicate opening points (the reference unwrap()s an invert() of zero)
        return fail(ctx, KZG_ERR_SHAPE, "duplicate opening points (reference: invert().unwrap() panic)");
    }
    if (hflag & 2) return fail(ctx, KZG_ERR_POINT_NOT_ON_POLY, "point not on polynomial!");
    *out_r_len = k;
    return KZG_OK;
}

// ---------------------------------------------------------------------------------------------
// the remaining EvaluationDomain operations (src/ft.rs:180-271): z, divide_by_z_on_coset, mul_assign, sub_assign
// ---------------------------------------------------------------------------------------------
namespace kzg {
__global__ __launch_bounds__(256) void k_scale_fr(Fr *a, Fr s, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = mul(a[i], s);
}
// a[i] = a[i] * b[i] (op 0) or a[i] - b[i] (op 1).  Montgomery form only when multiplying: a product of two canonical values
// would carry a stray 2^-256, so canonical operands are converted on the fly
__global__ __launch_bounds__(256) void k_vec_op(Fr *a, const Fr *b, size_t n, int op, int canonical) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (op == 1) {
        a[i] = sub(a[i], b[i]);
    } else {
        a[i] = canonical ? mul(a[i], to_mont(b[i])) : mul(a[i], b[i]);
    }
}
}  // namespace kzg

extern "C" int kzg_domain_z(size_t d, const void *tau, int sfmt, void *out) {
    // EvaluationDomain::z (src/ft.rs:182-187): tau^d - 1.  Host-only.
    if (!tau || !out) return KZG_ERR_SHAPE;
    Fr t;
    memcpy(t.v, tau, 32);
    if (sfmt == KZG_FR_CANONICAL_LE_32) {
        if (!is_canonical(t)) return KZG_ERR_SHAPE;
        t = to_mont(t);
    } else if (sfmt != KZG_FR_MONT_LE_32) {
        return KZG_ERR_SHAPE;
    }
    Fr z = sub(pow_u64(t, (uint64_t)d), Fr::one());
    if (sfmt == KZG_FR_CANONICAL_LE_32) z = from_mont(z);
    memcpy(out, z.v, 32);
    return KZG_OK;
}

static int vec_entry(kzg_ctx *ctx, void *a, const void *b, size_t n, int sfmt, int flags, int op, const Fr *scale) {
    kzg::Lease ls;
    KZG_TRY(lease_lane(ctx, &ls));
    const int lane = ls.lane;
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (sfmt != KZG_FR_MONT_LE_32 && sfmt != KZG_FR_CANONICAL_LE_32) return fail(ctx, KZG_ERR_SHAPE, "unknown scalar format");
    if (n > ((size_t)1 << 40)) return fail(ctx, KZG_ERR_SHAPE, "vector too long");
    if (n == 0) return KZG_OK;
    hipStream_t st = ctx->lanes[lane].stream;
    const bool dev = (flags & KZG_IN_DEVICE) != 0;
    KZG_TRY(lane_reserve(ctx, lane, dev ? 4096 : 2 * n * 32 + 8192));
    Fr *da = (Fr *)a, *db = (Fr *)b;
    if (!dev) {
        da = (Fr *)lane_alloc(ctx, lane, n * 32);
        db = b ? (Fr *)lane_alloc(ctx, lane, n * 32) : nullptr;
        if (!da || (b && !db)) return fail(ctx, KZG_ERR_ALLOC, "workspace");
        KZG_HIP_CHECK(ctx, hipMemcpyAsync(da, a, n * 32, hipMemcpyHostToDevice, st));
        if (b) KZG_HIP_CHECK(ctx, hipMemcpyAsync(db, b, n * 32, hipMemcpyHostToDevice, st));
    }
    if (scale) KZG_LAUNCH(ctx, st, "k_scale_fr", k_scale_fr, gridfor(n), 256, 0, da, *scale, n);
    else KZG_LAUNCH(ctx, st, "k_vec_op", k_vec_op, gridfor(n), 256, 0, da, db, n, op, sfmt == KZG_FR_CANONICAL_LE_32);
    if (!dev) KZG_HIP_CHECK(ctx, hipMemcpyAsync(a, da, n * 32, hipMemcpyDeviceToHost, st));
    KZG_HIP_CHECK(ctx, hipStreamSynchronize(st));
    if (ctx->prof) prof_collect(ctx);
    return KZG_OK;
}

extern "C" int kzg_divide_by_z_on_coset(kzg_ctx *ctx, void *data, uint32_t log_n, int sfmt, int flags) {
    // EvaluationDomain::divide_by_z_on_coset (src/ft.rs:192-217): every value times 1 / (g^d - 1), g = 7, d = 2^log_n.
    // A multiplication by a Montgomery-form constant preserves whichever form the data is in.
    if (!ctx || !data) return KZG_ERR_SHAPE;
    if (log_n >= FR_TWO_ADICITY) return fail(ctx, KZG_ERR_DEGREE_TOO_LARGE, "polynomial degree too large");
    const size_t d = (size_t)1 << log_n;
    const Fr zi = inv(sub(pow_u64(from_u64<FrParams>(FR_MULT_GENERATOR), (uint64_t)d), Fr::one()));
    return vec_entry(ctx, data, nullptr, d, sfmt, flags, 0, &zi);
}

extern "C" int kzg_fr_vec_mul(kzg_ctx *ctx, void *a, const void *b, size_t n, int sfmt, int flags) {
    // EvaluationDomain::mul_assign (src/ft.rs:220-244); the length assert is the caller's (both vectors have n elements)
    if (!ctx || (n && (!a || !b))) return KZG_ERR_SHAPE;
    return vec_entry(ctx, a, b, n, sfmt, flags, 0, nullptr);
}

extern "C" int kzg_fr_vec_sub(kzg_ctx *ctx, void *a, const void *b, size_t n, int sfmt, int flags) {
    // EvaluationDomain::sub_assign (src/ft.rs:247-271)
    if (!ctx || (n && (!a || !b))) return KZG_ERR_SHAPE;
    return vec_entry(ctx, a, b, n, sfmt, flags, 1, nullptr);
}
