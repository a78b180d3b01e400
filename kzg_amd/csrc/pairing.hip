// pairing.hip -- the verifier half of the crate on the GPU: the G2 side of KZGParams (hs, src/lib.rs:48-52), small
// G2 multi-exponentiations and the pairing checks of KZGVerifier::{verify_eval, verify_eval_batched}
// (src/coeff_form.rs:126-182) and KZGVerifierEvalForm::{verify_eval, verify_eval_all} (src/eval_form.rs:173-217).
//
// `lhs == rhs` of two pairings is evaluated as  e(w, H') * e(-(C - [..]G), h0) == 1: one shared Miller loop over
// both pairs and one final exponentiation.  A check is a serial chain of ~25 k Fq multiplies, so ONE THREAD runs
// ONE check and a batch of openings is one launch (kzg_verify_eval takes `count` tuples); tower.h documents the
// arithmetic.  Nothing here is on the prover's throughput path.
#include "common.h"
#include "tower.h"

struct kzg_srs_g2 {
    size_t n = 0;
    kzg::G2Affine *pts = nullptr;  // affine Montgomery (= blst_p2_affine), identity all-zero
    kzg::Fq2 *lines = nullptr;     // Miller-loop lines of pts[0] and pts[1] (2 x 2*MILLER_LINES Fq2): the verifier's
                                   // second pairing argument is always one of these two, so a check does no G2 arithmetic
    int device = 0;
};

namespace kzg {

// ------------------------------------------------------------------------------------------------
// encodings (zcash: x.c1 || x.c0 [|| y.c1 || y.c0], big-endian, flag bits in the first byte)
// ------------------------------------------------------------------------------------------------
static __device__ Fq rd_be48(const uint8_t *src, bool mask_flags) {
    Fq r = Fq::zero();
    for (int i = 0; i < 48; i++) {
        uint32_t byte = src[47 - i];
        if (mask_flags && i == 47) byte &= 0x1f;
        r.v[i >> 2] |= byte << (8 * (i & 3));
    }
    return r;
}
static __device__ void wr_be48(uint8_t *dst, const Fq &canon) {
    for (int i = 0; i < 48; i++) dst[47 - i] = (uint8_t)(canon.v[i >> 2] >> (8 * (i & 3)));
}
static __device__ bool gt_half_q(const Fq &canon) {  // canon > (q-1)/2
    for (int i = 11; i >= 0; i--) {
        uint32_t h = (FqParams::mod(i) >> 1) | (i < 11 ? (FqParams::mod(i + 1) << 31) : 0u);  // (q-1)/2 = q >> 1
        if (canon.v[i] > h) return true;
        if (canon.v[i] < h) return false;
    }
    return false;
}
static __device__ bool f2_lex_largest(const Fq2 &y) {  // compares (c1, c0) with the negation's
    Fq c1 = from_mont(y.c1);
    if (!c1.is_zero()) return gt_half_q(c1);
    return gt_half_q(from_mont(y.c0));
}
// a^(q >> shift): (q-3)/4 = q >> 2, (q-1)/2 = q >> 1
static __device__ __noinline__ void f2_pow_q_shr(Fq2 &r, const Fq2 &a, int shift) {
    Fq2 acc = Fq2::one();
    for (int i = 383; i >= shift; i--) {
        f2_sqr(acc, acc);
        if ((FqParams::mod(i >> 5) >> (i & 31)) & 1) f2_mul(acc, acc, a);
    }
    r = acc;
}
// square root in Fq2, q = 3 mod 4 (Adj & Rodriguez-Henriquez alg. 9); false if `a` is a non-residue
static __device__ __noinline__ bool f2_sqrt(Fq2 &r, const Fq2 &a) {
    if (a.is_zero()) {
        r = a;
        return true;
    }
    Fq2 a1, alpha, x0, t;
    f2_pow_q_shr(a1, a, 2);
    f2_sqr(alpha, a1);
    f2_mul(alpha, alpha, a);
    f2_mul(x0, a1, a);
    Fq2 minus_one = Fq2{neg(Fq::one()), Fq::zero()};
    if (alpha == minus_one) {
        Fq2 u = Fq2{Fq::zero(), Fq::one()};
        f2_mul(r, u, x0);
    } else {
        Fq2 one = Fq2::one();
        f2_add(t, one, alpha);
        f2_pow_q_shr(t, t, 1);
        f2_mul(r, t, x0);
    }
    f2_sqr(t, r);
    return t == a;
}

static __device__ bool f2_canonical(const Fq2 &a) { return is_canonical(a.c0) && is_canonical(a.c1); }

// [r]Q == O: membership in G2 (the twist has a large cofactor; G2Affine deserialisation checks this upstream)
static __device__ __noinline__ bool g2_in_subgroup(const G2Affine &a) {
    if (a.is_inf()) return true;
    uint32_t k[8];
#pragma unroll
    for (int i = 0; i < 8; i++) k[i] = FrParams::mod(i);
    G2Jacobian r;
    g2_scalar_mul(r, a, k);
    return r.z.is_zero();
}

// level: POINTS_TRUSTED / POINTS_ON_CURVE / POINTS_SUBGROUP (common.h)
__global__ __launch_bounds__(64) void k_g2_decode(const uint8_t *src, size_t n, int fmt, G2Affine *out, int *bad, int level) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G2Affine a;
    bool ok = true;
    if (fmt == KZG_G2_AFFINE_MONT_192) {
        a = *reinterpret_cast<const G2Affine *>(src + i * 192);
        if (level >= POINTS_ON_CURVE) ok = f2_canonical(a.x) && f2_canonical(a.y) && g2_on_curve(a);
    } else if (fmt == KZG_G2_JACOBIAN_MONT_288) {
        G2Jacobian j = *reinterpret_cast<const G2Jacobian *>(src + i * 288);
        if (level >= POINTS_ON_CURVE) ok = f2_canonical(j.x) && f2_canonical(j.y) && f2_canonical(j.z);
        g2_to_affine(a, j);
        if (level >= POINTS_ON_CURVE) ok = ok && g2_on_curve(a);
    } else if (fmt == KZG_G2_ZCASH_UNCOMPRESSED_192) {
        const uint8_t *p = src + i * 192;
        if (p[0] & 0x80) ok = false;
        if (p[0] & 0x40) {
            a.x = Fq2::zero();
            a.y = Fq2::zero();
        } else {
            Fq x1 = rd_be48(p, true), x0 = rd_be48(p + 48, false), y1 = rd_be48(p + 96, false), y0 = rd_be48(p + 144, false);
            ok = ok && is_canonical(x0) && is_canonical(x1) && is_canonical(y0) && is_canonical(y1);
            a.x = Fq2{to_mont(x0), to_mont(x1)};
            a.y = Fq2{to_mont(y0), to_mont(y1)};
            ok = ok && g2_on_curve(a);
        }
    } else {
        const uint8_t *p = src + i * 96;
        if (!(p[0] & 0x80)) ok = false;
        if (p[0] & 0x40) {
            a.x = Fq2::zero();
            a.y = Fq2::zero();
        } else {
            Fq x1 = rd_be48(p, true), x0 = rd_be48(p + 48, false);
            ok = ok && is_canonical(x0) && is_canonical(x1);
            a.x = Fq2{to_mont(x0), to_mont(x1)};
            Fq2 rhs, b;
            f2_sqr(rhs, a.x);
            f2_mul(rhs, rhs, a.x);
            b.c0 = from_u64<FqParams>(4);
            b.c1 = b.c0;
            f2_add(rhs, rhs, b);
            ok = f2_sqrt(a.y, rhs) && ok;
            if (f2_lex_largest(a.y) != ((p[0] & 0x20) != 0)) f2_neg(a.y, a.y);
        }
    }
    if (ok && level >= POINTS_SUBGROUP) ok = g2_in_subgroup(a);
    if (!ok) {
        atomicOr(bad, 1);
        a.x = Fq2::zero();
        a.y = Fq2::zero();
    }
    out[i] = a;
}

static __device__ void g2_encode(const G2Affine &a, int fmt, uint8_t *o) {
    if (fmt == KZG_G2_AFFINE_MONT_192) {
        *reinterpret_cast<G2Affine *>(o) = a;
    } else if (fmt == KZG_G2_JACOBIAN_MONT_288) {
        G2Jacobian j;
        g2_from_affine(j, a);
        *reinterpret_cast<G2Jacobian *>(o) = j;
    } else if (fmt == KZG_G2_ZCASH_UNCOMPRESSED_192) {
        if (a.is_inf()) {
            for (int k = 0; k < 192; k++) o[k] = 0;
            o[0] = 0x40;
        } else {
            wr_be48(o, from_mont(a.x.c1));
            wr_be48(o + 48, from_mont(a.x.c0));
            wr_be48(o + 96, from_mont(a.y.c1));
            wr_be48(o + 144, from_mont(a.y.c0));
        }
    } else {
        if (a.is_inf()) {
            for (int k = 0; k < 96; k++) o[k] = 0;
            o[0] = 0xc0;
        } else {
            wr_be48(o, from_mont(a.x.c1));
            wr_be48(o + 48, from_mont(a.x.c0));
            o[0] |= 0x80 | (f2_lex_largest(a.y) ? 0x20 : 0);
        }
    }
}

__global__ __launch_bounds__(64) void k_g2_encode(const G2Affine *in, size_t n, int fmt, uint8_t *out, size_t stride) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    g2_encode(in[i], fmt, out + i * stride);
}

static size_t g2_format_bytes(int fmt) {
    switch (fmt) {
        case KZG_G2_AFFINE_MONT_192: return 192;
        case KZG_G2_JACOBIAN_MONT_288: return 288;
        case KZG_G2_ZCASH_UNCOMPRESSED_192: return 192;
        case KZG_G2_ZCASH_COMPRESSED_96: return 96;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// G2 scalar multiplications
// ------------------------------------------------------------------------------------------------
static __device__ void scalar_bits(uint32_t k[8], const Fr &s, int is_mont) {
    Fr c = is_mont ? from_mont(s) : s;
#pragma unroll
    for (int i = 0; i < 8; i++) k[i] = c.v[i];
}

// out[i] = [s_i] H (generator) as affine points; setup(): hs[i] = hs[i-1]*s (src/lib.rs:48-52)
__global__ __launch_bounds__(64) void k_g2_gen_mul(const Fr *scalars_mont, size_t n, G2Affine *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t k[8];
    scalar_bits(k, scalars_mont[i], 1);
    G2Jacobian r;
    g2_scalar_mul(r, g2_generator(), k);
    G2Affine a;
    g2_to_affine(a, r);
    out[i] = a;
}

// terms[i] = [s_i] P_i
__global__ __launch_bounds__(64) void k_g2_msm_terms(const G2Affine *pts, const Fr *scalars, size_t n, int is_mont,
                                                     G2Jacobian *terms) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t k[8];
    scalar_bits(k, scalars[i], is_mont);
    G2Jacobian r;
    g2_scalar_mul(r, pts[i], k);
    terms[i] = r;
}

// one block: strided partial sums, then thread 0 folds them; result affine
__global__ __launch_bounds__(64) void k_g2_sum(const G2Jacobian *terms, size_t n, G2Jacobian *partial, G2Affine *out) {
    G2Jacobian acc;
    g2_set_inf(acc);
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) g2_add(acc, acc, terms[i]);
    partial[threadIdx.x] = acc;
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x == 0) {
        for (unsigned t = 1; t < blockDim.x; t++) g2_add(acc, acc, partial[t]);
        G2Affine a;
        g2_to_affine(a, acc);
        *out = a;
    }
}

// out = a - b (single thread)
__global__ __launch_bounds__(64) void k_g2_sub(const G2Affine *a, const G2Affine *b, G2Affine *out) {
    G2Jacobian x, y;
    G2Affine nb;
    g2_neg_affine(nb, *b);
    g2_from_affine(x, *a);
    g2_from_affine(y, nb);
    g2_add(x, x, y);
    G2Affine r;
    g2_to_affine(r, x);
    *out = r;
}

// lines[j] = stored Miller lines of pts[j], j < count <= 2
__global__ __launch_bounds__(64) void k_g2_lines(const G2Affine *pts, int count, Fq2 *lines) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    g2_precompute_lines(pts[j], lines + (size_t)j * 2 * MILLER_LINES);
}

// ------------------------------------------------------------------------------------------------
// pairing checks
// ------------------------------------------------------------------------------------------------
constexpr int MAX_PAIRS = 4;

// ok[c] = prod_{i<np} e(P[c*np+i], Q[c*np+i]) == 1
__global__ __launch_bounds__(64) void k_pairing_check(const G1Xyzz *Ps, const G2Affine *Qs, int np, size_t checks, uint8_t *ok) {
    size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= checks) return;
    G1Affine P[MAX_PAIRS];
    G2Affine Q[MAX_PAIRS], T[MAX_PAIRS];
    for (int i = 0; i < np; i++) {
        P[i] = g1_to_affine(Ps[c * np + i]);
        Q[i] = Qs[c * np + i];
    }
    ok[c] = pairing_product_is_one(P, Q, T, np) ? 1 : 0;
}

// verify_eval (src/coeff_form.rs:126-142):  e(w, h1 - [x]h0) == e(C - [y]g0, h0).  By bilinearity the product
// e(w, h1 - [x]h0) e(-(C - [y]g0), h0) equals  e(w, h1) e(-([x]w + C - [y]g0), h0): both G2 arguments are SRS points
// whose lines are stored, so the per-opening work is two G1 scalar multiplications, the shared Miller loop and the
// final exponentiation -- no G2 arithmetic.
__global__ __launch_bounds__(64) void k_verify_eval(const Fr *xs, const Fr *ys, int is_mont, const G1Xyzz *Cs, const G1Xyzz *Ws,
                                                    const G1Affine *g0, const G2Affine *hs, const Fq2 *lines, size_t count,
                                                    uint8_t *ok) {
    size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= count) return;
    uint32_t k[8];
    G1Affine P[2];
    G2Affine Q[2], T[2];
    const Fq2 *tabs[2] = {lines + 2 * MILLER_LINES, lines};  // pair 0 against hs[1], pair 1 against hs[0]
    G1Affine w = g1_to_affine(Ws[c]);
    // B = [x]w + C - [y]g0 ;  P1 = -B
    scalar_bits(k, ys[c], is_mont);
    G1Xyzz yg = g1_scalar_mul(*g0, k);
    if (!yg.y.is_zero()) yg.y = neg(yg.y);
    scalar_bits(k, xs[c], is_mont);
    G1Xyzz acc = g1_add(g1_add(g1_scalar_mul(w, k), Cs[c]), yg);
    P[0] = w;
    P[1] = g1_neg(g1_to_affine(acc));
    Q[0] = hs[1];
    Q[1] = hs[0];
    ok[c] = pairing_product_is_one(P, Q, T, 2, tabs) ? 1 : 0;
}

// e(w, hz) == e(C - gr, h0)   (verify_eval_batched src/coeff_form.rs:144-182, verify_eval_all src/eval_form.rs:192-217);
// hz varies per call (lines on the fly), h0 uses its stored lines
__global__ __launch_bounds__(64) void k_verify_finish(const G1Xyzz *C, const G1Affine *gr, const G1Xyzz *w, const G2Affine *hz, const G2Affine *h0,
                                const Fq2 *lines_h0, uint8_t *ok) {
    G1Affine P[2];
    G2Affine Q[2], T[2];
    const Fq2 *tabs[2] = {nullptr, lines_h0};
    G1Xyzz ngr = G1Xyzz::from_affine(g1_neg(*gr));
    P[0] = g1_to_affine(*w);
    P[1] = g1_neg(g1_to_affine(g1_add(*C, ngr)));
    Q[0] = *hz;
    Q[1] = *h0;
    ok[0] = pairing_product_is_one(P, Q, T, 2, tabs) ? 1 : 0;
}

}  // namespace kzg

using namespace kzg;

namespace {
typedef kzg::Guard Lock;

int load_scalar(kzg_ctx *ctx, const void *s, int sfmt, Fr *mont) {
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

// stored Miller lines of the first two points (called once the points are on the device)
int g2_make_lines(kzg_ctx *ctx, hipStream_t st, kzg_srs_g2 *s) {
    int cnt = s->n < 2 ? (int)s->n : 2;
    if (hipMalloc((void **)&s->lines, 2 * 2 * MILLER_LINES * sizeof(Fq2)) != hipSuccess) return fail(ctx, KZG_ERR_ALLOC, "hipMalloc(G2 lines)");
    if (hipMemsetAsync(s->lines, 0, 2 * 2 * MILLER_LINES * sizeof(Fq2), st) != hipSuccess) return fail(ctx, KZG_ERR_HIP, "memset");
    if (cnt) KZG_LAUNCH(ctx, st, "k_g2_lines", k_g2_lines, 1, 64, 0, s->pts, cnt, s->lines);
    return KZG_OK;
}

int g2_from_scalars(kzg_ctx *ctx, hipStream_t st, const Fr *d_scalars_mont, size_t n, kzg_srs_g2 **out) {
    kzg_srs_g2 *s = new kzg_srs_g2();
    s->n = n;
    s->device = ctx->device;
    if (hipMalloc((void **)&s->pts, (n ? n : 1) * sizeof(G2Affine)) != hipSuccess) {
        delete s;
        return fail(ctx, KZG_ERR_ALLOC, "hipMalloc(G2 SRS)");
    }
    if (n) KZG_LAUNCH(ctx, st, "k_g2_gen_mul", k_g2_gen_mul, (unsigned)((n + 63) / 64), 64, 0, d_scalars_mont, n, s->pts);
    int lrc = g2_make_lines(ctx, st, s);
    if (hipStreamSynchronize(st) != hipSuccess || lrc != KZG_OK) {
        hipFree(s->pts);
        if (s->lines) hipFree(s->lines);
        delete s;
        return lrc != KZG_OK ? lrc : fail(ctx, KZG_ERR_HIP, "G2 SRS kernel failed");
    }
    *out = s;
    return KZG_OK;
}

// decode `count` G1 points of format pfmt (host memory) into XYZZ on the device
int g1_inputs(kzg_ctx *ctx, const void *host, size_t count, int pfmt, G1Xyzz **d_out, int *d_bad) {
    size_t psz = point_format_bytes(pfmt);
    if (!psz) return fail(ctx, KZG_ERR_SHAPE, "unknown G1 point format");
    hipStream_t st = ctx->lanes[0].stream;
    void *raw = lane_alloc(ctx, 0, count * psz);
    G1Xyzz *pts = (G1Xyzz *)lane_alloc(ctx, 0, count * sizeof(G1Xyzz));
    if (!raw || !pts) return fail(ctx, KZG_ERR_ALLOC, "workspace");
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(raw, host, count * psz, hipMemcpyHostToDevice, st));
    KZG_TRY(decode_points(ctx, st, raw, count, pfmt, pts, d_bad, untrusted_level(ctx)));
    *d_out = pts;
    return KZG_OK;
}

int fetch_ok(kzg_ctx *ctx, const uint8_t *d_ok, const int *d_bad, size_t count, uint8_t *ok) {
    hipStream_t st = ctx->lanes[0].stream;
    KZG_TRY(lane_pinned(ctx, 0, count + 64));
    char *pin = ctx->lanes[0].pinned;
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(pin, d_bad, sizeof(int), hipMemcpyDeviceToHost, st));
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(pin + 64, d_ok, count, hipMemcpyDeviceToHost, st));
    KZG_HIP_CHECK(ctx, hipStreamSynchronize(st));
    if (ctx->prof) prof_collect(ctx);
    if (*(int *)pin) return fail(ctx, KZG_ERR_BAD_POINT, "an input point failed to decode, is not on the curve or not in the r-torsion subgroup");
    memcpy(ok, pin + 64, count);
    return KZG_OK;
}
}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI: G2 SRS
// ------------------------------------------------------------------------------------------------
extern "C" int kzg_srs_setup_g2(kzg_ctx *ctx, const void *sec, int sfmt, size_t n, kzg_srs_g2 **out) {
    if (!ctx || !out || !sec) return KZG_ERR_SHAPE;
    Lock g(ctx);
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    Fr tau;
    KZG_TRY(load_scalar(ctx, sec, sfmt, &tau));
    hipStream_t st = ctx->lanes[0].stream;
    Fr *sc = nullptr;
    KZG_HIP_CHECK(ctx, hipMalloc((void **)&sc, (n ? n : 1) * sizeof(Fr)));
    int rc = powers_run(ctx, st, tau, 0, n, sc);
    if (rc == KZG_OK) rc = g2_from_scalars(ctx, st, sc, n, out);
    hipStreamSynchronize(st);
    hipFree(sc);
    return rc;
}

extern "C" int kzg_srs_setup_lagrange_g2(kzg_ctx *ctx, const void *sec, int sfmt, size_t d, kzg_srs_g2 **out) {
    if (!ctx || !out || !sec) return KZG_ERR_SHAPE;
    Lock g(ctx);
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (d == 0 || (d & (d - 1))) return fail(ctx, KZG_ERR_SHAPE, "Lagrange basis needs a power-of-two size (src/eval_form.rs:255-256)");
    if ((uint32_t)ilog2_ceil(d) >= FR_TWO_ADICITY) return fail(ctx, KZG_ERR_DEGREE_TOO_LARGE, "domain too large");
    Fr tau;
    KZG_TRY(load_scalar(ctx, sec, sfmt, &tau));
    hipStream_t st = ctx->lanes[0].stream;
    Fr *sc = nullptr;
    KZG_HIP_CHECK(ctx, hipMalloc((void **)&sc, d * sizeof(Fr)));
    int rc = lagrange_scalars_run(ctx, st, tau, d, sc);
    if (rc == KZG_OK) rc = g2_from_scalars(ctx, st, sc, d, out);
    hipStreamSynchronize(st);
    hipFree(sc);
    return rc;
}

extern "C" int kzg_srs_upload_g2(kzg_ctx *ctx, const void *pts, size_t n, int pfmt, kzg_srs_g2 **out) {
    if (!ctx || !out || (!pts && n)) return KZG_ERR_SHAPE;
    Lock g(ctx);
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    size_t psz = g2_format_bytes(pfmt);
    if (!psz) return fail(ctx, KZG_ERR_SHAPE, "unknown G2 point format");
    hipStream_t st = ctx->lanes[0].stream;
    kzg_srs_g2 *s = new kzg_srs_g2();
    s->n = n;
    s->device = ctx->device;
    uint8_t *raw = nullptr;
    int *bad = nullptr;
    int hbad = 0, rc = KZG_OK;
    if (hipMalloc((void **)&s->pts, (n ? n : 1) * sizeof(G2Affine)) != hipSuccess ||
        hipMalloc((void **)&raw, (n ? n : 1) * psz) != hipSuccess || hipMalloc((void **)&bad, sizeof(int)) != hipSuccess)
        rc = fail(ctx, KZG_ERR_ALLOC, "hipMalloc(G2 SRS)");
    if (rc == KZG_OK && n) {
        hipMemsetAsync(bad, 0, sizeof(int), st);
        hipMemcpyAsync(raw, pts, n * psz, hipMemcpyHostToDevice, st);
        KZG_LAUNCH(ctx, st, "k_g2_decode", k_g2_decode, (unsigned)((n + 63) / 64), 64, 0, raw, n, pfmt, s->pts, bad, untrusted_level(ctx));
        hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, st);
        if (hipStreamSynchronize(st) != hipSuccess) rc = fail(ctx, KZG_ERR_HIP, "G2 decode failed");
        if (rc == KZG_OK && hbad) rc = fail(ctx, KZG_ERR_BAD_POINT, "a G2 point failed to decode, is not on the twist or not in the r-torsion subgroup");
    }
    if (rc == KZG_OK) {
        rc = g2_make_lines(ctx, st, s);
        if (rc == KZG_OK && hipStreamSynchronize(st) != hipSuccess) rc = fail(ctx, KZG_ERR_HIP, "G2 lines failed");
    }
    if (raw) hipFree(raw);
    if (bad) hipFree(bad);
    if (rc != KZG_OK) {
        if (s->pts) hipFree(s->pts);
        if (s->lines) hipFree(s->lines);
        delete s;
        return rc;
    }
    *out = s;
    return KZG_OK;
}

extern "C" int kzg_srs_download_g2(kzg_ctx *ctx, const kzg_srs_g2 *srs, size_t offset, size_t n, void *out, int pfmt) {
    if (!ctx || !srs || (!out && n)) return KZG_ERR_SHAPE;
    Lock g(ctx);
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (n > srs->n || offset > srs->n - n) return fail(ctx, KZG_ERR_SHAPE, "SRS download range out of bounds");
    size_t psz = g2_format_bytes(pfmt);
    if (!psz) return fail(ctx, KZG_ERR_SHAPE, "unknown G2 point format");
    if (!n) return KZG_OK;
    hipStream_t st = ctx->lanes[0].stream;
    uint8_t *enc = nullptr;
    KZG_HIP_CHECK(ctx, hipMalloc((void **)&enc, n * psz));
    KZG_LAUNCH(ctx, st, "k_g2_encode", k_g2_encode, (unsigned)((n + 63) / 64), 64, 0, srs->pts + offset, n, pfmt, enc, psz);
    hipError_t e = hipMemcpyAsync(out, enc, n * psz, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hipFree(enc);
    if (e != hipSuccess) return fail(ctx, KZG_ERR_HIP, hipGetErrorString(e));
    return KZG_OK;
}

extern "C" size_t kzg_srs_g2_len(const kzg_srs_g2 *srs) { return srs ? srs->n : 0; }

extern "C" void kzg_srs_g2_free(kzg_ctx *ctx, kzg_srs_g2 *srs) {
    if (!srs) return;
    if (ctx) {
        Lock g(ctx);
        hipSetDevice(ctx->device);
        hipStreamSynchronize(ctx->lanes[0].stream);
    }
    hipFree(srs->pts);
    if (srs->lines) hipFree(srs->lines);
    delete srs;
}

// sum_i scalars[i] * srs[offset + i] on the device -> one affine point at d_out
static int g2_msm_device(kzg_ctx *ctx, const kzg_srs_g2 *srs, size_t offset, const Fr *d_scalars, size_t n, int is_mont,
                         G2Affine *d_out) {
    if (srs->device != ctx->device) return fail(ctx, KZG_ERR_SHAPE, "the G2 points are resident on another GPU than this context's");
    hipStream_t st = ctx->lanes[0].stream;
    G2Jacobian *terms = (G2Jacobian *)lane_alloc(ctx, 0, (n + 64) * sizeof(G2Jacobian));
    if (!terms) return fail(ctx, KZG_ERR_ALLOC, "workspace");
    if (n) KZG_LAUNCH(ctx, st, "k_g2_msm_terms", k_g2_msm_terms, (unsigned)((n + 63) / 64), 64, 0, srs->pts + offset, d_scalars, n, is_mont, terms);
    KZG_LAUNCH(ctx, st, "k_g2_sum", k_g2_sum, 1, 64, 0, terms, n, terms + n, d_out);
    return KZG_OK;
}

// compute_lagrange_basis (src/eval_form.rs:254-280), G2 half: row i = MSM(hs[..d], w^(-ij)/d), as for G1
// (kzg_srs_lagrange_from_monomial_g1).  d*d scalar multiplications: meant for the sizes the reference's O(d^3)
// construction can handle; larger bases come from kzg_srs_setup_lagrange_g2 or an upload.
extern "C" int kzg_srs_lagrange_from_monomial_g2(kzg_ctx *ctx, const kzg_srs_g2 *hs, kzg_srs_g2 **out) {
    if (!ctx || !hs || !out) return KZG_ERR_SHAPE;
    Lock g(ctx);
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    size_t d = hs->n;
    if (d == 0 || (d & (d - 1))) return fail(ctx, KZG_ERR_SHAPE, "assert!(d & (d - 1) == 0) (src/eval_form.rs:255-256)");
    if (d > 1024) return fail(ctx, KZG_ERR_SHAPE, "compute_lagrange_basis (G2) from hs is limited to d <= 1024; use kzg_srs_setup_lagrange_g2 or upload the basis");
    uint32_t exp = (uint32_t)ilog2_ceil(d);
    hipStream_t st = ctx->lanes[0].stream;
    kzg_srs_g2 *s = new kzg_srs_g2();
    s->n = d;
    s->device = ctx->device;
    if (hipMalloc((void **)&s->pts, d * sizeof(G2Affine)) != hipSuccess) {
        delete s;
        return fail(ctx, KZG_ERR_ALLOC, "hipMalloc(G2 basis)");
    }
    Fr omega_inv = inv(host_omega(exp));
    Fr dinv = inv(from_u64<FrParams>((uint64_t)d));
    int rc = KZG_OK;
    for (size_t i = 0; i < d && rc == KZG_OK; i++) {
        rc = lane_reserve(ctx, 0, d * 32 + (d + 64) * sizeof(G2Jacobian) + 65536);
        Fr *sc = rc == KZG_OK ? (Fr *)lane_alloc(ctx, 0, d * 32) : nullptr;
        if (rc == KZG_OK && !sc) rc = fail(ctx, KZG_ERR_ALLOC, "workspace");
        if (rc == KZG_OK) rc = pow_table(ctx, st, pow_u64(omega_inv, (uint64_t)i), dinv, d, sc);
        if (rc == KZG_OK) rc = g2_msm_device(ctx, hs, 0, sc, d, 1, s->pts + i);
    }
    if (rc == KZG_OK) rc = g2_make_lines(ctx, st, s);
    if (hipStreamSynchronize(st) != hipSuccess && rc == KZG_OK) rc = fail(ctx, KZG_ERR_HIP, "G2 basis kernels failed");
    if (rc != KZG_OK) {
        hipFree(s->pts);
        if (s->lines) hipFree(s->lines);
        delete s;
        return rc;
    }
    *out = s;
    return KZG_OK;
}

extern "C" int kzg_msm_g2(kzg_ctx *ctx, const kzg_srs_g2 *srs, size_t offset, const void *scalars, size_t n, int sfmt, void *out,
                          int ofmt) {
    // G2Projective::multi_exp (src/coeff_form.rs:156): small-n G2 multi-exponentiation, host-resident scalars
    if (!ctx || !srs || !out || (!scalars && n)) return KZG_ERR_SHAPE;
    Lock g(ctx);
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (sfmt != KZG_FR_MONT_LE_32 && sfmt != KZG_FR_CANONICAL_LE_32) return fail(ctx, KZG_ERR_SHAPE, "unknown scalar format");
    size_t psz = g2_format_bytes(ofmt);
    if (!psz) return fail(ctx, KZG_ERR_SHAPE, "unknown G2 point format");
    if (n > srs->n || offset > srs->n - n) return fail(ctx, KZG_ERR_SHAPE, "MSM range exceeds the SRS (reference: slice index panic)");
    KZG_TRY(lane_reserve(ctx, 0, n * 32 + (n + 64) * sizeof(G2Jacobian) + 8192));
    hipStream_t st = ctx->lanes[0].stream;
    Fr *ds = (Fr *)lane_alloc(ctx, 0, n * 32 + 32);
    G2Affine *res = (G2Affine *)lane_alloc(ctx, 0, sizeof(G2Affine));
    uint8_t *enc = (uint8_t *)lane_alloc(ctx, 0, 512);
    if (!ds || !res || !enc) return fail(ctx, KZG_ERR_ALLOC, "workspace");
    if (n) KZG_HIP_CHECK(ctx, hipMemcpyAsync(ds, scalars, n * 32, hipMemcpyHostToDevice, st));
    KZG_TRY(g2_msm_device(ctx, srs, offset, ds, n, sfmt == KZG_FR_MONT_LE_32, res));
    KZG_LAUNCH(ctx, st, "k_g2_encode", k_g2_encode, 1, 64, 0, res, (size_t)1, ofmt, enc, psz);
    KZG_TRY(lane_pinned(ctx, 0, 4096));
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(ctx->lanes[0].pinned, enc, psz, hipMemcpyDeviceToHost, st));
    KZG_HIP_CHECK(ctx, hipStreamSynchronize(st));
    memcpy(out, ctx->lanes[0].pinned, psz);
    if (ctx->prof) prof_collect(ctx);
    return KZG_OK;
}

// ------------------------------------------------------------------------------------------------
// C ABI: pairing checks
// ------------------------------------------------------------------------------------------------
extern "C" int kzg_pairing_check(kzg_ctx *ctx, const void *g1_points, int pfmt1, const void *g2_points, int pfmt2,
                                 size_t pairs_per_check, size_t checks, uint8_t *ok) {
    if (!ctx || !ok || ((!g1_points || !g2_points) && checks)) return KZG_ERR_SHAPE;
    Lock g(ctx);
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (pairs_per_check < 1 || pairs_per_check > (size_t)MAX_PAIRS) return fail(ctx, KZG_ERR_SHAPE, "1..4 pairs per check");
    size_t p2 = g2_format_bytes(pfmt2), p1 = point_format_bytes(pfmt1);
    if (!p1 || !p2) return fail(ctx, KZG_ERR_SHAPE, "unknown point format");
    if (!checks) return KZG_OK;
    size_t total = pairs_per_check * checks;
    KZG_TRY(lane_reserve(ctx, 0, total * (p1 + p2 + sizeof(G1Xyzz) + sizeof(G2Affine) + 512) + checks + 8192));
    hipStream_t st = ctx->lanes[0].stream;
    int *bad = (int *)lane_alloc(ctx, 0, 256);
    uint8_t *d_ok = (uint8_t *)lane_alloc(ctx, 0, checks);
    uint8_t *raw2 = (uint8_t *)lane_alloc(ctx, 0, total * p2);
    G2Affine *q = (G2Affine *)lane_alloc(ctx, 0, total * sizeof(G2Affine));
    if (!bad || !d_ok || !raw2 || !q) return fail(ctx, KZG_ERR_ALLOC, "workspace");
    KZG_HIP_CHECK(ctx, hipMemsetAsync(bad, 0, sizeof(int), st));
    G1Xyzz *p = nullptr;
    KZG_TRY(g1_inputs(ctx, g1_points, total, pfmt1, &p, bad));
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(raw2, g2_points, total * p2, hipMemcpyHostToDevice, st));
    KZG_LAUNCH(ctx, st, "k_g2_decode", k_g2_decode, (unsigned)((total + 63) / 64), 64, 0, raw2, total, pfmt2, q, bad, untrusted_level(ctx));
    KZG_LAUNCH(ctx, st, "k_pairing_check", k_pairing_check, (unsigned)((checks + 63) / 64), 64, 0, p, q, (int)pairs_per_check, checks, d_ok);
    return fetch_ok(ctx, d_ok, bad, checks, ok);
}

extern "C" int kzg_verify_eval(kzg_ctx *ctx, const kzg_srs *gs, const kzg_srs_g2 *hs, const void *xs, const void *ys, int sfmt,
                               const void *commitments, const void *witnesses, int pfmt, size_t count, uint8_t *ok) {
    // KZGVerifier::verify_eval (src/coeff_form.rs:126-142), `count` independent openings, one GPU thread each
    if (!ctx || !gs || !hs || !ok || ((!xs || !ys || !commitments || !witnesses) && count)) return KZG_ERR_SHAPE;
    Lock g(ctx);
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (sfmt != KZG_FR_MONT_LE_32 && sfmt != KZG_FR_CANONICAL_LE_32) return fail(ctx, KZG_ERR_SHAPE, "unknown scalar format");
    if (gs->n < 1 || hs->n < 2) return fail(ctx, KZG_ERR_SHAPE, "verify_eval needs gs[0], hs[0], hs[1] (reference: index panic)");
    size_t psz = point_format_bytes(pfmt);
    if (!psz || pfmt == KZG_G1_JACOBIAN_MONT_144) return fail(ctx, KZG_ERR_SHAPE, "commitments / witnesses are affine (G1Affine)");
    if (!count) return KZG_OK;
    KZG_TRY(lane_reserve(ctx, 0, count * (64 + 2 * psz + 2 * sizeof(G1Xyzz) + 1024) + 8192));
    hipStream_t st = ctx->lanes[0].stream;
    int *bad = (int *)lane_alloc(ctx, 0, 256);
    uint8_t *d_ok = (uint8_t *)lane_alloc(ctx, 0, count);
    Fr *dx = (Fr *)lane_alloc(ctx, 0, count * 32), *dy = (Fr *)lane_alloc(ctx, 0, count * 32);
    if (!bad || !d_ok || !dx || !dy) return fail(ctx, KZG_ERR_ALLOC, "workspace");
    KZG_HIP_CHECK(ctx, hipMemsetAsync(bad, 0, sizeof(int), st));
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(dx, xs, count * 32, hipMemcpyHostToDevice, st));
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(dy, ys, count * 32, hipMemcpyHostToDevice, st));
    G1Xyzz *C = nullptr, *W = nullptr;
    KZG_TRY(g1_inputs(ctx, commitments, count, pfmt, &C, bad));
    KZG_TRY(g1_inputs(ctx, witnesses, count, pfmt, &W, bad));
    KZG_LAUNCH(ctx, st, "k_verify_eval", k_verify_eval, (unsigned)((count + 63) / 64), 64, 0, dx, dy, sfmt == KZG_FR_MONT_LE_32 ? 1 : 0, C, W,
               gs->table, hs->pts, hs->lines, count, d_ok);
    return fetch_ok(ctx, d_ok, bad, count, ok);
}

// shared tail of verify_eval_batched / verify_eval_all: gr = MSM(g1 basis, r), then the pairing check against d_hz
static int verify_with_hz(kzg_ctx *ctx, const kzg_srs *basis_g, const void *r, size_t r_len, int sfmt, const G2Affine *d_hz,
                          const G2Affine *d_h0, const Fq2 *d_lines_h0, const void *commitment, const void *witness, int pfmt, int *ok) {
    hipStream_t st = ctx->lanes[0].stream;
    int *bad = (int *)lane_alloc(ctx, 0, 256);
    uint8_t *d_ok = (uint8_t *)lane_alloc(ctx, 0, 256);
    G1Affine *gr = (G1Affine *)lane_alloc(ctx, 0, 256);
    Fr *dr = (Fr *)lane_alloc(ctx, 0, r_len * 32 + 32);
    if (!bad || !d_ok || !gr || !dr) return fail(ctx, KZG_ERR_ALLOC, "workspace");
    KZG_HIP_CHECK(ctx, hipMemsetAsync(bad, 0, sizeof(int), st));
    G1Xyzz *C = nullptr, *W = nullptr;
    KZG_TRY(g1_inputs(ctx, commitment, 1, pfmt, &C, bad));
    KZG_TRY(g1_inputs(ctx, witness, 1, pfmt, &W, bad));
    if (r_len) KZG_HIP_CHECK(ctx, hipMemcpyAsync(dr, r, r_len * 32, hipMemcpyHostToDevice, st));
    MsmPoint *res = nullptr;
    KZG_TRY(msm_run(ctx, 0, basis_g, 0, dr, r_len, sfmt, &res));
    KZG_TRY(emit_point(ctx, 0, res, gr, KZG_G1_AFFINE_MONT_96));
    KZG_LAUNCH(ctx, st, "k_verify_finish", k_verify_finish, 1, 1, 0, C, gr, W, d_hz, d_h0, d_lines_h0, d_ok);
    uint8_t r8 = 0;
    KZG_TRY(fetch_ok(ctx, d_ok, bad, 1, &r8));
    *ok = r8;
    return KZG_OK;
}

extern "C" int kzg_verify_eval_batched(kzg_ctx *ctx, const kzg_srs *gs, const kzg_srs_g2 *hs, const void *xs, size_t k,
                                       const void *r_coeffs, size_t r_len, int sfmt, const void *commitment, const void *witness,
                                       int pfmt, int *ok) {
    // KZGVerifier::verify_eval_batched (src/coeff_form.rs:144-182): z = prod (X - x_i); hz = MSM(hs, z);
    // gr = MSM(gs, witness.r); e(w, hz) == e(C - gr, hs[0])
    if (!ctx || !gs || !hs || !ok || !commitment || !witness || (!xs && k) || (!r_coeffs && r_len)) return KZG_ERR_SHAPE;
    Lock g(ctx);
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (sfmt != KZG_FR_MONT_LE_32 && sfmt != KZG_FR_CANONICAL_LE_32) return fail(ctx, KZG_ERR_SHAPE, "unknown scalar format");
    size_t psz = point_format_bytes(pfmt);
    if (!psz || pfmt == KZG_G1_JACOBIAN_MONT_144) return fail(ctx, KZG_ERR_SHAPE, "commitment / witness are affine (G1Affine)");
    if (k == 0) return fail(ctx, KZG_ERR_SHAPE, "no points (reference: op_tree over an empty set panics)");
    if (k > 16384) return fail(ctx, KZG_ERR_SHAPE, "batched verification is limited to 16384 points");
    if (k >= hs->n) return fail(ctx, KZG_ERR_SHAPE, "z longer than hs (reference: slice index panic)");
    if (r_len > gs->n) return fail(ctx, KZG_ERR_SHAPE, "witness.r longer than gs (reference: slice index panic)");
    KZG_TRY(lane_reserve(ctx, 0, msm_workspace_bytes(gs, r_len) + (k + 2) * (3 * 32 + sizeof(G2Jacobian)) + 64 * sizeof(G2Jacobian) +
                                     r_len * 32 + 65536));
    hipStream_t st = ctx->lanes[0].stream;
    Fr *dx = (Fr *)lane_alloc(ctx, 0, k * 32), *z0 = (Fr *)lane_alloc(ctx, 0, (k + 1) * 32), *z1 = (Fr *)lane_alloc(ctx, 0, (k + 1) * 32);
    G2Affine *hz = (G2Affine *)lane_alloc(ctx, 0, sizeof(G2Affine));
    if (!dx || !z0 || !z1 || !hz) return fail(ctx, KZG_ERR_ALLOC, "workspace");
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(dx, xs, k * 32, hipMemcpyHostToDevice, st));
    if (sfmt == KZG_FR_CANONICAL_LE_32) KZG_TRY(fr_convert(ctx, st, dx, k, 1));
    KZG_TRY(vanishing_poly_run(ctx, st, dx, k, z0, z1));  // Montgomery coefficients
    KZG_TRY(g2_msm_device(ctx, hs, 0, z0, k + 1, 1, hz));
    return verify_with_hz(ctx, gs, r_coeffs, r_len, sfmt, hz, hs->pts, hs->lines, commitment, witness, pfmt, ok);
}

extern "C" int kzg_verify_eval_all(kzg_ctx *ctx, const kzg_srs *lagrange_g, const kzg_srs_g2 *lagrange_h, const kzg_srs_g2 *hs,
                                   const void *ys, size_t ys_len, int sfmt, const void *commitment, const void *witness, int pfmt,
                                   int *ok) {
    // KZGVerifierEvalForm::verify_eval_all (src/eval_form.rs:192-217), as written there: z has -1 at index 0 and
    // 1 at index d-1 (d = the domain size of lagrange_h), hz = MSM(lagrange_h, z), gr = MSM(lagrange_g[..ys.len()], ys)
    if (!ctx || !lagrange_g || !lagrange_h || !hs || !ok || !commitment || !witness || (!ys && ys_len)) return KZG_ERR_SHAPE;
    Lock g(ctx);
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (sfmt != KZG_FR_MONT_LE_32 && sfmt != KZG_FR_CANONICAL_LE_32) return fail(ctx, KZG_ERR_SHAPE, "unknown scalar format");
    size_t psz = point_format_bytes(pfmt);
    if (!psz || pfmt == KZG_G1_JACOBIAN_MONT_144) return fail(ctx, KZG_ERR_SHAPE, "commitment / witness are affine (G1Affine)");
    size_t d = lagrange_h->n;
    if (d == 0 || hs->n == 0) return fail(ctx, KZG_ERR_SHAPE, "empty basis (reference: index panic)");
    if (ys_len > lagrange_g->n) return fail(ctx, KZG_ERR_SHAPE, "ys longer than the Lagrange basis (reference: slice index panic)");
    KZG_TRY(lane_reserve(ctx, 0, msm_workspace_bytes(lagrange_g, ys_len) + ys_len * 32 + 65536));
    hipStream_t st = ctx->lanes[0].stream;
    G2Affine *hz = (G2Affine *)lane_alloc(ctx, 0, sizeof(G2Affine));
    if (!hz) return fail(ctx, KZG_ERR_ALLOC, "workspace");
    if (d == 1) {
        KZG_HIP_CHECK(ctx, hipMemcpyAsync(hz, lagrange_h->pts, sizeof(G2Affine), hipMemcpyDeviceToDevice, st));  // the later write (1) wins
    } else {
        KZG_LAUNCH(ctx, st, "k_g2_sub", k_g2_sub, 1, 1, 0, lagrange_h->pts + (d - 1), lagrange_h->pts, hz);
    }
    return verify_with_hz(ctx, lagrange_g, ys, ys_len, sfmt, hz, hs->pts, hs->lines, commitment, witness, pfmt, ok);
}
