// srs.hip -- resident SRS objects: KZGParams.gs (src/lib.rs:14-19) and lagrange_basis_g
// (src/eval_form.rs:40-46), decoded once into the engine's layout: W rows of affine Montgomery
// points, row w = 2^(c*w) * P_i (see msm.hip for why).  Also the GPU versions of the untimed input
// generators: setup() (src/lib.rs:38-47, G1 half) and the Lagrange basis for a known secret.
#include "common.h"
#include "curve30.h"

namespace kzg {

// ---------------------------------------------------------------------------------------------
// batch XYZZ -> affine (Montgomery trick: one Fq inversion per K points)
// ---------------------------------------------------------------------------------------------
constexpr int BA_K = 16;

__global__ __launch_bounds__(256) void k_batch_affine(const G1Xyzz *in, G1Affine *out, size_t n) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t i0 = t * BA_K;
    if (i0 >= n) return;
    size_t i1 = i0 + BA_K < n ? i0 + BA_K : n;
    Fq prod = Fq::one();
    for (size_t i = i0; i < i1; i++) {  // out[i].x temporarily holds the prefix product before i
        out[i].x = prod;
        Fq zzz = in[i].zzz;
        if (!in[i].zz.is_zero()) prod = mul(prod, zzz);
    }
    Fq iv = inv(prod);
    for (size_t i = i1; i-- > i0;) {
        G1Xyzz p = in[i];
        if (p.is_inf()) {
            out[i] = G1Affine::inf();
            continue;
        }
        Fq izzz = mul(iv, out[i].x);
        iv = mul(iv, p.zzz);
        out[i] = g1_to_affine_with_inv(p, izzz);
    }
}

int batch_to_affine(kzg_ctx *ctx, hipStream_t stream, const G1Xyzz *d_in, G1Affine *d_out, size_t n) {
    if (n == 0) return KZG_OK;
    size_t threads = (n + BA_K - 1) / BA_K;
    KZG_LAUNCH(ctx, stream, "k_batch_affine", k_batch_affine, (unsigned)((threads + 255) / 256), 256, 0, d_in, d_out, n);
    return KZG_OK;
}

// ---------------------------------------------------------------------------------------------
// per-window precompute: row w = 2^c * row (w-1)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dbl_c(const G1Affine *in, G1Xyzz *out, size_t n, int c) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Xyzz p = G1Xyzz::from_affine(in[i]);
    for (int k = 0; k < c; k++) p = g1_dbl(p);
    out[i] = p;
}

int srs_choose_window(int opt_window_bits, size_t n) {
    int c = opt_window_bits;
    if (c == 0) {
        // Measured (tools/window_sweep*.sh, batch 64 and single latency, profiles/r02_window_sweep.txt): with the depth-organised
        // tail the bucket count is cheap, so wide windows pay much earlier than the old "log2 n - 4": 17 bits (15 windows, the
        // balanced-scalar mode) from 2^17 points on -- +7 % at 2^17, +20 % at 2^18, +13 % at 2^19 over 13 / 14 / 15 bits.
        // Widths whose TOP window holds only a few significant bits of a 255-bit scalar (255 - c (W - 1) <= 3: c = 9, 11, 12,
        // 14, 15) are avoided: every scalar then lands in the same handful of buckets of that window, which the fold handles
        // through its overflow path (correct, but a 0.2 ms serial stage).  c = 8, 10, 13, 16, 17 have 7, 5, 8, 15, 17 bits there.
        const int l = ilog2_ceil(n ? n : 1);
        // From 2^23 points on 20 bits (13 windows, 2^19 buckets, its own two-level sort): the accumulation's 13 % fewer additions
        // outgrow the 8x bucket reduction -- same box, batched, full-width scalars: 2^20 397 against 432/s, 2^21 224 against 221,
        // 2^22 +8.4 %, 2^23 +9.8 %, 2^24 +12.7 %.  u64-valued scalars (4 windows either way) only pay for the larger bucket
        // reduction: 2^22 -13.7 %, 2^24 -4.9 % -- which is why the switch is at 2^23 and not 2^22 (profiles/r03_window20.txt).
        c = l >= 23 ? 20 : l >= 17 ? 17 : l >= 14 ? 13 : l >= 12 ? 10 : 8;
    }
    if (c < 4) c = 4;
    // c <= 16: the 2^(c-1) u32 LDS counters fit the CU's 160 KiB; 17: same pipeline, 15 windows, a two-level sort (msm.hip);
    // 18, 19 use the two-pass ("wide") sort of msm.hip and are only taken when asked for (option window_bits); 20: two-level sort
    // of msm_wide.hip, the default from 2^23 points on
    if (c > 20) c = 20;
    return c;
}

// window bits, windows, resident table rows for an SRS of n points under the given options
constexpr int SORT20_C_BITS = 20;  // = SORT20_C of msm_internal.h
constexpr int NAF_ROWS = 255;  // positional tables: bit positions 0..254 of a balanced scalar (k or r - k, below 2^254)

void srs_shape(int opt_window_bits, int opt_window_rows, size_t n, int *c_out, int *W_out, int *rows_out, bool *narrow17_out, int opt_naf,
               int *naf_out) {
    if (naf_out) *naf_out = 0;
    // positional tables only on request (option naf_window = 18): see kzg_srs::naf for the trade
    if (naf_out && opt_naf == 18) {
        *naf_out = 18;
        *c_out = 17;       // 2^16 buckets
        *W_out = 15;       // digits are >= 18 positions apart: at most 15 of them below bit 255
        *rows_out = NAF_ROWS;
        *narrow17_out = true;
        return;
    }
    int c = srs_choose_window(opt_window_bits, n);
    int W = (256 + c - 1) / c;
    // c = 17: 255 = 15 x 17, and a scalar k >= 2^254 is replaced by -(r - k) (all digit signs flipped), so the top window never
    // carries out: 15 windows instead of 16.  2^16 buckets: sorted in two levels (1024 bins of 64 buckets, msm.hip).
    bool narrow17 = c == 17;
    if (narrow17) W = 15;
    int rows = W;
    if (opt_window_rows > 0 && opt_window_rows < W && c <= 17) rows = opt_window_rows;  // the wide path keeps every row
    *c_out = c;
    *W_out = W;
    *rows_out = rows;
    *narrow17_out = narrow17;
}

int srs_alloc(kzg_ctx *ctx, size_t n, kzg_srs **out) {
    kzg_srs *s = new kzg_srs();
    s->n = n;
    s->npad = n ? n : 1;
    srs_shape(ctx->opt_window_bits, ctx->opt_window_rows, n, &s->c, &s->W, &s->rows, &s->narrow17, ctx->opt_naf_window, &s->naf);
    s->row_shift = s->naf ? 1 : s->c;
    s->sort20 = !s->naf && s->c == SORT20_C_BITS;
    s->device = ctx->device;
    // `table` keeps row 0 only (the points themselves, canonical saturated form: download, re-upload);
    // the window rows live in the 30-bit table built by srs_precompute.
    hipError_t e = hipMalloc((void **)&s->table, s->npad * sizeof(G1Affine));
    if (e != hipSuccess) {
        delete s;
        ctx->err = std::string("hipMalloc(SRS table): ") + hipGetErrorString(e);
        return KZG_ERR_ALLOC;
    }
    *out = s;
    return KZG_OK;
}

// the 30-bit rows that k_accum_affine gathers from (one thread per point, 2 multiplies)
__global__ __launch_bounds__(256) void k_table_to30(const G1Affine *src, G1Affine30 *dst, size_t npoints) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npoints) return;
    dst[i] = g1_affine_to30(src[i]);
}

// rows w = 1..W-1: row w = 2^c * row (w-1), computed in the saturated representation through two ping-pong
// row buffers and converted row by row into the resident 30-bit table.
int srs_precompute(kzg_ctx *ctx, kzg_srs *srs) {
    if (srs->n == 0) return KZG_OK;
    hipStream_t st = ctx->lanes[0].stream;
    const size_t n = srs->n;
    size_t npts = (size_t)srs->rows * srs->npad;
    hipError_t e30 = hipMalloc(&srs->table30, npts * sizeof(G1Affine30));
    if (e30 != hipSuccess) return fail(ctx, KZG_ERR_ALLOC, std::string("hipMalloc(SRS window table): ") + hipGetErrorString(e30));
    G1Affine30 *t30 = (G1Affine30 *)srs->table30;
    const size_t CHUNK = (size_t)1 << 20;
    size_t chunk = n < CHUNK ? n : CHUNK;
    G1Xyzz *tmp = nullptr;
    G1Affine *rows[2] = {nullptr, nullptr};
    KZG_HIP_CHECK(ctx, hipMalloc((void **)&tmp, chunk * sizeof(G1Xyzz)));
    KZG_HIP_CHECK(ctx, hipMalloc((void **)&rows[0], n * sizeof(G1Affine)));
    KZG_HIP_CHECK(ctx, hipMalloc((void **)&rows[1], n * sizeof(G1Affine)));
    unsigned gn = (unsigned)((n + 255) / 256);
    KZG_LAUNCH(ctx, st, "k_table_to30", k_table_to30, gn, 256, 0, srs->table, t30, n);
    const G1Affine *prev = srs->table;
    for (int w = 1; w < srs->rows; w++) {
        G1Affine *cur = rows[w & 1];
        for (size_t o = 0; o < n; o += chunk) {
            size_t m = n - o < chunk ? n - o : chunk;
            KZG_LAUNCH(ctx, st, "k_dbl_c", k_dbl_c, (unsigned)((m + 255) / 256), 256, 0, prev + o, tmp, m, srs->row_shift);
            KZG_TRY(batch_to_affine(ctx, st, tmp, cur + o, m));
        }
        KZG_LAUNCH(ctx, st, "k_table_to30", k_table_to30, gn, 256, 0, cur, t30 + (size_t)w * srs->npad, n);
        prev = cur;
    }
    KZG_HIP_CHECK(ctx, hipStreamSynchronize(st));
    KZG_HIP_CHECK(ctx, hipFree(tmp));
    KZG_HIP_CHECK(ctx, hipFree(rows[0]));
    KZG_HIP_CHECK(ctx, hipFree(rows[1]));
    return KZG_OK;
}

int srs_finish_from_xyzz(kzg_ctx *ctx, kzg_srs *srs, G1Xyzz *d_row0_xyzz) {
    hipStream_t st = ctx->lanes[0].stream;
    KZG_TRY(batch_to_affine(ctx, st, d_row0_xyzz, srs->table, srs->n));
    return srs_precompute(ctx, srs);
}

// ---------------------------------------------------------------------------------------------
// decoding of the four input point formats into row 0
// ---------------------------------------------------------------------------------------------
__device__ Fq read_be48(const uint8_t *src, bool mask_flags) {
    Fq r = Fq::zero();
    for (int i = 0; i < 48; i++) {
        uint32_t byte = src[47 - i];
        if (mask_flags && i == 47) byte &= 0x1f;
        r.v[i >> 2] |= byte << (8 * (i & 3));
    }
    return r;
}

// y = sqrt(a) = a^((q+1)/4) (q = 3 mod 4); caller checks y^2 == a
__device__ Fq fq_sqrt_candidate(const Fq &a) {
    // (q + 1) / 4
    constexpr uint32_t E[12] = {0xffffeaabu, 0xee7fbfffu, 0xac54ffffu, 0x07aaffffu, 0x3dac3d89u, 0xd9cc34a8u,
                                0x3ce144afu, 0xd91dd2e1u, 0x90d2eb35u, 0x92c6e9edu, 0x8e5ff9a6u, 0x0680447au};
    Fq acc = Fq::one();
    for (int i = 383; i >= 0; i--) {
        acc = sqr(acc);
        uint32_t limb = 0;
#pragma unroll
        for (int k = 0; k < 12; k++) limb = (k == (i >> 5)) ? E[k] : limb;
        if ((limb >> (i & 31)) & 1) acc = mul(acc, a);
    }
    return acc;
}

__device__ bool fq_gt_half(const Fq &canon) {
    constexpr uint32_t H[12] = {0xffffd555u, 0xdcff7fffu, 0x58a9ffffu, 0x0f55ffffu, 0x7b587b12u, 0xb3986950u,
                                0x79c2895fu, 0xb23ba5c2u, 0x21a5d66bu, 0x258dd3dbu, 0x1cbff34du, 0x0d0088f5u};
    for (int i = 11; i >= 0; i--) {
        if (canon.v[i] > H[i]) return true;
        if (canon.v[i] < H[i]) return false;
    }
    return false;
}

// [r]P == O: membership in the prime-order subgroup G1 (what G1Affine deserialisation checks upstream; the curve has cofactor
// (z - 1)^2 / 3, and everything downstream -- the MSM's r - k trick, the verifier's rewritten pairing equation -- is only valid
// for r-torsion points)
__device__ bool g1_in_subgroup(const G1Xyzz &p) {
    if (p.is_inf()) return true;
    G1Xyzz acc = G1Xyzz::inf();
    for (int i = 254; i >= 0; i--) {
        acc = g1_dbl(acc);
        uint32_t limb = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) limb = (k == (i >> 5)) ? FrParams::mod(k) : limb;
        if ((limb >> (i & 31)) & 1) acc = g1_add(acc, p);
    }
    return acc.is_inf();
}

// any input point format -> XYZZ; *bad |= 1 on a decode / canonical-limb / on-curve / subgroup failure (per `level`)
__global__ __launch_bounds__(256) void k_decode_points(const uint8_t *src, size_t n, int fmt, G1Xyzz *out, int *bad, int level) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (fmt == KZG_G1_AFFINE_MONT_96 || fmt == KZG_G1_JACOBIAN_MONT_144) {
        G1Xyzz p;
        bool ok = true;
        if (fmt == KZG_G1_AFFINE_MONT_96) {
            const G1Affine a = *reinterpret_cast<const G1Affine *>(src + i * 96);
            if (level >= POINTS_ON_CURVE) ok = is_canonical(a.x) && is_canonical(a.y) && g1_on_curve(a);
            p = G1Xyzz::from_affine(a);
        } else {
            const G1Jacobian j = *reinterpret_cast<const G1Jacobian *>(src + i * 144);
            if (level >= POINTS_ON_CURVE) {
                ok = is_canonical(j.x) && is_canonical(j.y) && is_canonical(j.z);
                if (ok && !j.z.is_zero()) {  // Y^2 = X^3 + 4 Z^6
                    const Fq z2 = sqr(j.z), z6 = mul(sqr(z2), z2);
                    Fq four_z6 = add(z6, z6);
                    four_z6 = add(four_z6, four_z6);
                    ok = sqr(j.y) == add(mul(sqr(j.x), j.x), four_z6);
                }
            }
            p = g1_from_jacobian(j);
        }
        if (ok && level >= POINTS_SUBGROUP) ok = g1_in_subgroup(p);
        if (!ok) {
            atomicOr(bad, 1);
            p = G1Xyzz::inf();
        }
        out[i] = p;
        return;
    }
    G1Affine a;
    bool ok = true;
    if (fmt == KZG_G1_ZCASH_UNCOMPRESSED_96) {
        const uint8_t *p = src + i * 96;
        if (p[0] & 0x80) ok = false;
        if (p[0] & 0x40) {
            a = G1Affine::inf();
        } else {
            Fq x = read_be48(p, true), y = read_be48(p + 48, false);
            ok = ok && is_canonical(x) && is_canonical(y);
            a.x = to_mont(x);
            a.y = to_mont(y);
            ok = ok && g1_on_curve(a);
        }
    } else {
        const uint8_t *p = src + i * 48;
        if (!(p[0] & 0x80)) ok = false;
        if (p[0] & 0x40) {
            a = G1Affine::inf();
        } else {
            Fq x = read_be48(p, true);
            ok = ok && is_canonical(x);
            a.x = to_mont(x);
            Fq rhs = add(mul(sqr(a.x), a.x), from_u64<FqParams>(4));
            Fq y = fq_sqrt_candidate(rhs);
            ok = ok && (sqr(y) == rhs);
            bool want_big = (p[0] & 0x20) != 0;
            if (fq_gt_half(from_mont(y)) != want_big) y = neg(y);
            a.y = y;
        }
    }
    G1Xyzz p = G1Xyzz::from_affine(a);
    if (ok && level >= POINTS_SUBGROUP) ok = g1_in_subgroup(p);
    if (!ok) {
        atomicOr(bad, 1);
        p = G1Xyzz::inf();
    }
    out[i] = p;
}

int decode_points(kzg_ctx *ctx, hipStream_t st, const void *d_raw, size_t n, int fmt, G1Xyzz *d_out, int *d_bad, int level) {
    if (!n) return KZG_OK;
    KZG_LAUNCH(ctx, st, "k_decode_points", k_decode_points, (unsigned)((n + 255) / 256), 256, 0, (const uint8_t *)d_raw, n,
               fmt, d_out, d_bad, level);
    return KZG_OK;
}

// ---------------------------------------------------------------------------------------------
// fixed-base multiplication [s]G with an 8-bit window table (32 x 255 affine points, built once)
// ---------------------------------------------------------------------------------------------
struct FixedBaseTable {
    G1Affine *table = nullptr;  // [32][255]: (d) * 2^(8j) * G at [j*255 + d-1]
};

__global__ __launch_bounds__(64) void k_fb_build(G1Xyzz *out) {
    int j = threadIdx.x;
    if (j >= 32) return;
    G1Xyzz base = G1Xyzz::from_affine(g1_generator());
    for (int k = 0; k < 8 * j; k++) base = g1_dbl(base);
    G1Xyzz acc = base;
    for (int d = 1; d <= 255; d++) {
        out[j * 255 + d - 1] = acc;
        acc = g1_add(acc, base);
    }
}

void fixed_base_free(kzg_ctx *ctx) {
    if (!ctx->fixed_base) return;
    if (ctx->fixed_base->table) hipFree(ctx->fixed_base->table);
    delete ctx->fixed_base;
    ctx->fixed_base = nullptr;
}

static int fixed_base_table(kzg_ctx *ctx, hipStream_t st, FixedBaseTable **out) {
    if (!ctx->fixed_base) {
        FixedBaseTable *t = new FixedBaseTable();
        G1Xyzz *tmp = nullptr;
        KZG_HIP_CHECK(ctx, hipMalloc((void **)&tmp, 32 * 255 * sizeof(G1Xyzz)));
        KZG_HIP_CHECK(ctx, hipMalloc((void **)&t->table, 32 * 255 * sizeof(G1Affine)));
        KZG_LAUNCH(ctx, st, "k_fb_build", k_fb_build, 1, 64, 0, tmp);
        KZG_TRY(batch_to_affine(ctx, st, tmp, t->table, 32 * 255));
        KZG_HIP_CHECK(ctx, hipStreamSynchronize(st));
        KZG_HIP_CHECK(ctx, hipFree(tmp));
        ctx->fixed_base = t;
    }
    *out = ctx->fixed_base;
    return KZG_OK;
}

__global__ __launch_bounds__(256) void k_fixed_base_mul(const Fr *scalars_mont, size_t n, const G1Affine *table,
                                                        G1Xyzz *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr s = from_mont(scalars_mont[i]);
    G1Xyzz acc = G1Xyzz::inf();
    for (int j = 0; j < 32; j++) {
        uint32_t limb = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) limb = (k == (j >> 2)) ? s.v[k] : limb;
        uint32_t byte = (limb >> (8 * (j & 3))) & 0xffu;
        if (byte) acc = g1_madd(acc, table[j * 255 + byte - 1]);
    }
    out[i] = acc;
}

int fixed_base_mul(kzg_ctx *ctx, hipStream_t stream, const Fr *d_scalars_mont, size_t n, G1Xyzz *d_out) {
    FixedBaseTable *t = nullptr;
    KZG_TRY(fixed_base_table(ctx, stream, &t));
    if (n == 0) return KZG_OK;
    KZG_LAUNCH(ctx, stream, "k_fixed_base_mul", k_fixed_base_mul, (unsigned)((n + 255) / 256), 256, 0, d_scalars_mont, n,
               t->table, d_out);
    return KZG_OK;
}

// scalars for setup(): s^i ; for the Lagrange basis: (s^d - 1) w^i / (d (s - w^i))
__global__ __launch_bounds__(256) void k_powers(Fr base, size_t first, size_t n, Fr *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = pow_u64(base, (uint64_t)(first + i));
}

__global__ __launch_bounds__(256) void k_lagrange_den(Fr s, Fr omega, Fr d_mont, size_t d, Fr *den) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d) return;
    den[i] = mul(d_mont, sub(s, pow_u64(omega, (uint64_t)i)));
}

__global__ __launch_bounds__(256) void k_lagrange_scalars(Fr zt, Fr omega, size_t d, const Fr *den, const Fr *den_inv,
                                                          Fr *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d) return;
    if (den[i].is_zero()) {  // s == w^i: L_i(s) = 1 (and zt = 0 makes every other L_j vanish)
        out[i] = Fr::one();
        return;
    }
    out[i] = mul(mul(zt, pow_u64(omega, (uint64_t)i)), den_inv[i]);
}

}  // namespace kzg

using namespace kzg;

// ---------------------------------------------------------------------------------------------
// C ABI: SRS objects
// ---------------------------------------------------------------------------------------------
static int load_host_scalar(kzg_ctx *ctx, const void *s, int sfmt, Fr *out_mont) {
    Fr v;
    memcpy(v.v, s, 32);
    if (sfmt == KZG_FR_CANONICAL_LE_32) {
        if (!is_canonical(v)) return fail(ctx, KZG_ERR_SHAPE, "scalar not canonical");
        v = to_mont(v);
    } else if (sfmt != KZG_FR_MONT_LE_32) {
        return fail(ctx, KZG_ERR_SHAPE, "unknown scalar format");
    }
    *out_mont = v;
    return KZG_OK;
}

extern "C" int kzg_srs_upload_g1(kzg_ctx *ctx, const void *pts, size_t n, int pfmt, kzg_srs **out) {
    if (!ctx || !out || (!pts && n)) return KZG_ERR_SHAPE;
    kzg::Guard g(ctx);
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    size_t psz = point_format_bytes(pfmt);
    if (!psz) return fail(ctx, KZG_ERR_SHAPE, "unknown G1 point format");
    kzg_srs *s = nullptr;
    KZG_TRY(srs_alloc(ctx, n, &s));
    hipStream_t st = ctx->lanes[0].stream;
    int rc = KZG_OK;
    if (n) {
        // every format goes through the decoder: canonical limbs, on the curve, in the subgroup (option trusted_points = 1
        // skips the subgroup test), as G1Affine / G1Projective deserialisation guarantees upstream
        uint8_t *raw = nullptr;
        G1Xyzz *tmp = nullptr;
        int *bad = nullptr;
        int hbad = 0;
        if (hipMalloc((void **)&raw, n * psz) != hipSuccess || hipMalloc((void **)&tmp, n * sizeof(G1Xyzz)) != hipSuccess ||
            hipMalloc((void **)&bad, sizeof(int)) != hipSuccess) {
            rc = fail(ctx, KZG_ERR_ALLOC, "hipMalloc(decode staging)");
        } else {
            hipMemcpyAsync(raw, pts, n * psz, hipMemcpyHostToDevice, st);
            hipMemsetAsync(bad, 0, sizeof(int), st);
            rc = decode_points(ctx, st, raw, n, pfmt, tmp, bad, untrusted_level(ctx));
            hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, st);
            hipStreamSynchronize(st);
            if (rc == KZG_OK && hbad)
                rc = fail(ctx, KZG_ERR_BAD_POINT, "a G1 point failed to decode, is not on the curve or not in the r-torsion subgroup");
            if (rc == KZG_OK) rc = srs_finish_from_xyzz(ctx, s, tmp);
        }
        if (raw) hipFree(raw);
        if (tmp) hipFree(tmp);
        if (bad) hipFree(bad);
    }
    if (rc != KZG_OK) {
        hipFree(s->table);
        delete s;
        return rc;
    }
    *out = s;
    return KZG_OK;
}

static int srs_from_scalars(kzg_ctx *ctx, kzg_srs *s, Fr *d_scalars) {
    hipStream_t st = ctx->lanes[0].stream;
    G1Xyzz *tmp = nullptr;
    KZG_HIP_CHECK(ctx, hipMalloc((void **)&tmp, s->n * sizeof(G1Xyzz)));
    int rc = fixed_base_mul(ctx, st, d_scalars, s->n, tmp);
    if (rc == KZG_OK) rc = srs_finish_from_xyzz(ctx, s, tmp);
    hipStreamSynchronize(st);
    hipFree(tmp);
    return rc;
}

// scalar generators shared with the G2 SRS construction (pairing.hip): Montgomery-form outputs on the device
namespace kzg {
int powers_run(kzg_ctx *ctx, hipStream_t st, const Fr &base_mont, size_t first, size_t n, Fr *d_out) {
    if (n) KZG_LAUNCH(ctx, st, "k_powers", k_powers, (unsigned)((n + 255) / 256), 256, 0, base_mont, first, n, d_out);
    return KZG_OK;
}

// L_i(tau) = (tau^d - 1) w^i / (d (tau - w^i)); a tau on the domain gives the indicator vector
int lagrange_scalars_run(kzg_ctx *ctx, hipStream_t st, const Fr &tau, size_t d, Fr *d_out) {
    uint32_t exp = (uint32_t)ilog2_ceil(d);
    Fr omega = host_omega(exp);
    Fr zt = sub(pow_u64(tau, (uint64_t)d), Fr::one());
    Fr dm = from_u64<FrParams>((uint64_t)d);
    Fr *den = nullptr, *deni = nullptr;
    int rc = KZG_OK;
    if (hipMalloc((void **)&den, d * sizeof(Fr)) != hipSuccess || hipMalloc((void **)&deni, d * sizeof(Fr)) != hipSuccess)
        rc = fail(ctx, KZG_ERR_ALLOC, "hipMalloc(lagrange scalars)");
    if (rc == KZG_OK) {
        unsigned grid = (unsigned)((d + 255) / 256);
        KZG_LAUNCH(ctx, st, "k_lagrange_den", k_lagrange_den, grid, 256, 0, tau, omega, dm, d, den);
        rc = batch_inverse(ctx, st, den, deni, d);
        if (rc == KZG_OK) KZG_LAUNCH(ctx, st, "k_lagrange_scalars", k_lagrange_scalars, grid, 256, 0, zt, omega, d, den, deni, d_out);
    }
    hipStreamSynchronize(st);
    if (den) hipFree(den);
    if (deni) hipFree(deni);
    return rc;
}
}  // namespace kzg

extern "C" int kzg_srs_setup_g1(kzg_ctx *ctx, const void *sec, int sfmt, size_t n, kzg_srs **out) {
    return kzg_srs_setup_g1_shard(ctx, sec, sfmt, 0, n, out);
}

extern "C" int kzg_srs_setup_g1_shard(kzg_ctx *ctx, const void *sec, int sfmt, size_t first, size_t n, kzg_srs **out) {
    if (!ctx || !out || !sec) return KZG_ERR_SHAPE;
    kzg::Guard g(ctx);
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    Fr tau;
    KZG_TRY(load_host_scalar(ctx, sec, sfmt, &tau));
    kzg_srs *s = nullptr;
    KZG_TRY(srs_alloc(ctx, n, &s));
    int rc = KZG_OK;
    if (n) {
        hipStream_t st = ctx->lanes[0].stream;
        Fr *sc = nullptr;
        if (hipMalloc((void **)&sc, n * sizeof(Fr)) != hipSuccess) rc = fail(ctx, KZG_ERR_ALLOC, "hipMalloc(setup scalars)");
        if (rc == KZG_OK) {
            KZG_LAUNCH(ctx, st, "k_powers", k_powers, (unsigned)((n + 255) / 256), 256, 0, tau, first, n, sc);
            rc = srs_from_scalars(ctx, s, sc);
        }
        if (sc) hipFree(sc);
    }
    if (rc != KZG_OK) {
        hipFree(s->table);
        delete s;
        return rc;
    }
    *out = s;
    return KZG_OK;
}

extern "C" int kzg_srs_setup_lagrange_g1(kzg_ctx *ctx, const void *sec, int sfmt, size_t d, kzg_srs **out) {
    if (!ctx || !out || !sec) return KZG_ERR_SHAPE;
    kzg::Guard g(ctx);
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (d == 0 || (d & (d - 1))) return fail(ctx, KZG_ERR_SHAPE, "Lagrange basis needs a power-of-two size (src/eval_form.rs:255-256)");
    uint32_t exp = (uint32_t)ilog2_ceil(d);
    if (exp >= FR_TWO_ADICITY) return fail(ctx, KZG_ERR_DEGREE_TOO_LARGE, "domain too large");
    Fr tau;
    KZG_TRY(load_host_scalar(ctx, sec, sfmt, &tau));
    kzg_srs *s = nullptr;
    KZG_TRY(srs_alloc(ctx, d, &s));
    hipStream_t st = ctx->lanes[0].stream;
    Fr *den = nullptr, *deni = nullptr, *sc = nullptr;
    int rc = KZG_OK;
    if (hipMalloc((void **)&sc, d * sizeof(Fr)) != hipSuccess) rc = fail(ctx, KZG_ERR_ALLOC, "hipMalloc(lagrange scalars)");
    if (rc == KZG_OK) rc = lagrange_scalars_run(ctx, st, tau, d, sc);
    if (rc == KZG_OK) rc = srs_from_scalars(ctx, s, sc);
    hipStreamSynchronize(st);
    if (den) hipFree(den);
    if (deni) hipFree(deni);
    if (sc) hipFree(sc);
    if (rc != KZG_OK) {
        hipFree(s->table);
        delete s;
        return rc;
    }
    *out = s;
    return KZG_OK;
}

extern "C" int kzg_srs_download_g1(kzg_ctx *ctx, const kzg_srs *srs, size_t offset, size_t n, void *out, int pfmt) {
    if (!ctx || !srs || (!out && n)) return KZG_ERR_SHAPE;
    kzg::Guard g(ctx);
    KZG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (n > srs->n || offset > srs->n - n) return fail(ctx, KZG_ERR_SHAPE, "SRS download range out of bounds");
    if (n == 0) return KZG_OK;
    if (pfmt != KZG_G1_AFFINE_MONT_96) return fail(ctx, KZG_ERR_SHAPE, "SRS download supports KZG_G1_AFFINE_MONT_96 only");
    hipStream_t st = ctx->lanes[0].stream;
    KZG_HIP_CHECK(ctx, hipMemcpyAsync(out, srs->table + offset, n * 96, hipMemcpyDeviceToHost, st));
    KZG_HIP_CHECK(ctx, hipStreamSynchronize(st));
    return KZG_OK;
}

extern "C" size_t kzg_srs_len(const kzg_srs *srs) { return srs ? srs->n : 0; }

extern "C" int kzg_srs_footprint(size_t n, int window_bits, int window_rows, size_t *bytes) {
    if (!bytes || (window_bits != 0 && (window_bits < 4 || window_bits > 20)) || window_rows < 0) return KZG_ERR_SHAPE;
    int c, W, rows;
    bool n17;
    int naf = 0;
    srs_shape(window_bits, window_rows, n, &c, &W, &rows, &n17, 0, &naf);  // the default policy: window tables
    const size_t npad = n ? n : 1;
    *bytes = npad * (sizeof(G1Affine) + (size_t)rows * sizeof(G1Affine30));
    return KZG_OK;
}

extern "C" int kzg_srs_table_rows(const kzg_srs *srs) { return srs ? srs->rows : 0; }

extern "C" int kzg_srs_window_info(const kzg_srs *srs, int *window_bits, int *windows) {
    if (!srs) return KZG_ERR_SHAPE;
    if (window_bits) *window_bits = srs->naf ? srs->naf : srs->c;  // positional tables: the NAF width (18)
    if (windows) *windows = srs->W;
    return KZG_OK;
}

extern "C" void kzg_srs_free(kzg_ctx *ctx, kzg_srs *srs) {
    if (!srs) return;
    if (ctx) {
        kzg::Guard g(ctx);
        hipSetDevice(ctx->device);
        for (auto &l : ctx->lanes) hipStreamSynchronize(l.stream);
    }
    if (srs->table) hipFree(srs->table);
    if (srs->table30) hipFree(srs->table30);
    delete srs;
}
