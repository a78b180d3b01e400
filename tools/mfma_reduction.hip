// mfma_reduction.hip -- VERDICT r2 item 6: would the idle matrix pipe pay for the constant-operand half of the Montgomery
// reduction?  One reduction step of the 13 x 30-bit multiply is  t = (T + m*q) / R  with q constant, so the m*q products
// (169 of a multiply's 338 v_mad_i64_i32) are a product with a CONSTANT matrix: the Toeplitz matrix of q's digits.  On the
// matrix pipe the widest exact integer form is v_mfma_i32_32x32x32_i8, i.e. 8-bit digits: m becomes 49..52 signed bytes, q's
// Toeplitz matrix is 52 x ~100 bytes (banded: 6 of the 8 32x32 tiles are non-zero), the product comes back as ~100 byte
// columns of 22-bit sums per element, which must be carry-recombined into 13 limbs.  Element e of a wave lives in lane e
// (VALU layout); the 32x32x32 MFMA wants 16 consecutive k-bytes of a row per lane, rows = lane % 32, so operands and results
// cross the two lane halves with v_permlane32_swap.
//
// This microbenchmark prices the INSTRUCTION STREAMS (same opcodes, same counts, same dependencies and register traffic as a
// real implementation; the numeric values are not checked -- a rejection needs the cost, not the bits):
//   A  valu_full      one full multiply the way mul30 does it: 338 v_mad_i64_i32 + 26 64-bit shifts + 13 x (mul_lo, ashr) + 13 bfe
//   B  valu_half      only the a*b half (what would stay on the VALU): 169 mads + 26 shifts + 13 digit extractions
//   C  mfma_half      only the m*q half on the matrix pipe: byte split (13 limbs -> 13 words, xor to signed bytes), 8 lane-half
//                     swaps in, 12 MFMAs (two batches of 32 elements x 6 tiles), 32 swaps out, recombination of 100 byte columns
//                     into 13 limbs (shift-adds) and the carry pass
//   D  co_issue       B and C interleaved in one wave (the candidate kernel): does the MFMA work hide under the mads?
// at 2 waves per SIMD (k_accum_affine's occupancy) and, for reference, 1 and 4.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_reduction tools/mfma_reduction.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

#define MAD(acc, x, y) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y) : "vcc")

// the VALU stream of `NP` product halves (a*b and/or m*q) of one multiply: 13-limb product scan
template <int FULL>
__device__ __forceinline__ void valu_mul(int32_t (&a)[13], int32_t (&b)[13], int32_t (&r)[13], int32_t qinv, int32_t q0) {
    uint64_t acc = 0;
    int32_t m[13];
#pragma unroll
    for (int k = 0; k < 13; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) MAD(acc, a[i], b[k - i]);
        if (FULL) {
#pragma unroll
            for (int i = 0; i < k; i++) MAD(acc, m[i], b[(k - i + 5) % 13]);
            m[k] = (int32_t)((uint32_t)acc * (uint32_t)qinv) >> 2;
            MAD(acc, m[k], q0);
        } else {
            m[k] = (int32_t)((uint32_t)acc << 2) >> 2;   // the digit that goes to the byte split
        }
        acc = (uint64_t)((int64_t)acc >> 30);
    }
#pragma unroll
    for (int k = 13; k < 25; k++) {
#pragma unroll
        for (int i = k - 12; i < 13; i++) {
            MAD(acc, a[i], b[k - i]);
            if (FULL) MAD(acc, m[i], b[(k - i + 5) % 13]);
        }
        r[k - 13] = (int32_t)((uint32_t)acc << 2) >> 2;
        acc = (uint64_t)((int64_t)(acc + (1ull << 29)) >> 30);
    }
    r[12] = (int32_t)acc;
    if (!FULL) {
#pragma unroll
        for (int k = 0; k < 13; k++) a[k] = m[k];   // low half out (the T_lo digits the quotient is computed from)
    }
}

// the matrix-pipe stream for the m*q half of 64 elements (one per lane).  t[13]: low-half digits (30-bit limbs); r[13] += result
__device__ __forceinline__ void mfma_half(int32_t (&t)[13], int32_t (&r)[13], const v4i (&Bq)[6]) {
    // (1) byte split: 13 x 30-bit limbs -> 13 x 32-bit words of the same integer (funnel shifts), then to signed bytes
    uint32_t w[16];
#pragma unroll
    for (int j = 0; j < 12; j++) {
        const int bit = 32 * j, lo = bit / 30, sh = bit % 30;
        w[j] = ((uint32_t)t[lo] >> sh) | ((uint32_t)t[lo + 1] << (30 - sh)) | (sh > 2 && lo + 2 < 13 ? (uint32_t)t[lo + 2] << (60 - sh) : 0u);
    }
    w[12] = (uint32_t)t[12] >> 24;
    w[13] = w[14] = w[15] = 0;
#pragma unroll
    for (int j = 0; j < 13; j++) w[j] ^= 0x80808080u;  // unsigned bytes -> signed (the constant correction lives in the accumulator init)
    // (2) operands of the two 32-row batches: lane l < 32 supplies k-bytes 0..15 of row l, lane l + 32 bytes 16..31 of row l
    //     words 0..3 | 4..7 (k-tile 0) and 8..11 | 12..15 (k-tile 1): swap the upper half of the first with the lower of the second
    v4i A0[2], A1[2];  // [batch]
#pragma unroll
    for (int c = 0; c < 4; c++) {
        v2u s0 = __builtin_amdgcn_permlane32_swap(w[c], w[4 + c], false, false);
        v2u s1 = __builtin_amdgcn_permlane32_swap(w[8 + c], w[12 + c], false, false);
        A0[0][c] = (int)s0[0]; A0[1][c] = (int)s0[1];
        A1[0][c] = (int)s1[0]; A1[1][c] = (int)s1[1];
    }
    // (3) 2 batches x 6 MFMAs: k-tile 0 feeds column tiles 0..2, k-tile 1 feeds 1..3 (banded Toeplitz)
    v16i C[2][4];
#pragma unroll
    for (int bt = 0; bt < 2; bt++) {
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
#pragma unroll
            for (int e = 0; e < 16; e++) C[bt][nt][e] = 0;
        C[bt][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(Bq[0], A0[bt], C[bt][0], 0, 0, 0);
        C[bt][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(Bq[1], A0[bt], C[bt][1], 0, 0, 0);
        C[bt][2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(Bq[2], A0[bt], C[bt][2], 0, 0, 0);
        C[bt][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(Bq[3], A1[bt], C[bt][1], 0, 0, 0);
        C[bt][2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(Bq[4], A1[bt], C[bt][2], 0, 0, 0);
        C[bt][3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(Bq[5], A1[bt], C[bt][3], 0, 0, 0);
    }
    // (4) results back to one element per lane: with the digit index as the MFMA row, lane l holds 16 of the 32 digits of
    //     element l % 32 per tile; swapping the halves of batch 0 / batch 1 gives every lane all 32 digits of ITS element
    int32_t col[128];
#pragma unroll
    for (int nt = 0; nt < 4; nt++)
#pragma unroll
        for (int e = 0; e < 16; e++) {
            v2u s = __builtin_amdgcn_permlane32_swap((unsigned)C[0][nt][e], (unsigned)C[1][nt][e], false, false);
            col[32 * nt + 2 * e] = (int32_t)s[0];
            col[32 * nt + 2 * e + 1] = (int32_t)s[1];
        }
    // (5) recombination: limb i of the upper half gathers byte columns 49 + (30 i .. 30 i + 29) / 8 -> 4..5 columns each, then a
    //     carry pass over the 13 limbs
    int64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) {
        const int bit = 390 + 30 * i, c0 = bit / 8, sh = bit % 8;
        int64_t v = carry + ((int64_t)col[c0] >> sh);
#pragma unroll
        for (int j = 1; j <= 4; j++)
            if (c0 + j < 104) v += (int64_t)col[c0 + j] << (8 * j - sh);
        const int32_t d = (int32_t)((uint32_t)v << 2) >> 2;
        r[i] += d;
        carry = (v + (1ll << 29)) >> 30;
    }
    r[12] += (int32_t)carry;
}

template <int V>
__global__ __launch_bounds__(256, 2) void k_red(int32_t *out, int iters, int32_t seed) {
    int32_t a[13], b[13], r[13];
#pragma unroll
    for (int i = 0; i < 13; i++) {
        a[i] = (int32_t)((uint32_t)(seed * (i + 3) + threadIdx.x * 77) << 2) >> 2;
        b[i] = (int32_t)((uint32_t)(seed * (i + 11) + blockIdx.x * 131) << 2) >> 2;
        r[i] = 0;
    }
    v4i Bq[6];
#pragma unroll
    for (int t = 0; t < 6; t++)
#pragma unroll
        for (int c = 0; c < 4; c++) Bq[t][c] = seed * (17 * t + c + 1) + (int)threadIdx.x;
    for (int it = 0; it < iters; it++) {
        if (V == 0) valu_mul<1>(a, b, r, 0x12345679, seed);
        if (V == 1) valu_mul<0>(a, b, r, 0x12345679, seed);
        if (V == 2) mfma_half(a, r, Bq);
        if (V == 3) {
            valu_mul<0>(a, b, r, 0x12345679, seed);
            mfma_half(a, r, Bq);
        }
#pragma unroll
        for (int i = 0; i < 13; i++) {   // next round's operands depend on this round's result (a dependent chain, like the curve formulas)
            a[i] = V == 2 ? a[i] ^ (r[i] & 1) : b[i];
            b[i] = r[i];
        }
    }
    int32_t x = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) x ^= r[i] ^ a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

template <class K>
double time_kernel(K kern, dim3 grid, dim3 block, int reps, int32_t *out, int iters) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, grid, block, 0, 0, out, iters, 12345);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(kern, grid, block, 0, 0, out, iters, 12345);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    int cus = p.multiProcessorCount;
    int32_t *out; CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    const int iters = 1500;
    const char *names[] = {"A valu_full  (338 mads: a*b and m*q on the VALU, as mul30)", "B valu_half  (169 mads: a*b only)",
                           "C mfma_half  (m*q on the matrix pipe incl. byte split + recombination)", "D co_issue   (B + C in one wave)"};
    for (int w : {2, 1, 4}) {
        dim3 grid(cus * w), block(256);
        double ms[4];
        ms[0] = time_kernel(k_red<0>, grid, block, 3, out, iters);
        ms[1] = time_kernel(k_red<1>, grid, block, 3, out, iters);
        ms[2] = time_kernel(k_red<2>, grid, block, 3, out, iters);
        ms[3] = time_kernel(k_red<3>, grid, block, 3, out, iters);
        printf("waves/SIMD %d  (nominal cycles per multiply per wave-slot of a SIMD at 2.4 GHz; G multiplies/s over the chip)\n", w);
        for (int v = 0; v < 4; v++)
            printf("  %-78s %8.3f ms  %8.0f cyc  %7.2f G/s\n", names[v], ms[v], ms[v] * 1e-3 * 2.4e9 / ((double)iters * w),
                   (double)cus * w * 256 * iters / ms[v] / 1e6);
        printf("  => multiply with the reduction on the matrix pipe (D) takes %.2f x the VALU multiply (A)\n", ms[3] / ms[0]);
    }
    return 0;
}
