// mad_issue.hip -- what limits v_mad_i64_i32 issue at LOW occupancy (k_accum_affine holds 2 waves per SIMD)?
//   variants: number of independent accumulator chains per wave (1, 2, 3, 4, 8), carry-out destination (always vcc, or rotating
//   over SGPR pairs), and column-shaped streams with the non-mad instructions of mul30: one multiply alone (a single dependent
//   chain, as mul30_gfx950.inc emits it) against two and three independent multiplies interleaved instruction by instruction.
//   Loop bodies hold 96..192 mads so that the loop branch does not pace a lone wave.  Waves per SIMD 1, 2, 3, 4, 8.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/mad_issue tools/mad_issue.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

#define MAD_VCC(c) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b) : "vcc")
#define MAD_S(c, lo, hi) asm volatile("v_mad_i64_i32 %0, s[" #lo ":" #hi "], %1, %2, %0" : "+v"(c) : "v"(a), "v"(b) : "s" #lo, "s" #hi)

template <int CHAINS, int ROT>
__global__ __launch_bounds__(256) void k_mad(uint32_t *out, int iters, uint32_t seed) {
    int32_t a = (int32_t)(seed + threadIdx.x), b = (int32_t)(seed * 3 + blockIdx.x);
    uint64_t c[8];
    for (int k = 0; k < 8; k++) c[k] = a * (k + 1);
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 12; u++) {
            if (!ROT) {
                MAD_VCC(c[0 % CHAINS]); MAD_VCC(c[1 % CHAINS]); MAD_VCC(c[2 % CHAINS]); MAD_VCC(c[3 % CHAINS]);
                MAD_VCC(c[4 % CHAINS]); MAD_VCC(c[5 % CHAINS]); MAD_VCC(c[6 % CHAINS]); MAD_VCC(c[7 % CHAINS]);
            } else {
                MAD_S(c[0 % CHAINS], 20, 21); MAD_S(c[1 % CHAINS], 22, 23); MAD_S(c[2 % CHAINS], 24, 25); MAD_S(c[3 % CHAINS], 26, 27);
                MAD_S(c[4 % CHAINS], 28, 29); MAD_S(c[5 % CHAINS], 30, 31); MAD_S(c[6 % CHAINS], 36, 37); MAD_S(c[7 % CHAINS], 38, 39);
            }
        }
    }
    uint64_t s = 0;
    for (int k = 0; k < 8; k++) s ^= c[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}

// NMUL independent multiplies in flight, each the stream of mul30's low-half column: a chain of 13 mads, quotient digit
// (v_mul_lo_u32, sign extension), one more mad, 64-bit shift.  The chains are interleaved mad by mad.
template <int NMUL, int ROT>
__global__ __launch_bounds__(256) void k_columns(uint32_t *out, int iters, uint32_t seed) {
    int32_t a = (int32_t)(seed + threadIdx.x), b = (int32_t)(seed * 3 + blockIdx.x);
    uint64_t acc[3];
    for (int k = 0; k < 3; k++) acc[k] = a * (k + 1);
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int col = 0; col < 4; col++) {
#pragma unroll
            for (int k = 0; k < 13; k++) {
                if (!ROT) {
                    MAD_VCC(acc[0]);
                    if (NMUL > 1) MAD_VCC(acc[1]);
                    if (NMUL > 2) MAD_VCC(acc[2]);
                } else {
                    MAD_S(acc[0], 20, 21);
                    if (NMUL > 1) MAD_S(acc[1], 22, 23);
                    if (NMUL > 2) MAD_S(acc[2], 24, 25);
                }
            }
#pragma unroll
            for (int j = 0; j < NMUL; j++) {
                uint32_t m = (uint32_t)acc[j] * 0x12345679u;
                int32_t ms = (int32_t)(m << 2) >> 2;
                asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[j]) : "v"(ms), "v"(b) : "vcc");
                acc[j] = (uint64_t)((int64_t)acc[j] >> 30);
            }
        }
    }
    uint64_t s = acc[0] ^ acc[1] ^ acc[2];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}

template <class K>
double time_kernel(K kern, dim3 grid, dim3 block, int reps, uint32_t *out, int iters) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, grid, block, 0, 0, out, iters, 7u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(kern, grid, block, 0, 0, out, iters, 7u);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    int cus = p.multiProcessorCount;
    uint32_t *out; CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    const int iters = 3500;  // x 96 mads per lane: ~3 ms at 2 waves/SIMD, long enough for the clock to settle
    const int wl[] = {1, 2, 3, 4, 8};
    printf("v_mad_i64_i32, %d CUs; T lane-mad/s and nominal cycles per wave-instruction per SIMD at 2.4 GHz\n", cus);
    for (int wi = 0; wi < 5; wi++) {
        const int w = wl[wi];
        dim3 grid(cus * w), block(256);
        struct { const char *name; double ms; } r[] = {
            {"8 chains, vcc", time_kernel(k_mad<8, 0>, grid, block, 3, out, iters)},
            {"8 chains, rotating sdst", time_kernel(k_mad<8, 1>, grid, block, 3, out, iters)},
            {"4 chains, vcc", time_kernel(k_mad<4, 0>, grid, block, 3, out, iters)},
            {"4 chains, rotating sdst", time_kernel(k_mad<4, 1>, grid, block, 3, out, iters)},
            {"3 chains, vcc", time_kernel(k_mad<3, 0>, grid, block, 3, out, iters)},
            {"2 chains, vcc", time_kernel(k_mad<2, 0>, grid, block, 3, out, iters)},
            {"2 chains, rotating sdst", time_kernel(k_mad<2, 1>, grid, block, 3, out, iters)},
            {"1 chain, vcc", time_kernel(k_mad<1, 0>, grid, block, 3, out, iters)},
            {"1 chain, rotating sdst", time_kernel(k_mad<1, 1>, grid, block, 3, out, iters)},
        };
        for (auto &x : r) {
            double ops = (double)cus * w * 256 * iters * 96;
            printf("waves/SIMD %d  %-26s %8.3f ms  %7.2f T lane-mad/s  %6.2f cyc\n", w, x.name, x.ms, ops / x.ms / 1e9,
                   x.ms * 1e-3 * 2.4e9 / ((double)iters * 96 * w));
        }
        const int it2 = 3000;
        struct { const char *name; int nm; double ms; } c[] = {
            {"columns: 1 multiply alone, vcc", 1, time_kernel(k_columns<1, 0>, grid, block, 3, out, it2)},
            {"columns: 2 multiplies interleaved, vcc", 2, time_kernel(k_columns<2, 0>, grid, block, 3, out, it2 / 2)},
            {"columns: 2 interleaved, rotating sdst", 2, time_kernel(k_columns<2, 1>, grid, block, 3, out, it2 / 2)},
            {"columns: 3 multiplies interleaved, vcc", 3, time_kernel(k_columns<3, 0>, grid, block, 3, out, it2 / 3)},
            {"columns: 3 interleaved, rotating sdst", 3, time_kernel(k_columns<3, 1>, grid, block, 3, out, it2 / 3)},
        };
        for (auto &x : c) {
            double mads = (double)cus * w * 256 * (it2 / x.nm) * 4 * 14 * x.nm;
            printf("waves/SIMD %d  %-40s %8.3f ms  %7.2f T lane-mad/s\n", w, x.name, x.ms, mads / x.ms / 1e9);
        }
    }
    return 0;
}
