// microbench.hip -- instruction-issue rates that bound the MSM on gfx950 (run on the GPU box):
//   v_mad_u64_u32 (the Montgomery-multiply workhorse), v_mul_lo/hi_u32, v_mul_u32_u24, v_fma_f64,
//   v_lshl_add_u64, and the library's Fq / Fr multiply as compiled, at 1..8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/microbench tools/microbench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../kzg_amd/csrc/curve.h"
#include "../kzg_amd/csrc/field30.h"
using namespace kzg;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k_rate(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
    uint64_t c0 = a, c1 = b, c2 = a ^ b, c3 = a + b, c4 = a * 3, c5 = b * 5, c6 = a * 7, c7 = b * 9;
    double d0 = a, d1 = b, d2 = a + 1, d3 = b + 1, d4 = a + 2, d5 = b + 2, d6 = a + 3, d7 = b + 3, dm = 1.0000001, da = 0.5;
    for (int i = 0; i < iters; i++) {
        if (OP == 0) {
#define M(c) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b) : "vcc");
            M(c0) M(c1) M(c2) M(c3) M(c4) M(c5) M(c6) M(c7)
#undef M
        } else if (OP == 1) {
#define M(c) { uint32_t lo = (uint32_t)c; asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(a)); c = lo; }
            M(c0) M(c1) M(c2) M(c3) M(c4) M(c5) M(c6) M(c7)
#undef M
        } else if (OP == 2) {
#define M(c) { uint32_t lo = (uint32_t)c; asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo) : "v"(a)); c = lo; }
            M(c0) M(c1) M(c2) M(c3) M(c4) M(c5) M(c6) M(c7)
#undef M
        } else if (OP == 3) {
#define M(c) { uint32_t lo = (uint32_t)c; asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(lo) : "v"(a)); c = lo; }
            M(c0) M(c1) M(c2) M(c3) M(c4) M(c5) M(c6) M(c7)
#undef M
        } else if (OP == 4) {
#define M(d) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d) : "v"(dm), "v"(da));
            M(d0) M(d1) M(d2) M(d3) M(d4) M(d5) M(d6) M(d7)
#undef M
        } else if (OP == 5) {
#define M(c) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c) : "v"(c7));
            M(c0) M(c1) M(c2) M(c3) M(c4) M(c5) M(c6)
            asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c7) : "v"(c0));
#undef M
        } else if (OP == 6) {
#define M(c) { uint32_t lo = (uint32_t)c; asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(a)); c = lo; }
            M(c0) M(c1) M(c2) M(c3) M(c4) M(c5) M(c6) M(c7)
#undef M
        } else if (OP == 8) {  // the instruction mix of mul30: mostly v_mad_i64_i32 with a few 64-bit shifts / adds and 32-bit ops
#define M(c) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b) : "vcc");
            M(c0) M(c1) M(c2) M(c3) M(c4) M(c5)
#undef M
            asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c6) : "v"(c0));
            asm volatile("v_ashrrev_i64 %0, 30, %0" : "+v"(c7));
        } else if (OP == 7) {
#define M(c) { uint32_t lo = (uint32_t)c; asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(lo) : "v"(a)); c = lo; }
            M(c0) M(c1) M(c2) M(c3) M(c4) M(c5) M(c6) M(c7)
#undef M
        }
    }
    uint64_t s = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
    double ds = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32) ^ (uint32_t)ds;
}

template <class F>
__global__ __launch_bounds__(256) void k_fmul(F *out, const F *in, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    F x = in[i], y = in[i + 1];
    for (int k = 0; k < iters; k++) { F z = mul(x, y); x = y; y = z; }
    out[i] = y;
}

// k_fmul30x<0>: the portable C multiply as hipcc schedules it; <1>: the generated single-chain version (mul30_gfx950.inc)
template <int CHAIN>
__global__ __launch_bounds__(256) void k_fmul30x(int32_t *out, int iters, int32_t seed) {
    Fq30 x, y;
    for (int i = 0; i < F30_N; i++) { x.v[i] = sext30((uint32_t)(seed * (i + 3) + threadIdx.x * 77)); y.v[i] = sext30((uint32_t)(seed * (i + 11) + blockIdx.x * 131)); }
    x.v[F30_N - 1] >>= 12; y.v[F30_N - 1] >>= 12;
    for (int k = 0; k < iters; k++) { Fq30 z = CHAIN ? mul30(x, y) : mul30_inline(x, y); x = y; y = z; }  // mul30 = the generated version on the device
    int32_t t = 0;
    for (int i = 0; i < F30_N; i++) t ^= y.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

// chain of dependent mul30 (the MSM multiply, inlined), seeded with normalised limbs
__global__ __launch_bounds__(256) void k_fmul30(int32_t *out, int iters, int32_t seed) {
    Fq30 x, y;
    for (int i = 0; i < F30_N; i++) { x.v[i] = sext30((uint32_t)(seed * (i + 3) + threadIdx.x * 77)); y.v[i] = sext30((uint32_t)(seed * (i + 11) + blockIdx.x * 131)); }
    x.v[F30_N - 1] >>= 12; y.v[F30_N - 1] >>= 12;
    for (int k = 0; k < iters; k++) { Fq30 z = mul30_inline(x, y); x = y; y = z; }
    int32_t t = 0;
    for (int i = 0; i < F30_N; i++) t ^= y.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

__global__ __launch_bounds__(256) void k_madd(G1Xyzz *out, const G1Affine *pts, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    G1Xyzz acc = G1Xyzz::from_affine(pts[i & 4095]);
    for (int k = 0; k < iters; k++) acc = g1_madd(acc, pts[(i + k + 1) & 4095]);
    out[i] = acc;
}

__global__ __launch_bounds__(64) void k_init_points(G1Affine *pts, int n) {  // pts[i] = (i+1) * G via repeated madd in one thread
    if (threadIdx.x || blockIdx.x) return;
    G1Affine g = g1_generator();
    G1Xyzz acc = G1Xyzz::from_affine(g);
    for (int i = 0; i < n; i++) { pts[i] = g1_to_affine(acc); acc = g1_madd(acc, g); }
}

template <class K, class... A>
double time_kernel(K kern, dim3 grid, dim3 block, int reps, A... args) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, grid, block, 0, 0, args...);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(kern, grid, block, 0, 0, args...);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    int cus = p.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", p.name, cus, p.clockRate);
    uint32_t *out; CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4 * 8));
    const char *names[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24", "v_fma_f64", "v_lshl_add_u64", "v_add_u32", "v_mad_u32_u24", "mix 6 mad_i64:2 x64"};
    int iters = 4096;
    const int wlist[] = {1, 2, 3, 4, 6, 8};
    for (int wi = 0; wi < 6; wi++) {  // waves per SIMD = blocks of 256 threads per CU
        const int wps = wlist[wi];
        dim3 grid(cus * wps), block(256);
        double ms[9];
        ms[0] = time_kernel(k_rate<0>, grid, block, 5, out, iters, 7u);
        ms[1] = time_kernel(k_rate<1>, grid, block, 5, out, iters, 7u);
        ms[2] = time_kernel(k_rate<2>, grid, block, 5, out, iters, 7u);
        ms[3] = time_kernel(k_rate<3>, grid, block, 5, out, iters, 7u);
        ms[4] = time_kernel(k_rate<4>, grid, block, 5, out, iters, 7u);
        ms[5] = time_kernel(k_rate<5>, grid, block, 5, out, iters, 7u);
        ms[6] = time_kernel(k_rate<6>, grid, block, 5, out, iters, 7u);
        ms[7] = time_kernel(k_rate<7>, grid, block, 5, out, iters, 7u);
        ms[8] = time_kernel(k_rate<8>, grid, block, 5, out, iters, 7u);
        for (int o = 0; o < 9; o++) {
            double ops = (double)cus * wps * 256 * iters * 8;
            printf("waves/SIMD %d  %-16s %8.3f ms  %8.2f Tlane-op/s  (%.2f cycles/wave-instr/SIMD @2.4GHz)\n", wps, names[o], ms[o],
                   ops / ms[o] / 1e9, ms[o] * 1e-3 * 2.4e9 / ((double)iters * 8 * wps));
        }
    }
    for (int wi = 0; wi < 6; wi++) {
        const int wps = wlist[wi];
        dim3 grid(cus * wps), block(256);
        int it = 2000;
        double t = time_kernel(k_fmul30, grid, block, 3, (int32_t *)out, it, 12345);
        double t0 = time_kernel(k_fmul30x<0>, grid, block, 3, (int32_t *)out + 64, it, 12345);
        double t1 = time_kernel(k_fmul30x<1>, grid, block, 3, (int32_t *)out + 128, it, 12345);
        printf("waves/SIMD %d  Fq30 mul (inline chain) %7.2f G/s (%6.0f cyc/wave-mul/SIMD)   compiler-scheduled %7.2f G/s   single-chain asm %7.2f G/s\n", wps, (double)cus * wps * 256 * it / t / 1e6,
               t * 1e-3 * 2.4e9 / ((double)it * wps), (double)cus * wps * 256 * it / t0 / 1e6, (double)cus * wps * 256 * it / t1 / 1e6);
    }
    // field multiply / mixed add throughput
    size_t nthreads = (size_t)cus * 8 * 256;
    Fq *fq_in, *fq_out; Fr *fr_in, *fr_out;
    CHECK(hipMalloc(&fq_in, (nthreads + 8) * sizeof(Fq))); CHECK(hipMalloc(&fq_out, nthreads * sizeof(Fq)));
    CHECK(hipMalloc(&fr_in, (nthreads + 8) * sizeof(Fr))); CHECK(hipMalloc(&fr_out, nthreads * sizeof(Fr)));
    CHECK(hipMemset(fq_in, 0x5a, (nthreads + 8) * sizeof(Fq))); CHECK(hipMemset(fr_in, 0x3c, (nthreads + 8) * sizeof(Fr)));
    G1Affine *pts; G1Xyzz *pout; CHECK(hipMalloc(&pts, 4096 * sizeof(G1Affine))); CHECK(hipMalloc(&pout, nthreads * sizeof(G1Xyzz)));
    printf("init points...\n");
    hipLaunchKernelGGL(k_init_points, dim3(1), dim3(64), 0, 0, pts, 4096);
    CHECK(hipDeviceSynchronize());
    printf("init points done\n");
    for (int wps = 1; wps <= 8; wps *= 2) {
        dim3 grid(cus * wps), block(256);
        int it = 2000;
        double t1 = time_kernel(k_fmul<Fq>, grid, block, 3, fq_out, fq_in, it);
        double t2 = time_kernel(k_fmul<Fr>, grid, block, 3, fr_out, fr_in, it);
        double nm = (double)cus * wps * 256 * it;
        printf("waves/SIMD %d  Fq mul %7.2f G/s (%6.0f cyc/wave-mul/SIMD)   Fr mul %7.2f G/s (%6.0f cyc)\n", wps, nm / t1 / 1e6,
               t1 * 1e-3 * 2.4e9 / ((double)it * wps), nm / t2 / 1e6, t2 * 1e-3 * 2.4e9 / ((double)it * wps));
        if (wps <= 2) {
            int ia = 200;
            double t3 = time_kernel(k_madd, grid, block, 3, pout, pts, ia);
            printf("waves/SIMD %d  G1 mixed add %7.3f G/s  (%.1f us per add per wave)\n", wps, (double)cus * wps * 256 * ia / t3 / 1e6,
                   t3 * 1e3 / ia);
        }
    }
    return 0;
}
