// issue_cost.hip -- what ONE more instruction of each kind costs next to a v_mad_i64_i32 stream at 2 waves per SIMD
// (the occupancy of k_accum_affine).  Each kernel runs a loop body of 8 "columns"; a column is 14 dependent mads plus the
// extras under test, all in one asm statement on fixed registers (nothing for the compiler to add or schedule).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/issue_cost tools/issue_cost.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

#define MAD "v_mad_i64_i32 v[10:11], vcc, v2, v3, v[10:11]\n"
#define MAD14 MAD MAD MAD MAD MAD MAD MAD MAD MAD MAD MAD MAD MAD MAD
#define X8(s) s s s s s s s s
#define CLOB "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "vcc"

template <int V>
__global__ __launch_bounds__(256) void k_cost(uint32_t *out, int iters, uint32_t seed) {
    int32_t a = (int32_t)(seed + threadIdx.x), b = (int32_t)(seed * 3 + blockIdx.x);
    asm volatile("v_mov_b32 v2, %0\nv_mov_b32 v3, %1\nv_mov_b32 v10, %0\nv_mov_b32 v11, 0\nv_mov_b32 v12, %1\nv_mov_b32 v13, 0\nv_mov_b32 v14, 1\nv_mov_b32 v15, 2\nv_mov_b32 v16, 3\nv_mov_b32 v17, 4" ::"v"(a), "v"(b) : "v2", "v3", CLOB);
    for (int i = 0; i < iters; i++) {
        if (V == 0) asm volatile(X8(MAD14) ::: CLOB);
        if (V == 1) asm volatile(X8(MAD14 "s_nop 0\n") ::: CLOB);
        if (V == 2) asm volatile(X8(MAD14 "v_mul_lo_u32 v12, v10, v3\n") ::: CLOB);
        if (V == 3) asm volatile(X8(MAD14 "v_ashrrev_i32 v12, 2, v10\n") ::: CLOB);
        if (V == 4) asm volatile(X8(MAD14 "v_ashrrev_i64 v[10:11], 30, v[10:11]\n") ::: CLOB);
        if (V == 5) asm volatile(X8(MAD14 "v_bfe_i32 v12, v10, 0, 30\n") ::: CLOB);
        if (V == 6) asm volatile(X8(MAD14 "v_lshl_add_u64 v[10:11], v[10:11], 0, v[12:13]\n") ::: CLOB);
        if (V == 7) asm volatile(X8(MAD14 "v_add_u32 v14, v14, v15\nv_sub_u32 v16, v16, v17\nv_add_u32 v15, v15, v14\nv_sub_u32 v17, v17, v16\n") ::: CLOB);
        if (V == 8) asm volatile(X8(MAD14 "v_mul_lo_u32 v12, v10, v3\nv_ashrrev_i32 v12, 2, v12\nv_mad_i64_i32 v[10:11], vcc, v12, v3, v[10:11]\nv_ashrrev_i64 v[10:11], 30, v[10:11]\n") ::: CLOB);
        if (V == 9) asm volatile(X8(MAD14 "s_nop 0\nv_mul_lo_u32 v12, v10, v3\nv_ashrrev_i32 v12, 2, v12\nv_mad_i64_i32 v[10:11], vcc, v12, v3, v[10:11]\nv_ashrrev_i64 v[10:11], 30, v[10:11]\n") ::: CLOB);
        if (V == 10) asm volatile(X8(MAD14 "v_and_b32 v12, 0x3fffffff, v10\n") ::: CLOB);
        if (V == 11) asm volatile(X8(MAD14 "v_mad_i64_i32 v[10:11], vcc, v12, 1, v[10:11]\n") ::: CLOB);
        if (V == 12) asm volatile(X8(MAD14 "v_alignbit_b32 v12, v11, v10, 30\nv_ashrrev_i32 v13, 30, v11\n") ::: CLOB);
    }
    uint32_t r;
    asm volatile("v_xor_b32 %0, v10, v11\nv_xor_b32 %0, %0, v12\nv_xor_b32 %0, %0, v14\nv_xor_b32 %0, %0, v16" : "=v"(r)::CLOB);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
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
    const int iters = 3000;
    const char *names[] = {"14 mads (baseline)", "+ s_nop 0", "+ v_mul_lo_u32", "+ v_ashrrev_i32", "+ v_ashrrev_i64", "+ v_bfe_i32", "+ v_lshl_add_u64",
                           "+ 4 x v_add/sub_u32", "+ quotient epilogue (mul_lo, ashr, mad, ashr64)", "+ s_nop 0 + quotient epilogue", "+ v_and_b32 (literal)",
                           "+ v_mad_i64_i32 by inline constant 1", "+ v_alignbit_b32 + v_ashrrev_i32 (64-bit shift in two halves)"};
    for (int w : {2, 1, 4}) {
        dim3 grid(cus * w), block(256);
        double ms[13];
        ms[0] = time_kernel(k_cost<0>, grid, block, 3, out, iters);
        ms[1] = time_kernel(k_cost<1>, grid, block, 3, out, iters);
        ms[2] = time_kernel(k_cost<2>, grid, block, 3, out, iters);
        ms[3] = time_kernel(k_cost<3>, grid, block, 3, out, iters);
        ms[4] = time_kernel(k_cost<4>, grid, block, 3, out, iters);
        ms[5] = time_kernel(k_cost<5>, grid, block, 3, out, iters);
        ms[6] = time_kernel(k_cost<6>, grid, block, 3, out, iters);
        ms[7] = time_kernel(k_cost<7>, grid, block, 3, out, iters);
        ms[8] = time_kernel(k_cost<8>, grid, block, 3, out, iters);
        ms[9] = time_kernel(k_cost<9>, grid, block, 3, out, iters);
        ms[10] = time_kernel(k_cost<10>, grid, block, 3, out, iters);
        ms[11] = time_kernel(k_cost<11>, grid, block, 3, out, iters);
        ms[12] = time_kernel(k_cost<12>, grid, block, 3, out, iters);
        const double cyc = 2.4e9 * 1e-3 / ((double)iters * 8 * w);  // nominal cycles per column per SIMD, per ms
        printf("waves/SIMD %d: cycles per column of 14 mads per SIMD at a nominal 2.4 GHz; extra = cost of the added instructions\n", w);
        for (int v = 0; v < 13; v++)
            printf("  %-62s %7.3f ms  %7.2f cyc/column  extra %6.2f cyc  (%.2f mad-equivalents)\n", names[v], ms[v], ms[v] * cyc, (ms[v] - ms[0]) * cyc,
                   (ms[v] - ms[0]) / (ms[0] / 14.0));
    }
    return 0;
}
