// Cost of a launch boundary between dependent kernels on one stream (no profiler): hipcc --offload-arch=gfx950 -O2 -o tools/bin/launch_gap tools/launch_gap.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_empty(int *p) { if (p && threadIdx.x == 1000) *p = 1; }
__global__ void k_spin(long long cycles) { long long t0 = wall_clock64(); while (wall_clock64() - t0 < cycles) {} }
int main() {
    hipStream_t st; hipStreamCreate(&st);
    for (int rep = 0; rep < 3; rep++) {
        for (int n : {1, 10, 100, 1000}) {
            hipStreamSynchronize(st);
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_empty, 1, 64, 0, st, nullptr);
            hipStreamSynchronize(st);
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            printf("empty x%d: %.1f us total, %.2f us each\n", n, us, us / n);
        }
        // kernels that each spin 50 us (100 MHz wall clock -> 5000 ticks): GPU-bound chain, boundary = (total - n*50)/n
        for (int n : {1, 20}) {
            hipStreamSynchronize(st);
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_spin, 256, 256, 0, st, 5000LL);
            hipStreamSynchronize(st);
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            printf("spin50us x%d: %.1f us total, %.2f us each\n", n, us, us / n);
        }
    }
    return 0;
}
