// Micro-benchmark (diagnostic): how many HIP streams of one process execute kernels at the same time?  K streams each get a chain
// of `reps` one-workgroup kernels that spin for `us` microseconds; the wall time of all chains tells how many ran side by side.
// hipcc --offload-arch=gfx950 queue_concurrency.hip -o queue_concurrency;  GPU_MAX_HW_QUEUES=8 ./queue_concurrency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin(long long cycles, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (sink && threadIdx.x == 1024) *sink = 1;
}

int main() {
    int rate_khz = 0;
    (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
    const double us = 200.0;
    const long long cycles = (long long)(us * 1e-6 * rate_khz * 1e3);
    const int reps = 50;
    printf("wall clock %d kHz, %d kernels of %.0f us per stream\n", rate_khz, reps, us);
    for (int K : {1, 2, 3, 4, 5, 6, 8, 12, 16}) {
        std::vector<hipStream_t> st(K);
        for (auto& s : st) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        for (auto& s : st) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 1000, (int*)nullptr);   // queue creation, warm
        (void)hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r)
            for (auto& s : st) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, cycles, (int*)nullptr);
        (void)hipDeviceSynchronize();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("K=%2d streams: %.2f ms  (serial would be %.2f ms, fully concurrent %.2f ms) -> %.1f chains side by side\n", K, ms,
               K * reps * us * 1e-3, reps * us * 1e-3, K * reps * us * 1e-3 / ms);
        for (auto& s : st) (void)hipStreamDestroy(s);
    }
    return 0;
}
