// Micro-benchmark: the gap between DEPENDENT kernels of one stream -- plain launches against a captured hipGraph -- for chains of
// short kernels like a phasing step's transform block (six kernels of 15-50 us on 256-384 workgroups).
//   hipcc --offload-arch=gfx950 -O3 graph_gaps.hip -o graph_gaps && ./graph_gaps
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(512) k_spin(long long ticks, double* out) {
    const long long t0 = wall_clock64();
    double v = threadIdx.x;
    while (wall_clock64() - t0 < ticks) v = v * 1.0000001 + 1e-9;
    if (v == 12345.678) out[0] = v;
}

int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    double* d;
    CK(hipMalloc(&d, 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int chain = 6, reps = 200;
    for (int wgs : {256, 384}) {
        for (long long us : {5LL, 20LL, 40LL}) {
            const long long ticks = us * 100;                       // wall clock: 100 MHz
            // plain launches
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(512), 0, s, ticks, d);
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r)
                for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(512), 0, s, ticks, d);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms_plain = 0;
            CK(hipEventElapsedTime(&ms_plain, e0, e1));
            // captured graph of one chain
            hipGraph_t g;
            hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(512), 0, s, ticks, d);
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms_graph = 0;
            CK(hipEventElapsedTime(&ms_graph, e0, e1));
            const double per_plain = 1e3 * ms_plain / (reps * chain), per_graph = 1e3 * ms_graph / (reps * chain);
            printf("%d workgroups, %lld us kernels: plain %.2f us per kernel (gap %.2f), graph %.2f us per kernel (gap %.2f)\n", wgs, us, per_plain,
                   per_plain - us, per_graph, per_graph - us);
            (void)hipGraphExecDestroy(ge);
            (void)hipGraphDestroy(g);
        }
    }
    // hand-over between two streams: kernel on A, event, B waits, kernel on B, event, A waits, ... (what a ring of engines pays per turn)
    {
        hipStream_t s2;
        CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        hipEvent_t ea, eb;
        CK(hipEventCreateWithFlags(&ea, hipEventDisableTiming | hipEventDisableSystemFence));
        CK(hipEventCreateWithFlags(&eb, hipEventDisableTiming | hipEventDisableSystemFence));
        const long long us = 20, ticks = us * 100;
        const int n = 300;
        for (int warm = 0; warm < 2; ++warm) {
            CK(hipStreamSynchronize(s));
            CK(hipStreamSynchronize(s2));
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < n; ++r) {
                hipLaunchKernelGGL(k_spin, dim3(256), dim3(512), 0, s, ticks, d);
                CK(hipEventRecord(ea, s));
                CK(hipStreamWaitEvent(s2, ea, 0));
                hipLaunchKernelGGL(k_spin, dim3(256), dim3(512), 0, s2, ticks, d);
                CK(hipEventRecord(eb, s2));
                CK(hipStreamWaitEvent(s, eb, 0));
            }
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (warm) printf("ping-pong between two streams, %lld us kernels: %.2f us per kernel (hand-over %.2f us)\n", us, 1e3 * ms / (2 * n), 1e3 * ms / (2 * n) - us);
        }
    }
    return 0;
}
