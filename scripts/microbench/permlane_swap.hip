// Semantics and cost of v_permlane16_swap_b32 / v_permlane32_swap_b32 (gfx950): what each lane receives when both operands are
// the lane id, and cycles per instruction in a dependent chain.   hipcc --offload-arch=gfx950 permlane_swap.hip -o permlane_swap
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_sem(unsigned* out) {
    const unsigned v = threadIdx.x, w = threadIdx.x + 100u;
    auto r = __builtin_amdgcn_permlane16_swap(v, w, false, false);   // (vdst_old, src0_old) -> {vdst_new, src0_new}
    auto q = __builtin_amdgcn_permlane32_swap(v, w, false, false);
    out[threadIdx.x * 4 + 0] = r[0];
    out[threadIdx.x * 4 + 1] = r[1];
    out[threadIdx.x * 4 + 2] = q[0];
    out[threadIdx.x * 4 + 3] = q[1];
}
__global__ void k_time(unsigned* out, long long* cyc, int n) {
    unsigned a = threadIdx.x, b = threadIdx.x * 7u;
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
        a = r[0] + 1u; b = r[1];
        auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
        a = q[0]; b = q[1] + 1u;
    }
    const long long t1 = clock64();
    out[threadIdx.x] = a ^ b;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    unsigned* d; long long* c;
    hipMalloc(&d, 64 * 4 * 4); hipMalloc(&c, 8);
    hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, d);
    unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"permlane16_swap vdst_new", "permlane16_swap src0_new", "permlane32_swap vdst_new", "permlane32_swap src0_new"};
    for (int j = 0; j < 4; ++j) {
        printf("%s (vdst_old = lane, src0_old = lane + 100), first lane of each row of 16:", names[j]);
        for (int row = 0; row < 4; ++row) printf("  row %d: %u", row, h[(row * 16) * 4 + j]);
        printf("\n");
    }
    hipLaunchKernelGGL(k_time, dim3(1), dim3(64), 0, 0, d, c, 1000);
    long long hc; hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
    printf("dependent chain: %.1f cycles per (permlane16_swap + add + permlane32_swap + add) iteration, one wave\n", hc / 1000.0);
    return 0;
}
