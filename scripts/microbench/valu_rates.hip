// Micro-benchmark (diagnostic, not part of the library): cycles per wave-instruction of the FP64 vector ops the
// polar-factor kernel is built from, with 1 and 2 waves per SIMD.  hipcc --offload-arch=gfx950 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

#define REP 64
template <int MODE>
__global__ void __launch_bounds__(512) k(double* out, long long* cyc, int iters, int lp) {
    double a[16], b0 = out[threadIdx.x & 7] + 1.0, b1 = out[(threadIdx.x + 1) & 7] + 0.5;
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = out[(threadIdx.x + i) & 15];
    __syncthreads();
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {                       // 16 independent fma f64 chains, VGPR operands
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = fma(a[i], b0, b1);
        } else if (MODE == 1) {                // fma with a wave-uniform (SGPR) operand from v_readlane
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int lo = __builtin_amdgcn_readlane(__double2loint(a[(i + 1) & 15]), lp);
                    const int hi = __builtin_amdgcn_readlane(__double2hiint(a[(i + 1) & 15]), lp);
                    a[i] = fma(a[i], __hiloint2double(hi, lo), b1);
                }
        } else if (MODE == 2) {                // readlane only (result summed on the scalar side)
            int acc = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc += __builtin_amdgcn_readlane(__double2loint(a[i]), (lp + i) & 63);
            a[0] += acc;
        } else if (MODE == 3) {                // ds_bpermute
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] += __shfl(a[(i + 1) & 15], lp);
        } else if (MODE == 5 || MODE == 6 || MODE == 7) {   // f64 MFMA alone (5), MFMA + independent v_fma_f64 interleaved (6),
            typedef double d4 __attribute__((ext_vector_type(4)));     // the same v_fma_f64 count alone (7)
            d4 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = d4{a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]};
            double f[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = a[i] + 1.0;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (MODE != 7) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, b1, acc[i], 0, 0, 0);
                    if (MODE != 5) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) f[q & 7] = fma(f[q & 7], b0, b1);       // 16 v_fma_f64 per MFMA (64 cycles each)
                    }
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[4 * i] = acc[i][0]; a[4 * i + 1] = acc[i][1]; a[4 * i + 2] = acc[i][2]; a[4 * i + 3] = acc[i][3]; }
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] += f[i];
        } else if (MODE == 4) {                // f32 fma for reference
            float f[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) f[i] = (float)a[i];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) f[i] = fmaf(f[i], (float)b0, (float)b1);
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = f[i];
        }
    }
    long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int threads, int n_instr_per_iter) {
    const int blocks = 256, iters = 2000;
    double* out; long long* cyc;
    hipMalloc(&out, blocks * threads * sizeof(double)); hipMemset(out, 0, blocks * threads * sizeof(double));
    hipMalloc(&cyc, blocks * (threads / 64) * sizeof(long long));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 5);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * (threads / 64));
    hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += v; mean /= h.size();
    printf("%-28s %4d threads/CU (%d waves/SIMD): %.2f cycles per wave-instruction (per wave), %.2f per SIMD slot\n", name, threads,
           threads / 256, mean / iters / n_instr_per_iter, mean / iters / n_instr_per_iter / (threads / 256.0));
    hipFree(out); hipFree(cyc);
}

// precision of the hardware reciprocal-square-root estimate and of one / two Newton steps on it (what fast_rsqrt in k_proj.hip builds on)
__global__ void k_rsq(const double* x, double* e0, double* e1, double* e2, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double y = __builtin_amdgcn_rsq(v);
    const double ref = 1.0 / sqrt(v);
    e0[i] = fabs(y / ref - 1.0);
    y = y * (1.5 - 0.5 * v * y * y);
    e1[i] = fabs(y / ref - 1.0);
    y = y * (1.5 - 0.5 * v * y * y);
    e2[i] = fabs(y / ref - 1.0);
}

static void rsq_precision() {
    const int n = 1 << 20;
    std::vector<double> h(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (double)(s >> 11) / 9007199254740992.0;
        h[i] = ldexp(1.0 + u, (int)(s % 61) - 30);
    }
    double *x, *e0, *e1, *e2;
    hipMalloc(&x, n * 8); hipMalloc(&e0, n * 8); hipMalloc(&e1, n * 8); hipMalloc(&e2, n * 8);
    hipMemcpy(x, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_rsq, dim3(n / 256), dim3(256), 0, 0, x, e0, e1, e2, n);
    std::vector<double> r0(n), r1(n), r2(n);
    hipMemcpy(r0.data(), e0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), e1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), e2, n * 8, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0, m2 = 0;
    for (int i = 0; i < n; ++i) { m0 = fmax(m0, r0[i]); m1 = fmax(m1, r1[i]); m2 = fmax(m2, r2[i]); }
    printf("v_rsq_f64 max relative error: estimate %.3e, after one Newton step %.3e, after two %.3e\n", m0, m1, m2);
}

int main() {
    rsq_precision();
    for (int t : {256, 512}) {
        if (t == 256) { run<0>("v_fma_f64 (VGPR)", 256, 64); run<1>("2 v_readlane + v_fma_f64", 256, 64); run<2>("v_readlane_b32", 256, 64);
                        run<3>("ds_bpermute x2 + v_add_f64", 256, 64); run<4>("v_fma_f32", 256, 64); }
        else { run<0>("v_fma_f64 (VGPR)", 512, 64); run<1>("2 v_readlane + v_fma_f64", 512, 64); run<2>("v_readlane_b32", 512, 64);
               run<3>("ds_bpermute x2 + v_add_f64", 512, 64); run<4>("v_fma_f32", 512, 64); }
    }
    // matrix-core / vector co-issue: 16 MFMA (5), 16 MFMA + 256 v_fma_f64 (6), 256 v_fma_f64 (7) per iteration; cycles per iteration
    for (int t : {256, 512}) {
        if (t == 256) { run<5>("16 v_mfma_f64_16x16x4 [per iter]", 256, 1); run<6>("16 mfma + 256 v_fma_f64 [per iter]", 256, 1); run<7>("256 v_fma_f64 [per iter]", 256, 1); }
        else { run<5>("16 v_mfma_f64_16x16x4 [per iter]", 512, 1); run<6>("16 mfma + 256 v_fma_f64 [per iter]", 512, 1); run<7>("256 v_fma_f64 [per iter]", 512, 1); }
    }
    return 0;
}
