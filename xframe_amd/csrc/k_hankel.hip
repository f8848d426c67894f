// Radial Hankel step (rows a1-a3 of SURVEY section 8): the reference's only live GPU kernel,
// OpenCL `apply_weights` (xframe/projects/fxs/projectLibrary/hankel_transforms.py:702-731 midpoint,
// 671-700 trapz):   out[k, lm] = c_l * sum_p W[l(lm)][p][k] * in[p (+1), lm]
// W is the *real* raw weight array (hankel_transforms.py:399-410); the complex prefactor
// c_l = (-/+ i)^l * scale of assemble_weights_mid (426-452) is applied once in the epilogue, so the
// contraction is real-matrix x complex-panel.
#include "mtip_internal.h"

// one thread per output element; consecutive threads = consecutive lm (coalesced panel reads,
// W broadcast within an order l)
__global__ void __launch_bounds__(256) k_hankel_simple(const double2* __restrict__ in, double2* __restrict__ out,
                                                       const double* __restrict__ W, int N, int Np, int L, int poffs,
                                                       double scale, int sign, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int nlm = (L + 1) * (L + 1);
    const int lm = (int)(idx % nlm);
    const long long bk = idx / nlm;
    const int k = (int)(bk % N);
    const long long b = bk / N;
    const int l = isqrt_lm(lm);
    const double* w = W + ((size_t)l * Np) * N + k;
    const double2* src = in + ((size_t)b * N + poffs) * nlm + lm;
    double ar = 0.0, ai = 0.0;
    for (int p = 0; p < Np; ++p) {
        const double wv = w[(size_t)p * N];
        const double2 v = src[(size_t)p * nlm];
        ar = fma(wv, v.x, ar);
        ai = fma(wv, v.y, ai);
    }
    out[idx] = cmul_ipow(make_double2(ar * scale, ai * scale), l, sign);
}

void launch_hankel(mtip_ctx* c, const double2* in, double2* out, int inverse) {
    ProfScope ps(c, "hankel");
    const long long total = (long long)c->B * c->N * c->nlm;
    hipLaunchKernelGGL(k_hankel_simple, dim3((unsigned)div_up(total, 256)), dim3(256), 0, c->stream, in, out,
                       (const double*)c->d_W, c->N, c->Np, c->L, c->cfg.hankel_trapz ? 1 : 0,
                       inverse ? c->inv_scale : c->fwd_scale, inverse ? +1 : -1, total);
}

// out = a - b for shells > 0, out = a for shell 0   (ft_stab add-back folded into coefficient space,
// misk.py:326-329 add_above_zero_index; used by the fused step only)
__global__ void __launch_bounds__(256) k_coeff_diff(const double2* __restrict__ a, const double2* __restrict__ b,
                                                    double2* __restrict__ out, int N, int nlm, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int q = (int)((idx / nlm) % N);
    double2 v = a[idx];
    if (q > 0) v = csub(v, b[idx]);
    out[idx] = v;
}

void launch_coeff_diff(mtip_ctx* c, const double2* a, const double2* b, double2* out) {
    const long long total = (long long)c->B * c->N * c->nlm;
    hipLaunchKernelGGL(k_coeff_diff, dim3((unsigned)div_up(total, 256)), dim3(256), 0, c->stream, a, b, out, c->N,
                       c->nlm, total);
}
