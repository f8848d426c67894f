// Radial Hankel step (rows a1-a3 of SURVEY section 8): the reference's only live GPU kernel,
// OpenCL `apply_weights` (xframe/projects/fxs/projectLibrary/hankel_transforms.py:702-731 midpoint,
// 671-700 trapz):   out[k, lm] = c_l * sum_p W[l(lm)][p][k] * in[p (+1), lm]
// W is the *real* raw weight array (hankel_transforms.py:399-410); the complex prefactor
// c_l = (-/+ i)^l * scale of assemble_weights_mid (426-452) is applied once in the epilogue, so the
// contraction is real-matrix x complex-panel.
#include "mtip_internal.h"
#include <vector>

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

void launch_hankel_mfma(mtip_ctx* c, const double2* in, double2* out, int inverse);

void launch_hankel(mtip_ctx* c, const double2* in, double2* out, int inverse) {
    ProfScope ps(c, "hankel");
    if (c->d_htiles != nullptr && !c->hankel_simple) {
        launch_hankel_mfma(c, in, out, inverse);
        return;
    }
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

// ------------------------------------------------------------------------------------------------------
// MFMA version: per order l the contraction is a real GEMM  D[k][col] = sum_p W_l[p][k] * X[p][col] with the
// (batch, m, re/im) axes flattened into columns (B (4l+2) of them, tiled by 16 without waste when B is a
// multiple of 8).  v_mfma_f64_16x16x4_f64: lane j holds A[i = j&15][kk = j>>4] = W_l[p0+kk][k0+i] (16 lanes read
// 128 contiguous bytes), B[kk = j>>4][col = j&15] = X[p0+kk][col0+col] (8 consecutive complex numbers), and
// D reg r = D[(j>>4) + 4r][j&15].  W_l streams through L1/L2 (131 KB per order at Nq = 128, shared by all
// column tiles).  The (-/+ i)^l * scale prefactor is applied in the epilogue: multiplying by
// +-i swaps the re/im columns, i.e. neighbouring lanes (shfl_xor 1).
#define HK_MT 2           // 16-row output tiles per wave
#define HK_PF 4           // k-steps (of 4 shells) whose fragments are in flight together

struct HankelTile { int l, cflat0; };

// One wave = one 16-column tile x HK_MT row tiles.  With HK_MT = 2 a 128-shell transform is spread over 4 waves per
// column tile (17 waves per CU at 8 restarts, L = 32): the kernel is bound by the latency of the W_l / panel
// fragment loads (L2 hits), so it wants many waves and HK_PF k-steps of loads in flight (double buffered in
// registers) rather than long per-wave MFMA chains.
__global__ void __launch_bounds__(256) k_hankel_mfma(const double* __restrict__ in, double* __restrict__ out,
                                                     const double* __restrict__ W, const HankelTile* __restrict__ tiles,
                                                     int n_tiles, int N, int Np, int L, int B, int poffs, double scale,
                                                     int sign) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= n_tiles) return;                                  // uniform per wave
    const int mg = blockIdx.y;                                    // row group of HK_MT x 16 output shells
    const HankelTile tinfo = tiles[tile];
    const int l = tinfo.l;
    const int ncl = 4 * l + 2;                                    // doubles per (batch, shell) of this order
    const int nlm2 = 2 * (L + 1) * (L + 1);                       // doubles per (batch, shell)
    const int li = lane & 15, kk = lane >> 4;
    // column of this lane
    const int cflat = tinfo.cflat0 + li;
    const bool col_ok = cflat < B * ncl;
    const int b = col_ok ? cflat / ncl : 0;
    const int within = col_ok ? cflat - b * ncl : 0;
    const size_t col_off = (size_t)b * N * nlm2 + 2 * (size_t)l * l + within;     // + shell * nlm2
    const double* Wl = W + (size_t)l * Np * N;
    const int k_base = mg * (HK_MT * 16);
    v4f64 acc[HK_MT];
#pragma unroll
    for (int t = 0; t < HK_MT; ++t) acc[t] = v4f64{0.0, 0.0, 0.0, 0.0};
    const int n_steps = (Np + 3) / 4;
    double a_cur[HK_PF][HK_MT], a_nxt[HK_PF][HK_MT], b_cur[HK_PF], b_nxt[HK_PF];
    bool k_ok[HK_MT];
#pragma unroll
    for (int t = 0; t < HK_MT; ++t) k_ok[t] = k_base + t * 16 + li < N;
    auto load_group = [&](int s0, double (&af)[HK_PF][HK_MT], double (&bf)[HK_PF]) {
#pragma unroll
        for (int u = 0; u < HK_PF; ++u) {
            const int p = (s0 + u) * 4 + kk;
            const bool p_ok = p < Np;
            bf[u] = (p_ok && col_ok) ? in[col_off + (size_t)(p + poffs) * nlm2] : 0.0;
#pragma unroll
            for (int t = 0; t < HK_MT; ++t) af[u][t] = (p_ok && k_ok[t]) ? Wl[(size_t)p * N + k_base + t * 16 + li] : 0.0;
        }
    };
    load_group(0, a_cur, b_cur);
    for (int s0 = 0; s0 < n_steps; s0 += HK_PF) {
        load_group(s0 + HK_PF, a_nxt, b_nxt);                      // beyond the last step: p >= Np, loads nothing
#pragma unroll
        for (int u = 0; u < HK_PF; ++u)
#pragma unroll
            for (int t = 0; t < HK_MT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[u][t], b_cur[u], acc[t], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < HK_PF; ++u) {
            b_cur[u] = b_nxt[u];
#pragma unroll
            for (int t = 0; t < HK_MT; ++t) a_cur[u][t] = a_nxt[u][t];
        }
    }
    // epilogue: * scale * (-/+ i)^l, store
    int r = l & 3;
    if (sign < 0) r = (4 - r) & 3;
    const bool is_im = (li & 1) != 0;
#pragma unroll
    for (int t = 0; t < HK_MT; ++t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double v = acc[t][j] * scale;
            const double partner = __shfl_xor(v, 1, 64);
            double o;
            if (r == 0) o = v;
            else if (r == 2) o = -v;
            else if (r == 1) o = is_im ? partner : -partner;       // * i : (a+bi) i = -b + a i
            else o = is_im ? -partner : partner;                   // * -i: (a+bi)(-i) = b - a i
            const int k = k_base + t * 16 + kk + 4 * j;
            if (col_ok && k < N) out[col_off + (size_t)k * nlm2] = o;
        }
    }
}

void launch_hankel_mfma(mtip_ctx* c, const double2* in, double2* out, int inverse) {
    const int n_tiles = c->n_htiles;
    const dim3 grid((unsigned)div_up(n_tiles, 4), (unsigned)div_up(c->N, HK_MT * 16));
    hipLaunchKernelGGL(k_hankel_mfma, grid, dim3(256), 0, c->stream, reinterpret_cast<const double*>(in),
                       reinterpret_cast<double*>(out), (const double*)c->d_W, (const HankelTile*)c->d_htiles, n_tiles,
                       c->N, c->Np, c->L, c->B, c->cfg.hankel_trapz ? 1 : 0, inverse ? c->inv_scale : c->fwd_scale,
                       inverse ? +1 : -1);
}

int build_hankel_tiles(mtip_ctx* c) {
    std::vector<HankelTile> t;
    for (int l = c->L; l >= 0; --l) {                       // heavy orders first
        const int ncols = c->B * (4 * l + 2);
        for (int c0 = 0; c0 < ncols; c0 += 16) t.push_back(HankelTile{l, c0});
    }
    c->n_htiles = (int)t.size();
    if (hipMalloc((void**)&c->d_htiles, t.size() * sizeof(HankelTile)) != hipSuccess) return MTIP_ENOMEM;
    (void)hipMemcpy(c->d_htiles, t.data(), t.size() * sizeof(HankelTile), hipMemcpyHostToDevice);
    return MTIP_OK;
}
