// Spherical-harmonic transform kernels (row a4 of SURVEY section 8):
//   forward  f(theta,phi) -> f_lm :  FFT along phi (LDS Stockham), then Gauss-Legendre sum over theta
//   inverse  f_lm -> f(theta,phi) :  Legendre synthesis, then inverse FFT along phi (+ fused epilogue)
// Convention: orthonormal Y_lm with Condon-Shortley phase, index l(l+1)+m -- what the reference gets
// from shtns (xframe/externalLibraries/shtns_plugin.py:20-24, 250-261).
#include "mtip_internal.h"
#include <cmath>

// ------------------------------------------------------------------------------------------------
// Legendre table  P[(poff[m] + l - m) * nt + t] = Y_lm(theta_t, 0),  m >= 0   (host, one-off)
// ------------------------------------------------------------------------------------------------
void build_legendre_tables(mtip_ctx* c, const double* cos_theta) {
    const int L = c->L, nt = c->nt;
    std::vector<int> poff(L + 2);
    for (int m = 0; m <= L + 1; ++m) poff[m] = m * (L + 1) - m * (m - 1) / 2;
    const size_t rows = (size_t)(L + 1) * (L + 2) / 2;
    std::vector<double> P(rows * nt);
    std::vector<double2> AB(rows, make_double2(0.0, 0.0));   // P_lm = a_lm (x P_l-1,m - b_lm P_l-2,m), l >= m + 2
    const double pi = 3.14159265358979323846;
    for (int t = 0; t < nt; ++t) {
        const double x = cos_theta[t];
        const double s = std::sqrt(std::fmax(0.0, 1.0 - x * x));
        double pmm = std::sqrt(1.0 / (4.0 * pi));
        for (int m = 0; m <= L; ++m) {
            if (m > 0) pmm = -std::sqrt((2.0 * m + 1.0) / (2.0 * m)) * s * pmm;
            P[(size_t)(poff[m]) * nt + t] = pmm;
            if (m < L) {
                double p2 = pmm;
                double p1 = std::sqrt(2.0 * m + 3.0) * x * pmm;
                P[(size_t)(poff[m] + 1) * nt + t] = p1;
                for (int l = m + 2; l <= L; ++l) {
                    const double a = std::sqrt((4.0 * l * l - 1.0) / ((double)l * l - (double)m * m));
                    const double b = std::sqrt((((double)l - 1.0) * (l - 1.0) - (double)m * m) / (4.0 * (l - 1.0) * (l - 1.0) - 1.0));
                    const double p = a * (x * p1 - b * p2);
                    P[(size_t)(poff[m] + l - m) * nt + t] = p;
                    AB[(size_t)poff[m] + l - m] = make_double2(a, b);
                    p2 = p1;
                    p1 = p;
                }
            }
        }
    }
    (void)mtip_copy(c, c->d_P, P.data(), P.size() * sizeof(double), hipMemcpyHostToDevice);
    (void)mtip_copy(c, c->d_poff, poff.data(), poff.size() * sizeof(int), hipMemcpyHostToDevice);
    if (c->d_AB != nullptr) (void)mtip_copy(c, c->d_AB, AB.data(), AB.size() * sizeof(double2), hipMemcpyHostToDevice);
    // theta-major copy of the northern half + (l,m) lookup for the fused kernels (k_sht_fused.hip)
    if (c->d_PT != nullptr) {
        const int nth = nt / 2;
        std::vector<double> PT((size_t)nth * rows);
        std::vector<int> lmtab(rows);
        for (int m = 0; m <= L; ++m)
            for (int l = m; l <= L; ++l) {
                const size_t idx = (size_t)poff[m] + l - m;
                lmtab[idx] = l | (m << 8);
                for (int t = 0; t < nth; ++t) PT[(size_t)t * rows + idx] = P[idx * nt + t];
            }
        (void)mtip_copy(c, c->d_PT, PT.data(), PT.size() * sizeof(double), hipMemcpyHostToDevice);
        (void)mtip_copy(c, c->d_lmtab, lmtab.data(), lmtab.size() * sizeof(int), hipMemcpyHostToDevice);
        // the same table in the CHUNK layout of the chained kernel's Legendre sums (k_sht_chain.hip, CHK): a thread owns up to three
        // orders l of ONE (m, parity of l - m), so the panel values G[theta][+-m] it multiplies them with are read from LDS once for
        // the three -- slot u * 256 + t = order u of chunk t (l | m << 8, or -1: no such order, table entry 0)
        c->chain_chunks = 0;
        if (c->d_PTc != nullptr) {
            std::vector<int> lmc(3 * 256, -1);
            std::vector<double> PTc((size_t)nth * 3 * 256, 0.0);
            int t = 0;
            bool fits = true;
            for (int m = 0; m <= L && fits; ++m)
                for (int par = 0; par < 2 && fits; ++par)
                    for (int l0 = m + par; l0 <= L; l0 += 6) {
                        if (t >= 256) { fits = false; break; }
                        for (int u = 0; u < 3; ++u) {
                            const int l = l0 + 2 * u;
                            if (l > L) break;
                            const size_t idx = (size_t)poff[m] + l - m;
                            lmc[u * 256 + t] = l | (m << 8);
                            for (int th = 0; th < nth; ++th) PTc[(size_t)th * 768 + u * 256 + t] = P[idx * nt + th];
                        }
                        ++t;
                    }
            if (fits) {
                c->chain_chunks = t;
                (void)mtip_copy(c, c->d_PTc, PTc.data(), PTc.size() * sizeof(double), hipMemcpyHostToDevice);
                (void)mtip_copy(c, c->d_lmc, lmc.data(), lmc.size() * sizeof(int), hipMemcpyHostToDevice);
            }
        }
    }
    // twiddles exp(-2 pi i j / n_phi), j < n_phi/2
    std::vector<double2> tw(c->np / 2);
    for (int j = 0; j < c->np / 2; ++j) {
        const double a = -2.0 * pi * j / c->np;
        tw[j] = make_double2(std::cos(a), std::sin(a));
    }
    (void)mtip_copy(c, c->d_tw, tw.data(), tw.size() * sizeof(double2), hipMemcpyHostToDevice);
    if (c->d_twN != nullptr) {                       // full-circle table for the register FFT kernels
        std::vector<double2> twn(c->np);
        for (int j = 0; j < c->np; ++j) {
            const double a = -2.0 * pi * j / c->np;
            twn[j] = make_double2(std::cos(a), std::sin(a));
        }
        (void)mtip_copy(c, c->d_twN, twn.data(), twn.size() * sizeof(double2), hipMemcpyHostToDevice);
    }
}

// ------------------------------------------------------------------------------------------------
// forward: FFT rows  (grid rows -> g[row][m+L] * w_theta * 2pi/n_phi)
// ------------------------------------------------------------------------------------------------
template <int PRE>
__global__ void __launch_bounds__(256) k_fft_fwd(const double2* __restrict__ grid, double2* __restrict__ g,
                                                 const double2* __restrict__ tw, const double* __restrict__ gw,
                                                 int np, int nt, int L, long long nrows, double norm,
                                                 const int* __restrict__ slot, int which, int B, long long rows_per_b) {
    HIP_DYNAMIC_SHARED(double2, sm)
    const int T = np >> 1;
    const int R = blockDim.x / T;
    const int r = threadIdx.x / T;
    const int i = threadIdx.x - r * T;
    const long long row = (long long)blockIdx.x * R + r;
    const bool active = row < nrows;
    double2* x = sm + (size_t)r * np;
    double2* y = sm + (size_t)(R + r) * np;
    if (active) {
        long long srow = row;                       // slot-indirect input: (3,B,G) pair array
        if (slot != nullptr) srow += (long long)slot[(row / rows_per_b) * SL_N + which] * B * rows_per_b;
        double2 a = grid[srow * np + i];
        double2 b = grid[srow * np + i + T];
        if (PRE == MTIP_PRE_SQUARE) {
            a = make_double2(cabs2(a), 0.0);
            b = make_double2(cabs2(b), 0.0);
        } else if (PRE == MTIP_PRE_ABS) {
            a = make_double2(sqrt(cabs2(a)), 0.0);
            b = make_double2(sqrt(cabs2(b)), 0.0);
        }
        x[i] = a;
        x[i + T] = b;
    }
    __syncthreads();
    for (int p = 1; p < np; p <<= 1) {
        if (active) {
            const int k = i & (p - 1);
            const int j = ((i - k) << 1) + k;
            const double2 w = tw[k * (T / p)];
            const double2 u0 = x[i];
            const double2 u1 = cmul(x[i + T], w);
            y[j] = cadd(u0, u1);
            y[j + p] = csub(u0, u1);
        }
        __syncthreads();
        double2* t = x;
        x = y;
        y = t;
    }
    if (active) {
        const int theta = (int)(row % nt);
        const double scale = gw[theta] * norm;
        const int nm = 2 * L + 1;
        for (int mi = i; mi < nm; mi += T) {
            const int m = mi - L;
            const int k = m < 0 ? m + np : m;
            g[row * nm + mi] = cscale(x[k], scale);
        }
    }
}

// forward: Legendre analysis  c[b,q,lm] = sum_theta P_lm(theta) g[b,q,theta,m]
__global__ void __launch_bounds__(256) k_leg_fwd(const double2* __restrict__ g, double2* __restrict__ coeff,
                                                 const double* __restrict__ P, const int* __restrict__ poff,
                                                 int nt, int L, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int nlm = (L + 1) * (L + 1);
    const int nm = 2 * L + 1;
    const int lm = (int)(idx % nlm);
    const long long bq = idx / nlm;
    const int l = isqrt_lm(lm);
    const int m = lm - l * (l + 1);
    const int am = m < 0 ? -m : m;
    const double* prow = P + (size_t)(poff[am] + l - am) * nt;
    const double2* gp = g + (size_t)bq * nt * nm + (m + L);
    double ar = 0.0, ai = 0.0;
    for (int t = 0; t < nt; ++t) {
        const double p = prow[t];
        const double2 v = gp[(size_t)t * nm];
        ar = fma(p, v.x, ar);
        ai = fma(p, v.y, ai);
    }
    if (m < 0 && (am & 1)) {
        ar = -ar;
        ai = -ai;
    }
    coeff[idx] = make_double2(ar, ai);
}

// ------------------------------------------------------------------------------------------------
// inverse: Legendre synthesis  g[b,q,theta,m] = sum_l P_lm(theta) c[b,q,lm]
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_leg_inv(const double2* __restrict__ coeff, double2* __restrict__ g,
                                                 const double* __restrict__ P, const int* __restrict__ poff,
                                                 int nt, int L, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int nlm = (L + 1) * (L + 1);
    const int nm = 2 * L + 1;
    const int mi = (int)(idx % nm);
    const long long rest = idx / nm;
    const int t = (int)(rest % nt);
    const long long bq = rest / nt;
    const int m = mi - L;
    const int am = m < 0 ? -m : m;
    const double* pcol = P + (size_t)poff[am] * nt + t;
    const double2* cp = coeff + (size_t)bq * nlm;
    double ar = 0.0, ai = 0.0;
    for (int l = am; l <= L; ++l) {
        const double p = pcol[(size_t)(l - am) * nt];
        const double2 v = cp[l * (l + 1) + m];
        ar = fma(p, v.x, ar);
        ai = fma(p, v.y, ai);
    }
    if (m < 0 && (am & 1)) {
        ar = -ar;
        ai = -ai;
    }
    g[idx] = make_double2(ar, ai);
}

// inverse: FFT rows with fused epilogue
template <int EPI>
__global__ void __launch_bounds__(256) k_fft_inv(const double2* __restrict__ g, double2* __restrict__ grid,
                                                 const double2* __restrict__ tw, int np, int nt, int L, int Nq,
                                                 long long nrows, const double2* __restrict__ Fin,
                                                 const double* __restrict__ shell_scale, const int* __restrict__ slot,
                                                 int which, int B) {
    HIP_DYNAMIC_SHARED(double2, sm)
    const int T = np >> 1;
    const int R = blockDim.x / T;
    const int r = threadIdx.x / T;
    const int i = threadIdx.x - r * T;
    const long long row = (long long)blockIdx.x * R + r;
    const bool active = row < nrows;
    double2* x = sm + (size_t)r * np;
    double2* y = sm + (size_t)(R + r) * np;
    const int nm = 2 * L + 1;
    if (active) {
        for (int e = i; e < np; e += T) {
            double2 v = make_double2(0.0, 0.0);
            if (e <= L) v = g[row * nm + (e + L)];
            else if (e >= np - L) v = g[row * nm + (e - np + L)];
            x[e] = v;
        }
    }
    __syncthreads();
    for (int p = 1; p < np; p <<= 1) {
        if (active) {
            const int k = i & (p - 1);
            const int j = ((i - k) << 1) + k;
            double2 w = tw[k * (T / p)];
            w.y = -w.y;                           // inverse transform: conjugate twiddle
            const double2 u0 = x[i];
            const double2 u1 = cmul(x[i + T], w);
            y[j] = cadd(u0, u1);
            y[j + p] = csub(u0, u1);
        }
        __syncthreads();
        double2* t = x;
        x = y;
        y = t;
    }
    if (active) {
        const int q = (int)((row / nt) % Nq);
        for (int e = i; e < np; e += T) {
            double2 v = x[e];
            const long long o = row * np + e;
            long long oo = o;                           // slot-indirect output: (3,B,G) pair array
            if (slot != nullptr) {
                const long long rows_per_b = (long long)Nq * nt;
                oo += (long long)slot[(row / rows_per_b) * SL_N + which] * B * rows_per_b * np;
            }
            if (EPI == EPI_MODULUS) {
                // project_to_modified_intensity, fxs_Projections.py:899-909
                const double2 Fv = Fin[o];
                const double I = cabs2(Fv);
                const bool ok = (I >= 0.0) && (v.x >= 0.0);
                const double mult = ok ? sqrt(v.x / I) : 0.0;
                v = cscale(Fv, mult);
            } else if (EPI == EPI_SCALE_SHELL) {
                v = cscale(v, shell_scale[q]);
            }
            grid[oo] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
static void fft_launch_dims(const mtip_ctx* c, long long nrows, dim3* grid, dim3* block, size_t* smem) {
    const int T = c->np / 2;
    int R = 256 / T;
    if (R < 1) R = 1;
    *block = dim3((unsigned)(R * T));
    *grid = dim3((unsigned)div_up(nrows, R));
    *smem = (size_t)2 * R * c->np * sizeof(double2);
}

void launch_sht_forward(mtip_ctx* c, const double2* grid, double2* coeff, int prologue, int in_slot) {
    ProfScope ps(c, "sht_fwd");
    if (sht_reg_supported(c)) {
        launch_sht_forward_reg(c, grid, coeff, prologue, in_slot);
        return;
    }
    if (sht_fused_supported(c)) {
        launch_sht_forward_fused(c, grid, coeff, prologue, in_slot);
        return;
    }
    const long long nrows = (long long)c->B * c->N * c->nt;
    dim3 gr, bl;
    size_t sm;
    fft_launch_dims(c, nrows, &gr, &bl, &sm);
    const double norm = 2.0 * 3.14159265358979323846 / c->np;
    const int* sl = in_slot >= 0 ? c->d_slot : nullptr;
    if (prologue == MTIP_PRE_SQUARE)
        hipLaunchKernelGGL(k_fft_fwd<MTIP_PRE_SQUARE>, gr, bl, sm, c->stream, grid, c->d_g, c->d_tw, c->d_gw, c->np, c->nt, c->L, nrows, norm, sl, in_slot, c->B, (long long)c->N * c->nt);
    else if (prologue == MTIP_PRE_ABS)
        hipLaunchKernelGGL(k_fft_fwd<MTIP_PRE_ABS>, gr, bl, sm, c->stream, grid, c->d_g, c->d_tw, c->d_gw, c->np, c->nt, c->L, nrows, norm, sl, in_slot, c->B, (long long)c->N * c->nt);
    else
        hipLaunchKernelGGL(k_fft_fwd<MTIP_PRE_NONE>, gr, bl, sm, c->stream, grid, c->d_g, c->d_tw, c->d_gw, c->np, c->nt, c->L, nrows, norm, sl, in_slot, c->B, (long long)c->N * c->nt);
    const long long total = (long long)c->B * c->N * c->nlm;
    hipLaunchKernelGGL(k_leg_fwd, dim3((unsigned)div_up(total, 256)), dim3(256), 0, c->stream,
                       (const double2*)c->d_g, coeff, (const double*)c->d_P, (const int*)c->d_poff, c->nt, c->L, total);
}

void launch_sht_inverse(mtip_ctx* c, const double2* coeff, double2* grid, const InvEpilogue& epi) {
    // timer families by epilogue: they move different amounts of data
    ProfScope ps(c, epi.mode == EPI_REAL_UPDATE ? "sht_inv_real" : (epi.mode == EPI_MODULUS || epi.mode == EPI_MODULUS_FIXED) ? "sht_inv_modulus" : "sht_inv");
    if (sht_reg_supported(c)) {
        launch_sht_inverse_reg(c, coeff, grid, epi);
        return;
    }
    if (sht_fused_supported(c)) {
        launch_sht_inverse_fused(c, coeff, grid, epi);
        return;
    }
    const long long nrows = (long long)c->B * c->N * c->nt;
    const long long total = nrows * c->nm;
    hipLaunchKernelGGL(k_leg_inv, dim3((unsigned)div_up(total, 256)), dim3(256), 0, c->stream, coeff, c->d_g,
                       (const double*)c->d_P, (const int*)c->d_poff, c->nt, c->L, total);
    dim3 gr, bl;
    size_t sm;
    fft_launch_dims(c, nrows, &gr, &bl, &sm);
    const double2* g = c->d_g;
    const double2* tw = c->d_tw;
    const int* sl = epi.out_slot >= 0 ? c->d_slot : nullptr;
    switch (epi.mode) {
        case EPI_MODULUS:
            hipLaunchKernelGGL(k_fft_inv<EPI_MODULUS>, gr, bl, sm, c->stream, g, grid, tw, c->np, c->nt, c->L, c->N, nrows, epi.F, epi.shell_scale, sl, epi.out_slot, c->B);
            break;
        case EPI_SCALE_SHELL:
            hipLaunchKernelGGL(k_fft_inv<EPI_SCALE_SHELL>, gr, bl, sm, c->stream, g, grid, tw, c->np, c->nt, c->L, c->N, nrows, epi.F, epi.shell_scale, sl, epi.out_slot, c->B);
            break;
        default:
            hipLaunchKernelGGL(k_fft_inv<EPI_STORE>, gr, bl, sm, c->stream, g, grid, tw, c->np, c->nt, c->L, c->N, nrows, epi.F, epi.shell_scale, sl, epi.out_slot, c->B);
            break;
    }
}
