// Single-kernel spherical-harmonic transforms (row a4 of SURVEY section 8): one workgroup per radial shell.
//
//   forward:  for each pass of RP grid rows (theta_j and its mirror n_theta-1-theta_j):
//             HBM -> LDS (fused prologue) -> Stockham FFT along phi in LDS -> fold the mirror pair into
//             even/odd parts (x Gauss weight) -> every thread accumulates "its" (l,m>=0) coefficients in
//             registers: c_lm += P_lm(theta_j) * E/O[j][m]  (and the -m twin with the same P).
//   inverse:  coefficients of the shell -> LDS; Legendre synthesis of a chunk of theta rows into LDS
//             (even/odd split gives theta and its mirror from one recursion-free table walk), then inverse
//             FFT passes along phi with the fused epilogue and coalesced HBM stores.
//
// The grid array is read / written exactly once and the (theta, m) intermediate never leaves the CU
// (the unfused path in k_sht.hip writes + re-reads a (Nq, n_theta, 2L+1) array: ~2x the traffic).
// Legendre values come from the table PT[theta < n_theta/2][(l,m) pair index] (L2 resident, 144 KB at
// L = 32): forward lanes walk consecutive pair indices (coalesced), inverse threads walk l for fixed (theta, m).
// Symmetries used: Y_l,-m(theta,0) = (-1)^m Y_lm(theta,0);  Y_lm(pi - theta, 0) = (-1)^(l+m) Y_lm(theta, 0).
#include "mtip_internal.h"

#define SF_THREADS 256

__device__ __forceinline__ void stockham_pass_lds(double2*& x, double2*& y, const double2* __restrict__ tw, int np, int T,
                                                  int i, bool active, bool inverse) {
    for (int p = 1; p < np; p <<= 1) {
        if (active) {
            const int k = i & (p - 1);
            const int j = ((i - k) << 1) + k;
            double2 w = tw[k * (T / p)];
            if (inverse) w.y = -w.y;
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
}

// ------------------------------------------------------------------------------------------------------
template <int PRE, int MAXI>
__global__ void __launch_bounds__(SF_THREADS) k_sht_fwd_fused(const double2* __restrict__ grid, double2* __restrict__ coeff,
                                                              const double* __restrict__ PT, const int* __restrict__ lmtab,
                                                              const double2* __restrict__ tw_g, const double* __restrict__ gw,
                                                              int np, int nt, int L, int npairs, int rpe, double norm,
                                                              const int* __restrict__ slot, int which, int B, int Nq) {
    HIP_DYNAMIC_SHARED(double2, sm)
    const int T = np >> 1;
    const int nm = 2 * L + 1;
    const int nlm = (L + 1) * (L + 1);
    double2* tw = sm;                               // T
    double2* fx = tw + T;                           // rpe * np
    double2* fy = fx + (size_t)rpe * np;            // rpe * np
    double2* ge = fy + (size_t)rpe * np;            // (rpe/2) * nm   even part
    double2* go = ge + (size_t)(rpe / 2) * nm;      // (rpe/2) * nm   odd part
    const int tid = threadIdx.x;
    const long long shell = blockIdx.x;             // b * Nq + q
    const int r = tid / T;                          // local row of the FFT pass
    const int i = tid - r * T;
    const bool active = r < rpe;
    for (int e = tid; e < T; e += blockDim.x) tw[e] = tw_g[e];
    long long src_shell = shell;                    // slot-indirect input: (3, B, Nq, ...) pair array
    if (slot != nullptr) src_shell += (long long)slot[(shell / Nq) * SL_N + which] * B * Nq;
    const double2* gsrc = grid + (size_t)src_shell * nt * np;
    // pair indices owned by this thread
    int my_l[MAXI], my_m[MAXI];
    double2 accp[MAXI], accm[MAXI];
#pragma unroll
    for (int u = 0; u < MAXI; ++u) {
        const int idx = tid + u * SF_THREADS;
        const int lm = idx < npairs ? lmtab[idx] : 0;
        my_l[u] = lm & 0xff;
        my_m[u] = lm >> 8;
        accp[u] = make_double2(0.0, 0.0);
        accm[u] = make_double2(0.0, 0.0);
    }
    const int half = rpe >> 1;
    const int n_pass = nt / rpe;
    __syncthreads();
    for (int pass = 0; pass < n_pass; ++pass) {
        double2* x = fx + (size_t)r * np;
        double2* y = fy + (size_t)r * np;
        if (active) {
            const int j = r >> 1;
            const int th = pass * half + j;
            const int row = (r & 1) ? (nt - 1 - th) : th;
            double2 a = gsrc[(size_t)row * np + i];
            double2 b = gsrc[(size_t)row * np + i + T];
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
        stockham_pass_lds(x, y, tw, np, T, i, active, false);
        // fold theta / mirror into even and odd parts (x now points at the transformed rows)
        double2* xr = (x == fx + (size_t)r * np) ? fx : fy;
        for (int e = tid; e < half * nm; e += blockDim.x) {
            const int j = e / nm, mi = e - j * nm;
            const int m = mi - L;
            const int k = m < 0 ? m + np : m;
            const double2 a = xr[(size_t)(2 * j) * np + k];
            const double2 b = xr[(size_t)(2 * j + 1) * np + k];
            const double sc = gw[pass * half + j] * norm;
            ge[e] = make_double2((a.x + b.x) * sc, (a.y + b.y) * sc);
            go[e] = make_double2((a.x - b.x) * sc, (a.y - b.y) * sc);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < MAXI; ++u) {
            const int idx = tid + u * SF_THREADS;
            if (idx < npairs) {
                const int l = my_l[u], m = my_m[u];
                const double2* src = ((l + m) & 1) ? go : ge;
                for (int j = 0; j < half; ++j) {
                    const double p = PT[(size_t)(pass * half + j) * npairs + idx];
                    const double2 vp = src[j * nm + L + m];
                    const double2 vm = src[j * nm + L - m];
                    accp[u].x = fma(p, vp.x, accp[u].x);
                    accp[u].y = fma(p, vp.y, accp[u].y);
                    accm[u].x = fma(p, vm.x, accm[u].x);
                    accm[u].y = fma(p, vm.y, accm[u].y);
                }
            }
        }
        // the next pass overwrites fx/fy only after its own barrier; ge/go are rewritten after >= 1 barrier
    }
    double2* cdst = coeff + (size_t)shell * nlm;
#pragma unroll
    for (int u = 0; u < MAXI; ++u) {
        const int idx = tid + u * SF_THREADS;
        if (idx < npairs) {
            const int l = my_l[u], m = my_m[u];
            cdst[l * (l + 1) + m] = accp[u];
            if (m > 0) cdst[l * (l + 1) - m] = (m & 1) ? make_double2(-accm[u].x, -accm[u].y) : accm[u];
        }
    }
}

// ------------------------------------------------------------------------------------------------------
template <int EPI>
__global__ void __launch_bounds__(SF_THREADS) k_sht_inv_fused(const double2* __restrict__ coeff, double2* __restrict__ grid,
                                                              const double* __restrict__ PT, const int* __restrict__ poff,
                                                              const double2* __restrict__ tw_g, int np, int nt, int L,
                                                              int npairs, int rpe, int jl, int Nq,
                                                              const double2* __restrict__ Fin,
                                                              const double* __restrict__ shell_scale,
                                                              const int* __restrict__ slot, int which, int B) {
    HIP_DYNAMIC_SHARED(double2, sm)
    const int T = np >> 1;
    const int nm = 2 * L + 1;
    const int nlm = (L + 1) * (L + 1);
    double2* tw = sm;                               // T
    double2* cl = tw + T;                           // nlm
    double2* gl = cl + nlm;                         // 2*jl rows * nm
    double2* fx = gl + (size_t)2 * jl * nm;         // rpe * np
    double2* fy = fx + (size_t)rpe * np;
    const int tid = threadIdx.x;
    const long long shell = blockIdx.x;
    const int q = (int)(shell % Nq);
    const double2* csrc = coeff + (size_t)shell * nlm;
    for (int e = tid; e < T; e += blockDim.x) tw[e] = tw_g[e];
    for (int e = tid; e < nlm; e += blockDim.x) cl[e] = csrc[e];
    long long dst_shell = shell;
    if (slot != nullptr) dst_shell += (long long)slot[(shell / Nq) * SL_N + which] * B * Nq;
    double2* gdst = grid + (size_t)dst_shell * nt * np;
    const double2* fsrc = Fin ? Fin + (size_t)shell * nt * np : nullptr;
    const int nth = nt >> 1;                        // northern thetas
    const int ipt = (L + 2) / 2;                    // items per theta: pairs (mm, L - mm)
    const int r = tid / T;
    const int i = tid - r * T;
    __syncthreads();
    for (int th0 = 0; th0 < nth; th0 += jl) {
        const int jn = min(jl, nth - th0);
        // ---- Legendre synthesis of thetas th0 .. th0+jn-1 and their mirrors into gl
        for (int item = tid; item < jn * ipt; item += blockDim.x) {
            const int j = item / ipt, mm = item - j * ipt;
            const int th = th0 + j;
            const double* prow = PT + (size_t)th * npairs;
            double2* g_n = gl + (size_t)(2 * j) * nm;       // theta
            double2* g_s = g_n + nm;                        // mirror
            for (int pass = 0; pass < 2; ++pass) {
                const int m = pass == 0 ? mm : L - mm;
                if (pass == 1 && m == mm) break;            // middle m of an even L is done once
                const double* pp = prow + poff[m];
                double2 ep = make_double2(0.0, 0.0), op = ep, em = ep, om = ep;
                for (int l = m; l <= L; ++l) {
                    const double p = pp[l - m];
                    const double2 cp = cl[l * (l + 1) + m];
                    const double2 cm = cl[l * (l + 1) - m];
                    if ((l - m) & 1) {
                        op.x = fma(p, cp.x, op.x); op.y = fma(p, cp.y, op.y);
                        om.x = fma(p, cm.x, om.x); om.y = fma(p, cm.y, om.y);
                    } else {
                        ep.x = fma(p, cp.x, ep.x); ep.y = fma(p, cp.y, ep.y);
                        em.x = fma(p, cm.x, em.x); em.y = fma(p, cm.y, em.y);
                    }
                }
                // parity of (l+m) = parity of (l-m): even part symmetric, odd part antisymmetric under the mirror
                const double sg = (m & 1) ? -1.0 : 1.0;
                g_n[L + m] = cadd(ep, op);
                g_s[L + m] = csub(ep, op);
                if (m > 0) {
                    g_n[L - m] = make_double2(sg * (em.x + om.x), sg * (em.y + om.y));
                    g_s[L - m] = make_double2(sg * (em.x - om.x), sg * (em.y - om.y));
                }
            }
        }
        __syncthreads();
        // ---- inverse FFT passes over the 2*jn rows
        for (int r0 = 0; r0 < 2 * jn; r0 += rpe) {
            const int lr = r0 + r;                          // local row in gl
            const bool active = (r < rpe) && (lr < 2 * jn);
            double2* x = fx + (size_t)r * np;
            double2* y = fy + (size_t)r * np;
            if (active) {
                const double2* gr = gl + (size_t)lr * nm;
                for (int e = i; e < np; e += T) {
                    double2 v = make_double2(0.0, 0.0);
                    if (e <= L) v = gr[e + L];
                    else if (e >= np - L) v = gr[e - np + L];
                    x[e] = v;
                }
            }
            __syncthreads();
            stockham_pass_lds(x, y, tw, np, T, i, active, true);
            if (active) {
                const int th = th0 + (lr >> 1);
                const int row = (lr & 1) ? (nt - 1 - th) : th;
                for (int e = i; e < np; e += T) {
                    double2 v = x[e];
                    const size_t o = (size_t)row * np + e;
                    if (EPI == EPI_MODULUS) {
                        // project_to_modified_intensity, fxs_Projections.py:899-909
                        const double2 Fv = fsrc[o];
                        const double I = cabs2(Fv);
                        const bool ok = (I >= 0.0) && (v.x >= 0.0);
                        const double mult = ok ? sqrt(v.x / I) : 0.0;
                        v = cscale(Fv, mult);
                    } else if (EPI == EPI_SCALE_SHELL) {
                        v = cscale(v, shell_scale[q]);
                    }
                    gdst[o] = v;
                }
            }
            __syncthreads();                                 // fx/fy are refilled by the next pass
        }
    }
}

// ------------------------------------------------------------------------------------------------------
bool sht_fused_supported(const mtip_ctx* c) {
    return c->np <= SF_THREADS && c->np >= 4 && (c->nt % 2) == 0 && c->d_PT != nullptr && !c->sht_unfused;
}

static int fused_rpe(const mtip_ctx* c) {
    int rp = SF_THREADS / (c->np / 2);
    if (rp > c->nt) rp = c->nt;
    if (rp < 2) rp = 2;                  // np = 512: T = 256 -> one row per pass is not supported (needs a pair)
    while (c->nt % rp) --rp;             // passes must tile n_theta; rp stays even because n_theta is even
    if (rp & 1) rp = 2;
    return rp;
}

template <int PRE>
static void launch_fwd_t(mtip_ctx* c, const double2* grid, double2* coeff, int in_slot) {
    const int rpe = fused_rpe(c);
    const int T = c->np / 2;
    const size_t smem = ((size_t)T + 2 * (size_t)rpe * c->np + (size_t)rpe * c->nm) * sizeof(double2);
    const double norm = 2.0 * 3.14159265358979323846 / c->np;
    const int* sl = in_slot >= 0 ? c->d_slot : nullptr;
    const dim3 gr((unsigned)(c->B * c->N)), bl(SF_THREADS);
    const int per = div_up(c->npairs, SF_THREADS);
#define FWD_ARGS grid, coeff, (const double*)c->d_PT, (const int*)c->d_lmtab, (const double2*)c->d_tw, (const double*)c->d_gw, \
                 c->np, c->nt, c->L, c->npairs, rpe, norm, sl, in_slot, c->B, c->N
    if (per <= 3) hipLaunchKernelGGL((k_sht_fwd_fused<PRE, 3>), gr, bl, smem, c->stream, FWD_ARGS);
    else if (per <= 5) hipLaunchKernelGGL((k_sht_fwd_fused<PRE, 5>), gr, bl, smem, c->stream, FWD_ARGS);
    else hipLaunchKernelGGL((k_sht_fwd_fused<PRE, 9>), gr, bl, smem, c->stream, FWD_ARGS);
#undef FWD_ARGS
}

void launch_sht_forward_fused(mtip_ctx* c, const double2* grid, double2* coeff, int prologue, int in_slot) {
    if (prologue == MTIP_PRE_SQUARE) launch_fwd_t<MTIP_PRE_SQUARE>(c, grid, coeff, in_slot);
    else if (prologue == MTIP_PRE_ABS) launch_fwd_t<MTIP_PRE_ABS>(c, grid, coeff, in_slot);
    else launch_fwd_t<MTIP_PRE_NONE>(c, grid, coeff, in_slot);
}

void launch_sht_inverse_fused(mtip_ctx* c, const double2* coeff, double2* grid, const InvEpilogue& epi) {
    const int rpe = fused_rpe(c);
    const int T = c->np / 2;
    const int nth = c->nt / 2;
    const int ipt = (c->L + 2) / 2;
    const int nchunks = std::max(1, div_up((long long)nth * ipt, SF_THREADS));
    const int jl = div_up(nth, nchunks);
    const size_t smem = ((size_t)T + c->nlm + (size_t)2 * jl * c->nm + 2 * (size_t)rpe * c->np) * sizeof(double2);
    const int* sl = epi.out_slot >= 0 ? c->d_slot : nullptr;
    const dim3 gr((unsigned)(c->B * c->N)), bl(SF_THREADS);
#define INV_ARGS coeff, grid, (const double*)c->d_PT, (const int*)c->d_poff, (const double2*)c->d_tw, c->np, c->nt, c->L, \
                 c->npairs, rpe, jl, c->N, epi.F, epi.shell_scale, sl, epi.out_slot, c->B
    switch (epi.mode) {
        case EPI_MODULUS: hipLaunchKernelGGL(k_sht_inv_fused<EPI_MODULUS>, gr, bl, smem, c->stream, INV_ARGS); break;
        case EPI_SCALE_SHELL: hipLaunchKernelGGL(k_sht_inv_fused<EPI_SCALE_SHELL>, gr, bl, smem, c->stream, INV_ARGS); break;
        default: hipLaunchKernelGGL(k_sht_inv_fused<EPI_STORE>, gr, bl, smem, c->stream, INV_ARGS); break;
    }
#undef INV_ARGS
}
