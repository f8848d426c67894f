// Spherical-harmonic transforms with register-resident FFTs (row a4 of SURVEY section 8), one workgroup per shell.
//
// The phi-FFT of length N = R1 * R2 (16 = 4x4, 32 = 4x8, 64 = 8x8, 128 = 8x16, 256 = 16x16) is done as two
// rounds of small FFTs held entirely in registers with ONE transpose through LDS in between (Cooley-Tukey
// n = R2 n1 + n2, k = k1 + R1 k2), instead of log2(N) LDS round trips + barriers per row:
//   forward  thread (pair j, n2): loads x[R2 n1 + n2] of row theta_j and of its mirror row (coalesced), folds them
//            into even/odd parts (FFT is linear, so the mirror fold happens before the transform), two R1-point
//            FFTs, twiddle w_N^(n2 k1), store to LDS;   thread (row, k1): R2-point FFT, keeps |m| <= L, scales
//            by the Gauss weight and writes the (theta, m) panel to LDS;   then every thread accumulates its
//            (l, m >= 0) coefficients in registers:  c_lm += P_lm(theta_j) * E/O[j][m]  (table PT, L2 resident).
//   inverse  Legendre synthesis of a block of thetas (+ mirrors) into LDS, R2-point inverse FFTs over k2, twiddle,
//            LDS transpose, R1-point inverse FFTs, fused epilogue, coalesced stores.
// ~4 barriers per 32 (forward) / 16 (inverse) grid rows; ~5x fewer instructions than the LDS Stockham variant
// (k_sht_fused.hip), which stays as the fallback for other n_phi.
#include "mtip_internal.h"
#include "k_sht_common.h"
#include "k_sht_legendre.h"

// (l, m) pairs per thread from which the table loads of eight theta pairs are requested together: with few rows per
// pass and many pairs per thread (n_phi = 256) every single load was a full L2 round trip (357 -> 260 us at 256 x L48);
// at n_phi = 128 (3 pairs per thread) the extra registers cost more than they save (54 -> 59 us)
#define FWD_BATCH_MIN 5
// ------------------------------------------------------------------------------------------------------
template <int PRE, int R1, int R2, int MAXI>
__global__ void __launch_bounds__(SR_THREADS) k_sht_fwd_reg(const double2* __restrict__ grid, double2* __restrict__ coeff,
                                                            const double* __restrict__ PT, const int* __restrict__ lmtab,
                                                            const double2* __restrict__ twN_g, const double* __restrict__ gw,
                                                            int nt, int L, int npairs, int RP, double norm,
                                                            const int* __restrict__ slot, int which, int B, int Nq) {
    constexpr int N = R1 * R2;
    constexpr int AS = R2 + 1;                      // padded row of the transpose buffer
    HIP_DYNAMIC_SHARED(double2, sm)
    const int nm = 2 * L + 1;
    const int nlm = (L + 1) * (L + 1);
    double2* twN = sm;                              // N       exp(-2 pi i j / N)
    double2* A = twN + N;                           // RP * R1 * AS   (aliased by the (theta, m) panel G: RP * nm)
    double2* G = A;
    const int tid = threadIdx.x;
    const long long shell = blockIdx.x;
    for (int e = tid; e < N; e += blockDim.x) twN[e] = twN_g[e];
    long long src_shell = shell;                    // slot-indirect input: (3, B, Nq, ...) pair array
    if (slot != nullptr) src_shell += (long long)slot[(shell / Nq) * SL_N + which] * B * Nq;
    const double2* gsrc = grid + (size_t)src_shell * nt * N;
    int my_l[MAXI], my_m[MAXI];
    double2 accp[MAXI], accm[MAXI];
#pragma unroll
    for (int u = 0; u < MAXI; ++u) {
        const int idx = tid + u * SR_THREADS;
        const int lm = idx < npairs ? lmtab[idx] : 0;
        my_l[u] = lm & 0xff;
        my_m[u] = lm >> 8;
        accp[u] = make_double2(0.0, 0.0);
        accm[u] = make_double2(0.0, 0.0);
    }
    const int half = RP >> 1;
    const int n_pass = nt / RP;
    __syncthreads();
    for (int pass = 0; pass < n_pass; ++pass) {
        // ---- phase 1: fold theta / mirror, R1-point FFTs over n1, twiddle, transpose store
        if (tid < half * R2) {
            const int j = tid / R2, n2 = tid - j * R2;
            const int th = pass * half + j;
            const double2* rn = gsrc + (size_t)th * N + n2;
            const double2* rs = gsrc + (size_t)(nt - 1 - th) * N + n2;
            double2 ev[R1], ov[R1];
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) {
                double2 a = rn[R2 * n1];
                double2 b = rs[R2 * n1];
                if (PRE == MTIP_PRE_SQUARE) {
                    a = make_double2(cabs2(a), 0.0);
                    b = make_double2(cabs2(b), 0.0);
                } else if (PRE == MTIP_PRE_ABS) {
                    a = make_double2(sqrt(cabs2(a)), 0.0);
                    b = make_double2(sqrt(cabs2(b)), 0.0);
                }
                ev[n1] = cadd(a, b);
                ov[n1] = csub(a, b);
            }
            SmallFFT<R1, false>::run(ev);
            SmallFFT<R1, false>::run(ov);
            double2* ae = A + (size_t)(2 * j) * R1 * AS + n2;
            double2* ao = ae + (size_t)R1 * AS;
#pragma unroll
            for (int k1 = 0; k1 < R1; ++k1) {
                const double2 w = twN[n2 * k1];
                ae[k1 * AS] = cmul(ev[k1], w);
                ao[k1 * AS] = cmul(ov[k1], w);
            }
        }
        __syncthreads();
        // ---- phase 2: R2-point FFTs over n2; keep |m| <= L
        double2 uv[R2];
        const bool act2 = tid < RP * R1;
        const int r2 = tid / R1, k1 = tid - r2 * R1;
        if (act2) {
            const double2* ar = A + (size_t)(r2 * R1 + k1) * AS;
#pragma unroll
            for (int n2 = 0; n2 < R2; ++n2) uv[n2] = ar[n2];
        }
        __syncthreads();                                // A is dead from here on: G may overwrite it
        if (act2) {
            SmallFFT<R2, false>::run(uv);
            const double sc = gw[pass * half + (r2 >> 1)] * norm;
            double2* gr = G + (size_t)r2 * nm + L;
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2) {
                const int k = k1 + R1 * k2;
                if (k <= L) gr[k] = cscale(uv[k2], sc);
                else if (k >= N - L) gr[k - N] = cscale(uv[k2], sc);
            }
        }
        __syncthreads();
        // ---- Legendre accumulation (rows 2j = even part, 2j+1 = odd part of theta pair j)
#pragma unroll
        for (int u = 0; u < MAXI; ++u) {
            const int idx = tid + u * SR_THREADS;
            if (idx < npairs) {
                const int l = my_l[u], m = my_m[u];
                const double2* src = G + (size_t)((l + m) & 1) * nm + L;
                const double* pt = PT + (size_t)(pass * half) * npairs + idx;
                double2 ap = accp[u], am = accm[u];
                if (MAXI >= FWD_BATCH_MIN) {
                    // large grids (few rows per pass, many (l, m) per thread): the table values of eight theta pairs
                    // are requested together, otherwise every one of them costs a full L2 round trip
                    for (int jb = 0; jb < half; jb += 8) {
                        double pv[8];
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) pv[jj] = pt[(size_t)min(jb + jj, half - 1) * npairs];
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) {
                            const int j = jb + jj;
                            if (j < half) {
                                const double2 vp = src[(size_t)(2 * j) * nm + m];
                                const double2 vm = src[(size_t)(2 * j) * nm - m];
                                ap.x = fma(pv[jj], vp.x, ap.x);
                                ap.y = fma(pv[jj], vp.y, ap.y);
                                am.x = fma(pv[jj], vm.x, am.x);
                                am.y = fma(pv[jj], vm.y, am.y);
                            }
                        }
                    }
                } else
                for (int j = 0; j < half; ++j) {
                    const double p = pt[(size_t)j * npairs];
                    const double2 vp = src[(size_t)(2 * j) * nm + m];
                    const double2 vm = src[(size_t)(2 * j) * nm - m];
                    ap.x = fma(p, vp.x, ap.x);
                    ap.y = fma(p, vp.y, ap.y);
                    am.x = fma(p, vm.x, am.x);
                    am.y = fma(p, vm.y, am.y);
                }
                accp[u] = ap;
                accm[u] = am;
            }
        }
        __syncthreads();                                // the next pass rewrites A / G
    }
    double2* cdst = coeff + (size_t)shell * nlm;
#pragma unroll
    for (int u = 0; u < MAXI; ++u) {
        const int idx = tid + u * SR_THREADS;
        if (idx < npairs) {
            const int l = my_l[u], m = my_m[u];
            cdst[l * (l + 1) + m] = accp[u];
            if (m > 0) cdst[l * (l + 1) - m] = (m & 1) ? make_double2(-accm[u].x, -accm[u].y) : accm[u];
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// NS shells per workgroup (NS = 1 is what ships; NS = 2 shares the table registers between two shells and measured no faster).  k_sht_fwd_reg above loads, per shell, the
// Legendre table rows of all its (l, m) pairs -- more bytes than the shell's grid rows at 128 x L32 (144 KB against 128 KB) and
// each table value is used once; the loads sit in the accumulation loop, four in flight, so a pass of 16 theta pairs is
// twelve dependent L2 round trips.  Here a workgroup owns TWO shells and half as many theta pairs per pass: the table
// values of a pass (MAXI x TH doubles per thread) are requested at the top of the pass, next to the grid rows, arrive while
// the FFT phases run and are used for both shells -- half the table bytes per shell, and one round trip per pass.
// The phases are those of k_sht_fwd_reg with the thread roles split over the two shells; LDS is the same size.
template <int PRE, int R1, int R2, int MAXI, int TH, int NS>
__global__ void __launch_bounds__(SR_THREADS, 2) k_sht_fwd_pair(const double2* __restrict__ grid, double2* __restrict__ coeff,
                                                             const double* __restrict__ PT, const int* __restrict__ lmtab,
                                                             const double2* __restrict__ twN_g, const double* __restrict__ gw,
                                                             int nt, int L, int npairs, double norm,
                                                             const int* __restrict__ slot, int which, int B, int Nq) {
    constexpr int N = R1 * R2;
    constexpr int AS = R2 + 1;                      // padded row of the transpose buffer
    constexpr int RPS = 2 * TH;                     // panel rows per pass and shell (even + odd part of TH theta pairs)
    constexpr int ASZ = RPS * R1 * AS;              // transpose buffer of one shell (aliased by its (theta, m) panel)
    constexpr int GS = R1 * AS;                     // panel row stride: a compile-time constant >= n_phi + 1 > 2 L + 1
    static_assert(NS * TH * R2 <= SR_THREADS && NS * RPS * R1 <= SR_THREADS, "thread roles");
    HIP_DYNAMIC_SHARED(double2, sm)
    const int nlm = (L + 1) * (L + 1);
    double2* twN = sm;                              // N       exp(-2 pi i j / N)
    double2* A = twN + N;                           // NS * ASZ
    const int tid = threadIdx.x;
    const long long shell0 = (long long)NS * blockIdx.x;
    for (int e = tid; e < N; e += blockDim.x) twN[e] = twN_g[e];
    // phase-1 role: (shell, theta pair, n2); phase-2 role: (shell, panel row, k1)
    constexpr bool ALL1 = NS * TH * R2 == SR_THREADS, ALL2 = NS * RPS * R1 == SR_THREADS;   // every thread has a role: no exec masks
    const bool act1 = ALL1 || tid < NS * TH * R2;
    const int s1 = tid / (TH * R2), j1 = (tid - s1 * TH * R2) / R2, n2 = tid % R2;
    const bool act2 = ALL2 || tid < NS * RPS * R1;
    const int s2 = tid / (RPS * R1), r2 = (tid - s2 * RPS * R1) / R1, k1 = tid % R1;
    long long src_shell = shell0 + (act1 ? s1 : 0);                 // slot-indirect input: (3, B, Nq, ...) pair array
    if (slot != nullptr) src_shell += (long long)slot[((shell0 + (act1 ? s1 : 0)) / Nq) * SL_N + which] * B * Nq;
    const double2* gsrc = grid + (size_t)src_shell * nt * N;
    int my_l[MAXI], my_m[MAXI], my_i[MAXI];
    double2 accp[NS][MAXI], accm[NS][MAXI];
#pragma unroll
    for (int u = 0; u < MAXI; ++u) {
        const int idx = tid + u * SR_THREADS;
        my_i[u] = min(idx, npairs - 1);             // clamped: the table loads stay branch-free
        const int lm = lmtab[my_i[u]];
        my_l[u] = lm & 0xff;
        my_m[u] = lm >> 8;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            accp[s][u] = make_double2(0.0, 0.0);
            accm[s][u] = make_double2(0.0, 0.0);
        }
    }
    const int n_pass = (nt >> 1) / TH;
    __syncthreads();
    for (int pass = 0; pass < n_pass; ++pass) {
        // ---- phase 1: fold theta / mirror, R1-point FFTs over n1, twiddle, transpose store
        double2 ev[R1], ov[R1];
        if (act1) {
            const int th = pass * TH + j1;
            const double2* rn = gsrc + (size_t)th * N + n2;
            const double2* rs = gsrc + (size_t)(nt - 1 - th) * N + n2;
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) {
                ev[n1] = rn[R2 * n1];
                ov[n1] = rs[R2 * n1];
            }
        }
        if (act1) {
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) {
                double2 a = ev[n1], b = ov[n1];
                if (PRE == MTIP_PRE_SQUARE) {
                    a = make_double2(cabs2(a), 0.0);
                    b = make_double2(cabs2(b), 0.0);
                } else if (PRE == MTIP_PRE_ABS) {
                    a = make_double2(sqrt(cabs2(a)), 0.0);
                    b = make_double2(sqrt(cabs2(b)), 0.0);
                }
                ev[n1] = cadd(a, b);
                ov[n1] = csub(a, b);
            }
            SmallFFT<R1, false>::run(ev);
            SmallFFT<R1, false>::run(ov);
            double2* ae = A + (size_t)s1 * ASZ + (size_t)(2 * j1) * R1 * AS + n2;
            double2* ao = ae + (size_t)R1 * AS;
#pragma unroll
            for (int q = 0; q < R1; ++q) {
                const double2 w = twN[n2 * q];
                ae[q * AS] = cmul(ev[q], w);
                ao[q * AS] = cmul(ov[q], w);
            }
        }
        // the table rows of this pass, requested as soon as the registers of phase 1 are free: they arrive while phase 2 runs
        double tab[MAXI][TH];
#pragma unroll
        for (int u = 0; u < MAXI; ++u)
#pragma unroll
            for (int jj = 0; jj < TH; ++jj) {
                const double* row = PT + (size_t)(pass * TH + jj) * npairs;       // uniform base + 32-bit lane offset
                tab[u][jj] = row[(unsigned)my_i[u]];
            }
        __syncthreads();
        // ---- phase 2: R2-point FFTs over n2; keep |m| <= L
        double2 uv[R2];
        if (act2) {
            const double2* ar = A + (size_t)s2 * ASZ + (size_t)(r2 * R1 + k1) * AS;
#pragma unroll
            for (int q = 0; q < R2; ++q) uv[q] = ar[q];
        }
        __syncthreads();                                // A is dead from here on: the panels overwrite it
        if (act2) {
            SmallFFT<R2, false>::run(uv);
            const double sc = gw[pass * TH + (r2 >> 1)] * norm;
            double2* gr = A + (size_t)s2 * ASZ + (size_t)r2 * GS + L;
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2) {
                const int k = k1 + R1 * k2;
                if (k <= L) gr[k] = cscale(uv[k2], sc);
                else if (k >= N - L) gr[k - N] = cscale(uv[k2], sc);
            }
        }
        __syncthreads();
        // ---- Legendre accumulation (rows 2j = even part, 2j+1 = odd part of theta pair j), both shells per table value
#pragma unroll
        for (int u = 0; u < MAXI; ++u) {
            const int l = my_l[u], m = my_m[u];
            const double2* srcp = A + ((l + m) & 1) * GS + L + m;       // one lane base per slot, immediates for (shell, row)
            const double2* srcm = A + ((l + m) & 1) * GS + L - m;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                double2 ap = accp[s][u], am = accm[s][u];
#pragma unroll
                for (int jj = 0; jj < TH; ++jj) {
                    const double p = tab[u][jj];
                    const double2 vp = srcp[s * ASZ + 2 * jj * GS];
                    const double2 vm = srcm[s * ASZ + 2 * jj * GS];
                    ap.x = fma(p, vp.x, ap.x);
                    ap.y = fma(p, vp.y, ap.y);
                    am.x = fma(p, vm.x, am.x);
                    am.y = fma(p, vm.y, am.y);
                }
                accp[s][u] = ap;
                accm[s][u] = am;
            }
        }
        __syncthreads();                                // the next pass rewrites A
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        double2* cdst = coeff + (size_t)(shell0 + s) * nlm;
#pragma unroll
        for (int u = 0; u < MAXI; ++u) {
            const int idx = tid + u * SR_THREADS;
            if (idx < npairs) {
                const int l = my_l[u], m = my_m[u];
                cdst[l * (l + 1) + m] = accp[s][u];
                if (m > 0) cdst[l * (l + 1) - m] = (m & 1) ? make_double2(-accm[s][u].x, -accm[s][u].y) : accm[s][u];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------------
template <int EPI, int R1, int R2>
__global__ void __launch_bounds__(SR_THREADS) k_sht_inv_reg(const double2* __restrict__ coeff, double2* __restrict__ grid,
                                                            const double* __restrict__ PT, const int* __restrict__ poff,
                                                            const double2* __restrict__ twN_g, int nt, int L, int npairs,
                                                            int RP, int Nq, const double2* __restrict__ Fin,
                                                            const double* __restrict__ shell_scale,
                                                            const int* __restrict__ slot, int which, int B) {
    constexpr int N = R1 * R2;
    constexpr int AS = R2 + 1;
    HIP_DYNAMIC_SHARED(double2, sm)
    const int nm = 2 * L + 1;
    const int nlm = (L + 1) * (L + 1);
    double2* twN = sm;                              // N
    double2* cl = twN + N;                          // nlm
    double2* Gs = cl + nlm;                         // RP * nm        spectra: row 2j = theta_j, 2j+1 = mirror
    double2* Bm = Gs + (size_t)RP * nm;             // RP * R1 * AS   transpose buffer
    const int tid = threadIdx.x;
    const long long shell = blockIdx.x;
    const int q = (int)(shell % Nq);
    const double2* csrc = coeff + (size_t)shell * nlm;
    for (int e = tid; e < N; e += blockDim.x) twN[e] = twN_g[e];
    for (int e = tid; e < nlm; e += blockDim.x) cl[e] = csrc[e];
    long long dst_shell = shell;
    if (slot != nullptr) dst_shell += (long long)slot[(shell / Nq) * SL_N + which] * B * Nq;
    double2* gdst = grid + (size_t)dst_shell * nt * N;
    const double2* fsrc = Fin ? Fin + (size_t)shell * nt * N : nullptr;
    const int half = RP >> 1;
    const int n_pass = nt / RP;
    const int ipt = (L + 2) / 2;                    // items per theta: m pairs (mm, L - mm) of equal total length
    __syncthreads();
    for (int pass = 0; pass < n_pass; ++pass) {
        // ---- Legendre synthesis of `half` thetas and their mirrors
        for (int item = tid; item < half * ipt; item += blockDim.x) {
            const int j = item / ipt, mm = item - j * ipt;
            const int th = pass * half + j;
            const double* prow = PT + (size_t)th * npairs;
            double2* g_n = Gs + (size_t)(2 * j) * nm + L;
            double2* g_s = g_n + nm;
            for (int side = 0; side < 2; ++side) {
                const int m = side == 0 ? mm : L - mm;
                if (side == 1 && m == mm) break;        // middle m of an even L is done once
                const double* pp = prow + poff[m] - m;
                double2 ep = make_double2(0.0, 0.0), op = ep, em = ep, om = ep;
                for (int l = m; l <= L; ++l) {
                    const double p = pp[l];
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
                const double sg = (m & 1) ? -1.0 : 1.0;
                g_n[m] = cadd(ep, op);
                g_s[m] = csub(ep, op);
                if (m > 0) {
                    g_n[-m] = make_double2(sg * (em.x + om.x), sg * (em.y + om.y));
                    g_s[-m] = make_double2(sg * (em.x - om.x), sg * (em.y - om.y));
                }
            }
        }
        __syncthreads();
        // ---- step 1: inverse R2-point FFTs over k2 of the zero padded spectrum, twiddle, transpose store
        if (tid < RP * R1) {
            const int r = tid / R1, k1 = tid - r * R1;
            const double2* gr = Gs + (size_t)r * nm + L;
            double2 uv[R2];
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2) {
                const int k = k1 + R1 * k2;
                double2 v = make_double2(0.0, 0.0);
                if (k <= L) v = gr[k];
                else if (k >= N - L) v = gr[k - N];
                uv[k2] = v;
            }
            SmallFFT<R2, true>::run(uv);
            double2* br = Bm + (size_t)(r * R1 + k1) * AS;
#pragma unroll
            for (int n2 = 0; n2 < R2; ++n2) {
                double2 w = twN[n2 * k1];
                w.y = -w.y;
                br[n2] = cmul(uv[n2], w);
            }
        }
        __syncthreads();
        // ---- step 2: inverse R1-point FFTs over k1, epilogue, coalesced store
        if (tid < RP * R2) {
            const int r = tid / R2, n2 = tid - r * R2;
            const double2* br = Bm + (size_t)r * R1 * AS + n2;
            double2 vv[R1];
#pragma unroll
            for (int k1 = 0; k1 < R1; ++k1) vv[k1] = br[k1 * AS];
            SmallFFT<R1, true>::run(vv);
            const int th = pass * half + (r >> 1);
            const int row = (r & 1) ? (nt - 1 - th) : th;
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) {
                double2 v = vv[n1];
                const size_t o = (size_t)row * N + R2 * n1 + n2;
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
        __syncthreads();                                // Gs / Bm are rewritten by the next pass
    }
}

// ------------------------------------------------------------------------------------------------------
// Inverse transform, "wide" variant: the Legendre synthesis of ALL thetas of the shell happens once, with the
// wave as the unit of work -- lanes 0-31 = 32 northern thetas for +m, lanes 32-63 = the same thetas for -m, so
// a wave has one m (uniform trip count, no divergence) and the coefficient c_l,+-m is an LDS broadcast.  Each
// lane runs the three-term recurrence P_lm = a_lm (x P_l-1,m - b_lm P_l-2,m) for its theta in registers
// ("LDS-staged Legendre recursion": a_lm, b_lm sit in LDS, the two start values P_mm, P_m+1,m come from the
// table and are prefetched one item ahead), so the synthesis loop touches no global memory.  Even and odd l-m
// accumulate separately (north = E + O, south = E - O).  Items (m, theta chunk) are dealt to the waves in
// snake order of decreasing length, which balances them to within one column.  Then the spectra of all rows
// sit in LDS and the two register-FFT steps run over RP rows per pass as in k_sht_inv_reg.  One 512-thread
// workgroup per shell and CU (LDS: n_theta (2L+1) spectra + transpose buffer, 147 KB at 64 x 128, L = 32).
template <int EPI, int R1, int R2>
__global__ void __launch_bounds__(SW_THREADS) k_sht_inv_wide(const double2* __restrict__ coeff, double2* __restrict__ grid,
                                                             const double* __restrict__ P, const int* __restrict__ poff,
                                                             const double2* __restrict__ AB, const double* __restrict__ cost,
                                                             int npairs, const double2* __restrict__ twN_g, int nt, int L,
                                                             int RP, int Nq, const double2* __restrict__ Fin,
                                                             const double* __restrict__ shell_scale,
                                                             const int* __restrict__ slot, int which, int B,
                                                             const double2* __restrict__ coeff_sub, RealEpi re, int nsplit) {
    constexpr int N = R1 * R2;
    constexpr int AS = R2 + 1;
    HIP_DYNAMIC_SHARED(double2, sm)
    const int nm = 2 * L + 1;
    const int nlm = (L + 1) * (L + 1);
    // nsplit > 1: a shell is shared by nsplit workgroups, each with nt / nsplit rows (theta_j and its mirror for a
    // contiguous range of j); the twiddles then stay in global memory to make room (256 x L48: 157 KB)
    const int ntl = nt / nsplit;                    // rows of this workgroup
    const double2* twN = nsplit > 1 ? twN_g : sm;   // N
    double2* Gs = sm + (nsplit > 1 ? 0 : N);        // ntl * nm       spectra: row 2j = theta_(j0+j), 2j+1 = its mirror
    double2* ABs = Gs + (size_t)ntl * nm;           // npairs         recurrence coefficients
    double2* cl = ABs + npairs;                     // nlm            (Legendre phase)
    double2* Bm = cl;                               // RP * R1 * AS   transpose buffer (FFT passes; reuses cl)
    const int tid = threadIdx.x;
    const long long shell = blockIdx.x / nsplit;
    const int split = (int)(blockIdx.x - shell * nsplit);
    const int q = (int)(shell % Nq);
    const double2* csrc = coeff + (size_t)shell * nlm;
    // Legendre work items (k_sht_legendre.h): the wave index as a scalar, so that items, m and the l loop are wave-uniform for
    // the compiler too (scalar loop control)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
    const int nth = ntl >> 1;                       // theta pairs of this workgroup, first one j0
    const int j0 = split * nth;
    // start values P_mm, P_m+1,m of the first item: in flight while the tables are staged
    LegendreStart ls;
    legendre_prefetch_first(ls, P, nt, L, nth, j0, wave, tid & 63);
    if (nsplit == 1)
        for (int e = tid; e < N; e += blockDim.x) sm[e] = twN_g[e];
    for (int e = tid; e < npairs; e += blockDim.x) ABs[e] = AB[e];
    if (coeff_sub != nullptr && q > 0) {                    // ft_stab: IFT(F') - IFT(F) on shells > 0 (misk.py:326-329)
        const double2* ssrc = coeff_sub + (size_t)shell * nlm;
        for (int e = tid; e < nlm; e += blockDim.x) cl[e] = csub(csrc[e], ssrc[e]);
    } else {
        for (int e = tid; e < nlm; e += blockDim.x) cl[e] = csrc[e];
    }
    long long dst_shell = shell;
    if (slot != nullptr && which >= 0) dst_shell += (long long)slot[(shell / Nq) * SL_N + which] * B * Nq;
    double2* gdst = grid + (size_t)dst_shell * nt * N;
    const double2* fsrc = Fin ? Fin + (size_t)shell * nt * N : nullptr;
    // EPI_REAL_UPDATE: previous / new density and support of this shell through the slot table
    const double2* rprev = nullptr;
    const uint8_t* rsup = nullptr;
    const uint8_t* rS0 = nullptr;
    double err_num = 0.0, err_den = 0.0;
    if (EPI == EPI_REAL_UPDATE) {
        const int bb = (int)(shell / Nq);
        const int* sl = slot + bb * SL_N;
        const size_t gsh = (size_t)nt * N;
        rprev = re.prev + ((size_t)sl[SL_CUR] * B * Nq + shell) * gsh;
        gdst = re.out + ((size_t)sl[SL_OUT] * B * Nq + shell) * gsh;
        rsup = re.sup + ((size_t)sl[SL_SUP] * B * Nq + shell) * gsh;
        rS0 = re.S0 + (size_t)q * gsh;
    }
    __syncthreads();
    // ---- Legendre synthesis of every row: P_lm(theta) by the three-term recurrence in l (registers only)
    legendre_synthesis_rows(ls, Gs, cl, ABs, P, cost, nt, L, nth, j0, wave, nw, tid & 63);
    __syncthreads();
    const int n_pass = ntl / RP;
    for (int pass = 0; pass < n_pass; ++pass) {
        // epilogue operands of this thread's step-2 outputs: requested now, they arrive behind step 1
        double2 pre[(EPI == EPI_MODULUS || EPI == EPI_REAL_UPDATE) ? R1 : 1];
        uint8_t pre_s[EPI == EPI_REAL_UPDATE ? R1 : 1], pre_0[EPI == EPI_REAL_UPDATE ? R1 : 1];
        if ((EPI == EPI_MODULUS || EPI == EPI_REAL_UPDATE) && tid < RP * R2) {
            const int r = tid / R2, n2 = tid - r * R2;
            const int rr = pass * RP + r;
            const int th = j0 + (rr >> 1);
            const int row = (rr & 1) ? (nt - 1 - th) : th;
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) {
                const size_t o = (size_t)row * N + R2 * n1 + n2;
                if (EPI == EPI_MODULUS) pre[n1] = fsrc[o];
                if (EPI == EPI_REAL_UPDATE) {
                    pre[n1] = rprev[o];
                    pre_s[n1] = rsup[o];
                    pre_0[n1] = re.err_use_mask ? rS0[o] : (uint8_t)1;
                }
            }
        }
        // ---- step 1: inverse R2-point FFTs over k2 of the zero padded spectrum, twiddle, transpose store
        if (tid < RP * R1) {
            const int r = tid / R1, k1 = tid - r * R1;
            const double2* gr = Gs + (size_t)(pass * RP + r) * nm + L;
            double2 uv[R2];
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2) {
                const int k = k1 + R1 * k2;
                double2 v = make_double2(0.0, 0.0);
                if (k <= L) v = gr[k];
                else if (k >= N - L) v = gr[k - N];
                uv[k2] = v;
            }
            SmallFFT<R2, true>::run(uv);
            double2* br = Bm + (size_t)(r * R1 + k1) * AS;
#pragma unroll
            for (int n2 = 0; n2 < R2; ++n2) {
                double2 w = twN[n2 * k1];
                w.y = -w.y;
                br[n2] = cmul(uv[n2], w);
            }
        }
        __syncthreads();
        // ---- step 2: inverse R1-point FFTs over k1, epilogue, coalesced store
        if (tid < RP * R2) {
            const int r = tid / R2, n2 = tid - r * R2;
            const double2* br = Bm + (size_t)r * R1 * AS + n2;
            double2 vv[R1];
#pragma unroll
            for (int k1 = 0; k1 < R1; ++k1) vv[k1] = br[k1 * AS];
            SmallFFT<R1, true>::run(vv);
            const int rr = pass * RP + r;
            const int th = j0 + (rr >> 1);
            const int row = (rr & 1) ? (nt - 1 - th) : th;
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) {
                double2 v = vv[n1];
                const size_t o = (size_t)row * N + R2 * n1 + n2;
                if (EPI == EPI_MODULUS) {
                    // project_to_modified_intensity, fxs_Projections.py:899-909
                    const double2 Fv = pre[n1];
                    const double I = cabs2(Fv);
                    const bool ok = (I >= 0.0) && (v.x >= 0.0);
                    const double mult = ok ? sqrt(v.x / I) : 0.0;
                    v = cscale(Fv, mult);
                } else if (EPI == EPI_SCALE_SHELL) {
                    v = cscale(v, shell_scale[q]);
                } else if (EPI == EPI_REAL_UPDATE) {
                    const double2 pv = pre[n1];
                    const double2 w = (re.add_prev && q > 0 && (re.add_mask == nullptr || re.add_mask[shell / Nq] != 0)) ? cadd(v, pv) : v;
                    double2 P;
                    v = real_update_point(re.rp, re.method, re.beta, w, pv, pre_s[n1] != 0, P);
                    if (pre_0[n1] != 0) {                            // l2_projection_diff, fxs_IO_methods.py:97-128
                        const double wg = re.wr[q] * re.wt[row];
                        const double dx = w.x - P.x, dy = w.y - P.y;
                        err_num = fma(wg, dx * dx + dy * dy, err_num);
                        err_den = fma(wg, w.x * w.x + w.y * w.y, err_den);
                    }
                }
                gdst[o] = v;
            }
        }
        __syncthreads();                                // Bm is rewritten by the next pass
    }
    if (EPI == EPI_REAL_UPDATE) {
        // per-shell error partial sums in a fixed order (bitwise reproducible): wave butterflies, then the waves
        for (int o = 32; o > 0; o >>= 1) {
            err_num += __shfl_xor(err_num, o, 64);
            err_den += __shfl_xor(err_den, o, 64);
        }
        double* red = reinterpret_cast<double*>(Bm);
        if ((tid & 63) == 0) {
            red[2 * (tid >> 6)] = err_num;
            red[2 * (tid >> 6) + 1] = err_den;
        }
        __syncthreads();
        if (tid == 0) {
            double sn = 0.0, sd = 0.0;
            for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) {
                sn += red[2 * wv];
                sd += red[2 * wv + 1];
            }
            re.partial[((size_t)shell * nsplit + split) * 2] = sn;
            re.partial[((size_t)shell * nsplit + split) * 2 + 1] = sd;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
bool sht_inverse_fuses_real_update(const mtip_ctx* c);

bool sht_reg_supported(const mtip_ctx* c) {
    int r1, r2;
    if (c->sht_mode < 2 || c->d_PT == nullptr || c->d_twN == nullptr || (c->nt & 1)) return false;
    if (!reg_radices(c->np, &r1, &r2)) return false;
    const int rpf = largest_even_divisor_le(c->nt, std::min(2 * SR_THREADS / r2, SR_THREADS / r1));
    const int rpi = largest_even_divisor_le(c->nt, std::min(SR_THREADS / r2, SR_THREADS / r1));
    if (rpf < 2 || rpi < 2) return false;
    const size_t lds_f = ((size_t)c->np + (size_t)rpf * r1 * (r2 + 1)) * sizeof(double2);
    const size_t lds_i = ((size_t)c->np + c->nlm + (size_t)rpi * c->nm + (size_t)rpi * r1 * (r2 + 1)) * sizeof(double2);
    return lds_f <= 160 * 1024 && lds_i <= 160 * 1024;
}

template <int PRE, int R1, int R2>
static void launch_fwd_r(mtip_ctx* c, const double2* grid, double2* coeff, int in_slot) {
    const int RP = largest_even_divisor_le(c->nt, std::min(2 * SR_THREADS / R2, SR_THREADS / R1));
    const size_t smem = ((size_t)c->np + (size_t)RP * R1 * (R2 + 1)) * sizeof(double2);
    const double norm = 2.0 * 3.14159265358979323846 / c->np;
    const int* sl = in_slot >= 0 ? c->d_slot : nullptr;
    const dim3 gr((unsigned)(c->B * c->N)), bl(SR_THREADS);
    const int per = div_up(c->npairs, SR_THREADS);
    // k_sht_fwd_pair: half the rows per pass, the table rows of a pass prefetched (one round trip per pass).  Measured at
    // 128 x L32 (hipEvent brackets, k_sht_fwd_reg / this kernel with one / with two shells per workgroup): 8 restarts per launch
    // 54.5 / 49.9 / 50.4 us, 3 restarts per launch 37.6 / 33.9 / 42.8 us -- one shell per workgroup it is
    const int th = RP / 4;
    const bool pair_geom = RP % 4 == 0 && (th == 2 || th == 4 || th == 8) && (c->nt / 2) % std::max(th, 1) == 0 && per <= 5;
    if (c->sht_fwd_pair && pair_geom) {
        const size_t smem_p = ((size_t)c->np + (size_t)(RP / 2) * R1 * (R2 + 1)) * sizeof(double2);
#define PAIR_ARGS grid, coeff, (const double*)c->d_PT, (const int*)c->d_lmtab, (const double2*)c->d_twN, (const double*)c->d_gw, \
                  c->nt, c->L, c->npairs, norm, sl, in_slot, c->B, c->N
#define PAIR_NS(MAXI, TH)                                                                                                        \
        if constexpr (TH * R2 <= SR_THREADS && 2 * TH * R1 <= SR_THREADS) {                                                      \
            hipLaunchKernelGGL((k_sht_fwd_pair<PRE, R1, R2, MAXI, TH, 1>), gr, bl, smem_p, c->stream, PAIR_ARGS);                 \
            return;                                                                                                              \
        }
#define PAIR_TH(MAXI)                                                                                                            \
        if (th == 8) { PAIR_NS(MAXI, 8) } else if (th == 4) { PAIR_NS(MAXI, 4) } else { PAIR_NS(MAXI, 2) }
        if (per <= 3) { PAIR_TH(3) } else { PAIR_TH(5) }
#undef PAIR_TH
#undef PAIR_NS
#undef PAIR_ARGS
    }
#define FWD_ARGS grid, coeff, (const double*)c->d_PT, (const int*)c->d_lmtab, (const double2*)c->d_twN, (const double*)c->d_gw, \
                 c->nt, c->L, c->npairs, RP, norm, sl, in_slot, c->B, c->N
    if (per <= 3) hipLaunchKernelGGL((k_sht_fwd_reg<PRE, R1, R2, 3>), gr, bl, smem, c->stream, FWD_ARGS);
    else if (per <= 5) hipLaunchKernelGGL((k_sht_fwd_reg<PRE, R1, R2, 5>), gr, bl, smem, c->stream, FWD_ARGS);
    else hipLaunchKernelGGL((k_sht_fwd_reg<PRE, R1, R2, 9>), gr, bl, smem, c->stream, FWD_ARGS);
#undef FWD_ARGS
}

template <int PRE>
static void launch_fwd_p(mtip_ctx* c, const double2* grid, double2* coeff, int in_slot) {
    switch (c->np) {
        case 16: launch_fwd_r<PRE, 4, 4>(c, grid, coeff, in_slot); break;
        case 32: launch_fwd_r<PRE, 4, 8>(c, grid, coeff, in_slot); break;
        case 64: launch_fwd_r<PRE, 8, 8>(c, grid, coeff, in_slot); break;
        case 128: launch_fwd_r<PRE, 8, 16>(c, grid, coeff, in_slot); break;
        default: launch_fwd_r<PRE, 16, 16>(c, grid, coeff, in_slot); break;
    }
}

void launch_sht_forward_reg(mtip_ctx* c, const double2* grid, double2* coeff, int prologue, int in_slot) {
    if (prologue == MTIP_PRE_SQUARE) launch_fwd_p<MTIP_PRE_SQUARE>(c, grid, coeff, in_slot);
    else if (prologue == MTIP_PRE_ABS) launch_fwd_p<MTIP_PRE_ABS>(c, grid, coeff, in_slot);
    else launch_fwd_p<MTIP_PRE_NONE>(c, grid, coeff, in_slot);
}

// LDS of the wide inverse kernel for `nsplit` workgroups per shell (see the kernel); *rp_out = rows per FFT pass
static size_t wide_lds_n(const mtip_ctx* c, int r1, int r2, int nsplit, int* rp_out) {
    if (c->nt % (2 * nsplit) != 0) return (size_t)1 << 40;
    const int ntl = c->nt / nsplit;
    if (nsplit > 1 && (ntl / 2) % 32 != 0) return (size_t)1 << 40;      // whole 32-theta chunks per workgroup
    const size_t fixed = (nsplit > 1 ? 0 : (size_t)c->np) + (size_t)ntl * c->nm + c->npairs;
    int rp = largest_even_divisor_le(ntl, std::min(SW_THREADS / r2, SW_THREADS / r1));
    // the transpose buffer aliases the coefficient block: shrink the pass until both fit
    while (rp >= 2 && (fixed + std::max((size_t)c->nlm, (size_t)rp * r1 * (r2 + 1))) * sizeof(double2) > 158 * 1024)
        rp = largest_even_divisor_le(ntl, rp - 2);
    if (rp_out) *rp_out = rp;
    if (rp < 2) return (size_t)1 << 40;
    return (fixed + std::max((size_t)c->nlm, (size_t)rp * r1 * (r2 + 1))) * sizeof(double2);
}

// smallest split (1, 2) whose working set fits one CU's LDS
static size_t wide_lds(const mtip_ctx* c, int r1, int r2, int* rp_out, int* nsplit_out = nullptr) {
    for (int ns = 1; ns <= 2; ++ns) {
        int rp = 0;
        const size_t lds = wide_lds_n(c, r1, r2, ns, &rp);
        if (lds <= 158 * 1024) {
            if (rp_out) *rp_out = rp;
            if (nsplit_out) *nsplit_out = ns;
            return lds;
        }
    }
    if (rp_out) *rp_out = 0;
    if (nsplit_out) *nsplit_out = 1;
    return (size_t)1 << 40;
}

template <int EPI, int R1, int R2>
static void launch_inv_r(mtip_ctx* c, const double2* coeff, double2* grid, const InvEpilogue& epi) {
    int rpw = 0, nsplit = 1;
    const size_t lds_w = wide_lds(c, R1, R2, &rpw, &nsplit);
    if (c->sht_wide && c->d_AB != nullptr && lds_w <= 158 * 1024) {
        const int* slw = (epi.out_slot >= 0 || EPI == EPI_REAL_UPDATE) ? c->d_slot : nullptr;
        hipLaunchKernelGGL((k_sht_inv_wide<EPI, R1, R2>), dim3((unsigned)(c->B * c->N * nsplit)), dim3(SW_THREADS), lds_w, c->stream,
                           coeff, grid, (const double*)c->d_P, (const int*)c->d_poff, (const double2*)c->d_AB,
                           (const double*)c->d_cost, c->npairs, (const double2*)c->d_twN, c->nt, c->L, rpw, c->N, epi.F, epi.shell_scale, slw, epi.out_slot, c->B, epi.coeff_sub, epi.real, nsplit);
        return;
    }
    const int RP = largest_even_divisor_le(c->nt, std::min(SR_THREADS / R2, SR_THREADS / R1));
    const size_t smem = ((size_t)c->np + c->nlm + (size_t)RP * c->nm + (size_t)RP * R1 * (R2 + 1)) * sizeof(double2);
    const int* sl = epi.out_slot >= 0 ? c->d_slot : nullptr;
    const dim3 gr((unsigned)(c->B * c->N)), bl(SR_THREADS);
    hipLaunchKernelGGL((k_sht_inv_reg<EPI, R1, R2>), gr, bl, smem, c->stream, coeff, grid, (const double*)c->d_PT,
                       (const int*)c->d_poff, (const double2*)c->d_twN, c->nt, c->L, c->npairs, RP, c->N, epi.F,
                       epi.shell_scale, sl, epi.out_slot, c->B);
}

template <int EPI>
static void launch_inv_p(mtip_ctx* c, const double2* coeff, double2* grid, const InvEpilogue& epi) {
    switch (c->np) {
        case 16: launch_inv_r<EPI, 4, 4>(c, coeff, grid, epi); break;
        case 32: launch_inv_r<EPI, 4, 8>(c, coeff, grid, epi); break;
        case 64: launch_inv_r<EPI, 8, 8>(c, coeff, grid, epi); break;
        case 128: launch_inv_r<EPI, 8, 16>(c, coeff, grid, epi); break;
        default: launch_inv_r<EPI, 16, 16>(c, coeff, grid, epi); break;
    }
}

void launch_sht_inverse_reg(mtip_ctx* c, const double2* coeff, double2* grid, const InvEpilogue& epi) {
    switch (epi.mode) {
        case EPI_MODULUS: launch_inv_p<EPI_MODULUS>(c, coeff, grid, epi); break;
        case EPI_SCALE_SHELL: launch_inv_p<EPI_SCALE_SHELL>(c, coeff, grid, epi); break;
        case EPI_REAL_UPDATE: launch_inv_p<EPI_REAL_UPDATE>(c, coeff, grid, epi); break;
        default: launch_inv_p<EPI_STORE>(c, coeff, grid, epi); break;
    }
}

// error partial sums the fused real-space epilogue writes per restart (one per workgroup)
int sht_inverse_real_update_blocks(const mtip_ctx* c) {
    int r1, r2, ns = 1;
    if (!reg_radices(c->np, &r1, &r2)) return c->N;
    wide_lds(c, r1, r2, nullptr, &ns);
    return c->N * ns;
}

// the fused real-space epilogue and the on-load coefficient difference exist in the wide inverse kernel only
bool sht_inverse_fuses_real_update(const mtip_ctx* c) {
    int r1, r2;
    if (!sht_reg_supported(c) || !c->sht_wide || c->d_AB == nullptr || !c->fuse_real_update) return false;
    if (!reg_radices(c->np, &r1, &r2)) return false;
    return wide_lds(c, r1, r2, nullptr) <= 158 * 1024;
}
