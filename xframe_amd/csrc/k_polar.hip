// Polar factor of the square matrices X_l = I_l^+ D^2 V_l (row a8 of SURVEY section 8):
//   approximate_unknowns  xframe/projects/fxs/projectLibrary/fxs_Projections.py:752-767   U_l = u @ vh of svd(PD_l @ I_l)
// by the scaled Newton iteration  Z <- (mu Z + (mu Z)^-H) / 2,  mu = sqrt(|Z^-1|_F / |Z|_F)  (Higham; backward stable,
// 8-10 iterations at the condition numbers 1e12 the graded X_l have, see scripts/polar_algorithms_study.py), one
// workgroup of 8 waves per matrix, all matrices of a batch in one launch.
//
// The inverse is an in-place Gauss-Jordan elimination with partial (row) pivoting, organised as a dataflow inside the
// workgroup instead of barrier-separated rounds:
//   * the matrix lives in registers: a wave owns a contiguous panel of column groups, a group = 4 columns x 16 S rows,
//     lane = (row & 15) + 16 (column & 3), S row slots per lane -- 80 x 68 register slots for the 65 x 65 matrix of
//     l = 32 (a lane-per-row layout would spend two full slots on 65 rows);
//   * step k is published by the wave that owns column k as one LDS record {column k, pivot row p, 1 / pivot}; every
//     wave consumes the records in order and applies the rank-1 update to its own columns, taking the pivot-row values of
//     its columns from its own registers (one ds_bpermute pair per group) -- a step needs no barrier, the only
//     synchronisation is the record flag (LDS operations of one wave complete in order, the flag is written last);
//   * while a wave owns the pivot columns (its panel) it produces record k + 1 as soon as it has applied record k to
//     its own columns; the other waves trail it by the latency of one record;
//   * the pivot search compares the high words of |a|^2 with the row index in the low mantissa bits, so that the
//     16-lane maximum is four v_max_u32 with DPP row operands.
// Pivot rows are not moved: the permutation is undone together with the conjugate transpose of the Newton update, in one
// scatter through LDS.
#include "mtip_internal.h"

#define PN_THREADS 512
#define PN_WAVES 8
#define PN_MAXIT 16
#define PN_MAXN 80                  // 5 row slots of 16
#define PN_FIXED 4096               // bytes in front of the record area: headers, permutation, reduction scratch
#define PN_TOL2 1e-14               // stop when |Z_new - Z|_F^2 < PN_TOL2 |Z_new|_F^2 (quadratic convergence: Z_new is then exact to ~1e-14)
#define PN_UNSCALED2 1e-4           // below this relative change the scaling is switched off (Higham's criterion)

struct __align__(16) PnHdr {
    int flag, p, pad0, pad1;
    double inv_re, inv_im;
};

__device__ __forceinline__ double pn_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = y * (2.0 - x * y);
    y = y * (2.0 - x * y);
    return y;
}

// maximum over the 16 lanes of a DPP row (every lane of the row gets it)
__device__ __forceinline__ unsigned pn_row16_max(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false));   // row_half_mirror
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false));   // row_mirror
    return v;
}

__device__ __forceinline__ double pn_readlane(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// sums of NV values over the workgroup, in a fixed order (deterministic); every thread gets the totals
template <int NV>
__device__ __forceinline__ void pn_block_sum(double (&v)[NV], double* red) {
#pragma unroll
    for (int i = 0; i < NV; ++i)
        for (int o = 32; o > 0; o >>= 1) v[i] += __shfl_xor(v[i], o);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0)
        for (int i = 0; i < NV; ++i) red[wave * NV + i] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double t = 0.0;
        for (int w = 0; w < PN_WAVES; ++w) t += red[w * NV + i];
        v[i] = t;
    }
    __syncthreads();
}

// S: row slots per lane (n <= 16 S), GM: column groups per wave (ceil(ceil(n / 4) / 8) <= GM)
template <int S, int GM>
__device__ void pn_body(const double2* __restrict__ X, double2* __restrict__ U, int n, unsigned char* smem, int* diag) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lc = lane >> 4;
    PnHdr* hdr = reinterpret_cast<PnHdr*>(smem);                          // [PN_MAXN]
    int* prow = reinterpret_cast<int*>(smem + PN_MAXN * sizeof(PnHdr));   // [PN_MAXN] pivot row of step k
    int* rowstep = prow + PN_MAXN;                                        // [PN_MAXN] step at which row i was the pivot row
    double* red = reinterpret_cast<double*>(rowstep + PN_MAXN);           // [PN_WAVES * 4]
    double2* recs = reinterpret_cast<double2*>(smem + PN_FIXED);          // [n][16 S] column k as published at step k
    double2* stage = recs;                                                // [n][ld] aliases the records between two inversions
    const int ld = (n < 16 * S) ? n + 1 : n;
    const int NG = (n + 3) >> 2, base = NG / PN_WAVES, rem = NG % PN_WAVES;
    const int ng = base + (wave < rem ? 1 : 0);                           // column groups of this wave
    const int g0 = wave * base + min(wave, rem);                          // first of them
    for (int k = tid; k < PN_MAXN; k += PN_THREADS) hdr[k].flag = 0;

    double2 A[GM][S], Z[GM][S];
    double nrm[1] = {0.0};
#pragma unroll
    for (int g = 0; g < GM; ++g) {
        const int j = 4 * (g0 + g) + lc;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int i = 16 * s + lr;
            const bool ok = (g < ng) && (i < n) && (j < n);
            Z[g][s] = ok ? X[(size_t)j * n + i] : make_double2(0.0, 0.0);
            nrm[0] += cabs2(Z[g][s]);
        }
    }
    pn_block_sum<1>(nrm, red);                                            // (also orders the flag reset before the first record)
    double zn2 = nrm[0];
    if (!(zn2 > 0.0) || !(zn2 < __builtin_huge_val())) {                  // X = 0 (or not finite): U = 0, like 1/sigma -> 0 of the SVD route
#pragma unroll
        for (int g = 0; g < GM; ++g) {
            const int j = 4 * (g0 + g) + lc;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int i = 16 * s + lr;
                if ((g < ng) && (i < n) && (j < n)) U[(size_t)j * n + i] = make_double2(0.0, 0.0);
            }
        }
        if (tid == 0) *diag = (n << 8);
        return;
    }
    {
        const double sc = 1.0 / sqrt(zn2);                                // the polar factor does not depend on the scale of X
#pragma unroll
        for (int g = 0; g < GM; ++g)
#pragma unroll
            for (int s = 0; s < S; ++s) Z[g][s] = cscale(Z[g][s], sc);
        zn2 = 1.0;
    }

    bool unscaled = false;
    int it = 1;
    for (; it <= PN_MAXIT; ++it) {
        // ---------------------------------------------------------------- A <- Z^-1 in place (rows stay where they are)
#pragma unroll
        for (int g = 0; g < GM; ++g)
#pragma unroll
            for (int s = 0; s < S; ++s) A[g][s] = Z[g][s];
        unsigned rowdone = 0;                                              // bit s: row 16 s + lr has been a pivot row
        const double floor2 = 1e-40 * zn2;                                 // pivots below 1e-20 |Z|_F count as that (singular X_l)
        if (ng > 0) {
            for (int k = 0; k < n; ++k) {
                const int gk = k >> 2, cr = k & 3;
                const int ow = (gk < rem * (base + 1)) ? gk / (base + 1) : rem + (gk - rem * (base + 1)) / max(base, 1);
                const int gl = gk - g0;                                    // local group of column k (owner only)
                if (ow == wave) {
                    // ---- produce record k: pivot search in column k (rows not used yet), publish column and pivot
                    unsigned key = 0;
#pragma unroll
                    for (int g = 0; g < GM; ++g) {
                        if (g == gl) {
#pragma unroll
                            for (int s = 0; s < S; ++s) {
                                const int i = 16 * s + lr;
                                const double m2 = fma(A[g][s].x, A[g][s].x, A[g][s].y * A[g][s].y);
                                const unsigned ks = 0x80000000u | (((unsigned)__double2hiint(m2) >> 1) & 0x3FFFFF80u) | (unsigned)i;
                                const bool ok = (lc == cr) && (i < n) && !((rowdone >> s) & 1u);
                                key = max(key, ok ? ks : 0u);
                            }
                        }
                    }
                    key = pn_row16_max(key);
                    const int p = __builtin_amdgcn_readlane((int)key, 16 * cr) & 127;
                    const int sp = p >> 4, lp = p & 15;
                    double2 cand = make_double2(0.0, 0.0);
#pragma unroll
                    for (int g = 0; g < GM; ++g)
                        if (g == gl) {
#pragma unroll
                            for (int s = 0; s < S; ++s)
                                if (s == sp) cand = A[g][s];
                        }
                    double pr = pn_readlane(cand.x, 16 * cr + lp), pi = pn_readlane(cand.y, 16 * cr + lp);
                    double d = fma(pr, pr, pi * pi);
                    if (!(d >= floor2)) {
                        pr = sqrt(floor2);
                        pi = 0.0;
                        d = floor2;
                    }
                    const double rd = pn_rcp(d);
                    const double2 inv = make_double2(pr * rd, -pi * rd);
                    // the record carries piv - 1 on the pivot row: the generic update a_pj - (piv - 1) a_pj / piv = a_pj / piv
                    // then scales that row without a row select in the consumers
#pragma unroll
                    for (int g = 0; g < GM; ++g) {
                        if (g == gl && lc == cr) {
#pragma unroll
                            for (int s = 0; s < S; ++s) {
                                double2 v = A[g][s];
                                if (s == sp && lr == lp) v = make_double2(pr - 1.0, pi);
                                recs[(size_t)k * (16 * S) + 16 * s + lr] = v;
                                A[g][s] = make_double2(0.0, 0.0);          // column k of the inverse starts from zero
                            }
                        }
                    }
                    if (lane == 0) {
                        hdr[k].p = p;
                        hdr[k].inv_re = inv.x;
                        hdr[k].inv_im = inv.y;
                        prow[k] = p;
                        rowstep[p] = k;
                        __hip_atomic_store(&hdr[k].flag, it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                // ---- consume record k: rank-1 update of the own columns
                while (__hip_atomic_load(&hdr[k].flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != it) __builtin_amdgcn_s_sleep(1);
                const int p = __builtin_amdgcn_readfirstlane(hdr[k].p);
                const double2 inv = make_double2(hdr[k].inv_re, hdr[k].inv_im);
                const int sp = p >> 4, lp = p & 15;
                double2 f[S];
#pragma unroll
                for (int s = 0; s < S; ++s) f[s] = recs[(size_t)k * (16 * S) + 16 * s + lr];
                const int srcl = (lane & 48) | lp;
#pragma unroll
                for (int g = 0; g < GM; ++g) {
                    if (g < ng) {
                        double2 raw = A[g][0];
#pragma unroll
                        for (int s = 1; s < S; ++s)
                            if (s == sp) raw = A[g][s];
                        raw.x = __shfl(raw.x, srcl);
                        raw.y = __shfl(raw.y, srcl);
                        double2 r = cmul(raw, inv);
                        const bool mine = (ow == wave) && (g == gl) && (lc == cr);   // column k itself: a_ik <- -f_i / piv
                        if (mine) r = inv;
#pragma unroll
                        for (int s = 0; s < S; ++s) {
                            A[g][s].x = fma(-f[s].x, r.x, fma(f[s].y, r.y, A[g][s].x));
                            A[g][s].y = fma(-f[s].x, r.y, fma(-f[s].y, r.x, A[g][s].y));
                            if (mine && s == sp && lr == lp) A[g][s] = inv;   // ... and a_pk <- 1 / piv exactly
                        }
                    }
                }
                if (lr == lp) rowdone |= 1u << sp;
            }
        }
        __syncthreads();                                                   // every record consumed: the record area is free
        // ---------------------------------------------------------------- scaling mu = sqrt(|Z^-1|_F / |Z|_F)
        double yn2[1] = {0.0};
#pragma unroll
        for (int g = 0; g < GM; ++g)
#pragma unroll
            for (int s = 0; s < S; ++s) yn2[0] += cabs2(A[g][s]);
        pn_block_sum<1>(yn2, red);
        const double mu = unscaled ? 1.0 : sqrt(sqrt(yn2[0] / zn2));
        // ---------------------------------------------------------------- Z <- (mu Z + (mu Z)^-H) / 2
        // a_ij (row i was the pivot row of step rowstep[i], column j had pivot row prow[j]) is element (rowstep[i], prow[j]) of
        // Z^-1, i.e. conj(a_ij) is element (prow[j], rowstep[i]) of Z^-H
        const double c1 = 0.5 / mu, c0 = 0.5 * mu;
#pragma unroll
        for (int g = 0; g < GM; ++g) {
            const int j = 4 * (g0 + g) + lc;
            const int pj = (g < ng && j < n) ? prow[j] : 0;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int i = 16 * s + lr;
                if ((g < ng) && (i < n) && (j < n)) stage[(size_t)pj * ld + rowstep[i]] = make_double2(c1 * A[g][s].x, -c1 * A[g][s].y);
            }
        }
        __syncthreads();
        double sums[2] = {0.0, 0.0};
#pragma unroll
        for (int g = 0; g < GM; ++g) {
            const int j = 4 * (g0 + g) + lc;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int i = 16 * s + lr;
                if ((g < ng) && (i < n) && (j < n)) {
                    const double2 t = stage[(size_t)i * ld + j];
                    const double2 zn = make_double2(fma(c0, Z[g][s].x, t.x), fma(c0, Z[g][s].y, t.y));
                    sums[0] += cabs2(csub(zn, Z[g][s]));
                    sums[1] += cabs2(zn);
                    Z[g][s] = zn;
                }
            }
        }
        pn_block_sum<2>(sums, red);                                        // (its barriers also free the staging area for the records)
        zn2 = sums[1];
        if (sums[0] < PN_UNSCALED2 * sums[1]) unscaled = true;
        if (!(sums[0] >= PN_TOL2 * sums[1])) break;                        // converged (or not finite: stop)
    }
    // U_l = polar(X_l)^+ in the row-major (k_l, 2l+1) layout of mtip_get_unknowns = conj of the column-major polar factor
#pragma unroll
    for (int g = 0; g < GM; ++g) {
        const int j = 4 * (g0 + g) + lc;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int i = 16 * s + lr;
            if ((g < ng) && (i < n) && (j < n)) U[(size_t)j * n + i] = make_double2(Z[g][s].x, -Z[g][s].y);
        }
    }
    if (tid == 0) *diag = min(it, PN_MAXIT) | (n << 8);
}

__global__ void __launch_bounds__(PN_THREADS) k_polar_newton(const double2* __restrict__ Xall, double2* __restrict__ Uall,
                                                             const int* __restrict__ active, const int* __restrict__ xoff,
                                                             int xtot, int L, const int* __restrict__ jorder,
                                                             int* __restrict__ diag) {
    HIP_DYNAMIC_SHARED(unsigned char, pn_smem)
    const int b = blockIdx.x, l = jorder[blockIdx.y];
    if (!active[l]) return;
    const int n = 2 * l + 1;
    const double2* X = Xall + (size_t)b * xtot + xoff[l];
    double2* U = Uall + (size_t)b * xtot + xoff[l];
    int* dg = diag + (size_t)b * (L + 1) + l;
    if (n == 1) {                                                          // l = 0: a phase
        if (threadIdx.x == 0) {
            const double2 x = X[0];
            const double a2 = cabs2(x);
            const double ia = (a2 > 0.0 && a2 < __builtin_huge_val()) ? 1.0 / sqrt(a2) : 0.0;
            U[0] = make_double2(x.x * ia, -x.y * ia);
            *dg = 1 | (1 << 8);
        }
        return;
    }
    switch ((n + 15) >> 4) {
        case 1: pn_body<1, 1>(X, U, n, pn_smem, dg); break;
        case 2: pn_body<2, 1>(X, U, n, pn_smem, dg); break;
        case 3: pn_body<3, 2>(X, U, n, pn_smem, dg); break;
        case 4: pn_body<4, 2>(X, U, n, pn_smem, dg); break;
        default: pn_body<5, 3>(X, U, n, pn_smem, dg); break;
    }
}

bool polar_newton_supported(const mtip_ctx* c) {
    if (!c->polar_newton) return false;
    bool any = false;
    for (int l = 0; l <= c->L; ++l) {
        if (!c->active[l]) continue;
        any = true;
        if (c->kl[l] != 2 * l + 1 || 2 * l + 1 > PN_MAXN) return false;    // square X_l up to 80 x 80 (L <= 39)
    }
    return any;
}

// X_l (column-major (2l+1) x k_l, as k_proj_mfma<PG_X> leaves it in c->d_X) -> U_l in c->d_U
int launch_polar_newton(mtip_ctx* c) {
    int nmax = 1;
    for (int l = 0; l <= c->L; ++l)
        if (c->active[l]) nmax = std::max(nmax, 2 * l + 1);
    if (c->d_jorder == nullptr) {                                          // active orders, heaviest first
        std::vector<int> ord;
        for (int l = 0; l <= c->L; ++l)
            if (c->active[l]) ord.push_back(l);
        std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return c->kl[x] * (2 * x + 1) > c->kl[y] * (2 * y + 1); });
        c->n_jorder = (int)ord.size();
        if (ord.empty()) ord.push_back(0);
        if (hipMalloc((void**)&c->d_jorder, ord.size() * sizeof(int)) != hipSuccess) {
            c->err = "polar factor order list: out of device memory";
            return MTIP_ENOMEM;
        }
        (void)hipMemcpy(c->d_jorder, ord.data(), ord.size() * sizeof(int), hipMemcpyHostToDevice);
    }
    const int S = (nmax + 15) / 16;
    const size_t lds = PN_FIXED + (size_t)nmax * 16 * S * sizeof(double2);
    hipLaunchKernelGGL(k_polar_newton, dim3((unsigned)c->B, (unsigned)std::max(c->n_jorder, 1)), dim3(PN_THREADS), lds, c->stream,
                       (const double2*)c->d_X, c->d_U, (const int*)c->d_active, (const int*)c->d_xoff, c->xtot, c->L,
                       (const int*)c->d_jorder, c->d_sweeps);
    return MTIP_OK;
}
