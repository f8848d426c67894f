// Polar factor of the square matrices X_l = I_l^+ D^2 V_l (row a8 of SURVEY section 8):
//   approximate_unknowns  xframe/projects/fxs/projectLibrary/fxs_Projections.py:752-767   U_l = u @ vh of svd(PD_l @ I_l)
// by the scaled Newton iteration  Z <- (mu Z + (mu Z)^-H) / 2,  mu = sqrt(|Z^-1|_F / |Z|_F)  (Higham; backward stable,
// 8-10 iterations at the condition numbers 1e12 the graded X_l have, see scripts/polar_algorithms_study.py), one
// workgroup of 8 waves per matrix, all matrices of a batch in one launch.
//
// The inverse is an in-place Gauss-Jordan elimination with partial (row) pivoting, organised as a dataflow inside the
// workgroup instead of barrier-separated rounds:
//   * the matrix lives in registers, lane = row (S = 1 or 2 row slots of 64), a wave owns a contiguous panel of columns;
//   * step k is published by the wave that owns column k as one LDS record {pivot row p, f' = column k / pivot}; every
//     wave consumes the records in order and applies the rank-1 update a_ij -= f'_i a_pj to its own columns: f' is lane
//     aligned (one ds_read_b128 per slot), a_pj is a v_readlane of the wave's own registers -- no LDS round trip, no
//     barrier; the only synchronisation is the record flag (LDS operations of one wave complete in order, the flag is
//     written last);
//   * while a wave owns the pivot columns (its panel) nothing on its critical path touches LDS: pivot search, 1 / pivot,
//     f', update of its own columns from registers, next pivot search; the other waves trail it by one record;
//   * the record carries f'_p = 1 - 1/pivot on the pivot row, so the same update scales that row (no row select in the
//     consumers); column k of the inverse (-f', 1/pivot at row p) is set by its owner;
//   * the pivot search compares the high words of |a|^2 with the row index in the low mantissa bits: four v_max_u32
//     with DPP row operands, four v_readlane and three s_max_u32.
// Pivot rows are not moved: the permutation is undone together with the conjugate transpose of the Newton update, in one
// scatter through LDS.
#include "mtip_internal.h"

#define PN_THREADS 512
#define PN_WAVES 8
#define PN_MAXIT 16
#define PN_MAXN 72                  // 9 columns per wave, 2 row slots
#define PN_FIXED 2048               // bytes in front of the record area: headers, permutation, reduction scratch
#define PN_TOL2 1e-14               // stop when |Z_new - Z|_F^2 < PN_TOL2 |Z_new|_F^2 (quadratic convergence: Z_new is then exact to ~1e-14)
#define PN_UNSCALED2 1e-4           // below this relative change the scaling is switched off (Higham's criterion)

struct PnHdr {
    int flag, p;
};

__device__ __forceinline__ double pn_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = y * (2.0 - x * y);
    y = y * (2.0 - x * y);
    return y;
}

// maximum over the wave: DPP butterflies inside the rows of 16, then the four rows through the scalar unit (uniform result)
__device__ __forceinline__ unsigned pn_wave_max(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false));   // row_half_mirror
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false));   // row_mirror
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    return max(max(a, b), max(c, d));
}

__device__ __forceinline__ double2 pn_readlane2(double2 v, int src) {
    const int xl = __builtin_amdgcn_readlane(__double2loint(v.x), src), xh = __builtin_amdgcn_readlane(__double2hiint(v.x), src);
    const int yl = __builtin_amdgcn_readlane(__double2loint(v.y), src), yh = __builtin_amdgcn_readlane(__double2hiint(v.y), src);
    return make_double2(__hiloint2double(xh, xl), __hiloint2double(yh, yl));
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

// a_ij -= f'_i a_pj for all CM column registers of this wave, branch free (registers past the wave's last column hold
// zeros and stay zero; the owner of column k overwrites that column afterwards).  SP: slot of the pivot row (a
// compile-time constant inside a uniform branch).  The pivot-row values are fetched for all columns first so that their
// latency overlaps; BPERM: through ds_bpermute (LDS crossbar) instead of v_readlane.
template <int S, int CM, int SP, bool BPERM>
__device__ __forceinline__ void pn_apply(double2 (&A)[CM][S], const double2 (&fp)[S], int lp) {
    double2 raw[CM];
#pragma unroll
    for (int c = 0; c < CM; ++c) {
        if (BPERM) {
            raw[c].x = __shfl(A[c][SP].x, lp);
            raw[c].y = __shfl(A[c][SP].y, lp);
        } else {
            raw[c] = pn_readlane2(A[c][SP], lp);
        }
    }
#pragma unroll
    for (int c = 0; c < CM; ++c) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            A[c][s].x = fma(-fp[s].x, raw[c].x, fma(fp[s].y, raw[c].y, A[c][s].x));
            A[c][s].y = fma(-fp[s].x, raw[c].y, fma(-fp[s].y, raw[c].x, A[c][s].y));
        }
    }
}

// S: row slots of 64 per lane (n <= 64 S), CM: columns per wave (ceil(n / 8) <= CM)
template <int S, int CM>
__device__ void pn_body(const double2* __restrict__ X, double2* __restrict__ U, int n, unsigned char* smem, int* diag, long long* dbg,
                        int variant) {
    // the wave index through readfirstlane: the compiler then knows that everything derived from it (ownership of a
    // column, the column range) is wave-uniform and emits scalar branches instead of exec masks and per-lane selects
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    PnHdr* hdr = reinterpret_cast<PnHdr*>(smem);                          // [PN_MAXN]
    int* prow = reinterpret_cast<int*>(smem + PN_MAXN * sizeof(PnHdr));   // [PN_MAXN] pivot row of step k
    int* rowstep = prow + PN_MAXN;                                        // [PN_MAXN] step at which row i was the pivot row
    double* red = reinterpret_cast<double*>(rowstep + PN_MAXN);           // [PN_WAVES * 2]
    double2* recs = reinterpret_cast<double2*>(smem + PN_FIXED);          // [n][n] f' of step k
    double2* stage = recs;                                                // [n][n] aliases the records between two inversions
    double2* zs = recs + (size_t)n * n;                                   // [n][n] the iterate Z, column-major (registers hold the matrix being inverted)
    const int base = n / PN_WAVES, rem = n % PN_WAVES;
    const int nc = base + (wave < rem ? 1 : 0);                           // columns of this wave
    const int c0 = wave * base + min(wave, rem);                          // first of them
    for (int k = tid; k < PN_MAXN; k += PN_THREADS) hdr[k].flag = 0;

    double2 A[CM][S];
    double nrm[1] = {0.0};
#pragma unroll
    for (int c = 0; c < CM; ++c) {
        const int j = c0 + c;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int i = 64 * s + lane;
            const bool ok = (c < nc) && (i < n);
            A[c][s] = ok ? X[(size_t)j * n + i] : make_double2(0.0, 0.0);
            nrm[0] += cabs2(A[c][s]);
        }
    }
    pn_block_sum<1>(nrm, red);                                            // (also orders the flag reset before the first record)
    double zn2 = nrm[0];
    if (!(zn2 > 0.0) || !(zn2 < __builtin_huge_val())) {                  // X = 0 (or not finite): U = 0, like 1/sigma -> 0 of the SVD route
#pragma unroll
        for (int c = 0; c < CM; ++c) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int i = 64 * s + lane;
                if ((c < nc) && (i < n)) U[(size_t)(c0 + c) * n + i] = make_double2(0.0, 0.0);
            }
        }
        if (tid == 0) *diag = (n << 8);
        return;
    }
    {
        const double sc = 1.0 / sqrt(zn2);                                // the polar factor does not depend on the scale of X
#pragma unroll
        for (int c = 0; c < CM; ++c)
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int i = 64 * s + lane;
                A[c][s] = cscale(A[c][s], sc);
                if ((c < nc) && (i < n)) zs[(size_t)(c0 + c) * n + i] = A[c][s];     // own elements only: no barrier needed
            }
        zn2 = 1.0;
    }

    // diagnostic phase timers (dbg != nullptr only through mtip_debug_polar_timing): cycles of this wave spent producing
    // records, waiting for records, applying them, and in the whole kernel
    long long t_prod = 0, t_wait = 0, t_cons = 0, t_all = dbg ? clock64() : 0;
    bool unscaled = false;
    int it = 1;
    for (; it <= PN_MAXIT; ++it) {
        // ---------------------------------------------------------------- A (= Z) <- Z^-1 in place (rows stay where they are)
        unsigned rowdone = 0;                                              // bit s: row 64 s + lane has been a pivot row
        const double floor2 = 1e-40 * zn2;                                 // pivots below 1e-20 |Z|_F count as that (singular X_l)
        if (nc > 0) {
            for (int k = 0; k < n; ++k) {
                const int ow = (k < rem * (base + 1)) ? k / (base + 1) : rem + (k - rem * (base + 1)) / max(base, 1);
                const bool own = (ow == wave);
                const int ck = k - c0;                                     // local index of column k (owner only)
                double2 fp[S], col[S];
                int p;
                long long t0 = dbg ? clock64() : 0, t1 = 0, t2 = 0;
#pragma unroll
                for (int s = 0; s < S; ++s) col[s] = make_double2(0.0, 0.0);
                if (own) {
                    // ---- produce record k: pivot search in column k (rows not used yet), f' = column / pivot
#pragma unroll
                    for (int c = 0; c < CM; ++c)
                        if (c == ck) {
#pragma unroll
                            for (int s = 0; s < S; ++s) col[s] = A[c][s];
                        }
                    unsigned key = 0;
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const int i = 64 * s + lane;
                        const double m2 = fma(col[s].x, col[s].x, col[s].y * col[s].y);
                        const unsigned ks = 0x80000000u | (((unsigned)__double2hiint(m2) >> 1) & 0x3FFFFF80u) | (unsigned)i;
                        const bool ok = (i < n) && !((rowdone >> s) & 1u);
                        key = max(key, ok ? ks : 0u);
                    }
                    p = (int)(pn_wave_max(key) & 127u);
                    const int lp = p & 63;
                    double2 piv = pn_readlane2(col[0], lp);
                    if (S > 1 && p >= 64) piv = pn_readlane2(col[S - 1], lp);
                    double d = fma(piv.x, piv.x, piv.y * piv.y);
                    if (!(d >= floor2)) {
                        piv = make_double2(sqrt(floor2), 0.0);
                        d = floor2;
                    }
                    const double rd = pn_rcp(d);
                    const double2 inv = make_double2(piv.x * rd, -piv.y * rd);
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const int i = 64 * s + lane;
                        fp[s] = cmul(col[s], inv);
                        if (i == p) fp[s] = make_double2(1.0 - inv.x, -inv.y);   // a_pj - (1 - 1/piv) a_pj = a_pj / piv
                        if (i < n) recs[(size_t)k * n + i] = fp[s];
                        col[s] = (i == p) ? inv : make_double2(-fp[s].x, -fp[s].y);   // column k of the inverse
                        if (i >= n) col[s] = make_double2(0.0, 0.0);
                    }
                    if (lane == 0) {
                        hdr[k].p = p;
                        prow[k] = p;
                        rowstep[p] = k;
                        __hip_atomic_store(&hdr[k].flag, it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    t1 = t2 = dbg ? clock64() : 0;
                } else {
                    // ---- wait for record k
                    t1 = dbg ? clock64() : 0;
                    if (variant & 2) {
                        while (__hip_atomic_load(&hdr[k].flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != it) {}
                    } else {
                        while (__hip_atomic_load(&hdr[k].flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != it) __builtin_amdgcn_s_sleep(1);
                    }
                    t2 = dbg ? clock64() : 0;
                    p = __builtin_amdgcn_readfirstlane(hdr[k].p);
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const int i = 64 * s + lane;
                        fp[s] = recs[(size_t)k * n + min(i, n - 1)];
                        if (i >= n) fp[s] = make_double2(0.0, 0.0);
                    }
                }
                // ---- rank-1 update of the own columns; the owner then installs column k of the inverse
                const int lp = p & 63;
                if (variant & 1) {
                    if (S > 1 && p >= 64) pn_apply<S, CM, S - 1, true>(A, fp, lp);
                    else pn_apply<S, CM, 0, true>(A, fp, lp);
                } else {
                    if (S > 1 && p >= 64) pn_apply<S, CM, S - 1, false>(A, fp, lp);
                    else pn_apply<S, CM, 0, false>(A, fp, lp);
                }
                if (own) {
#pragma unroll
                    for (int c = 0; c < CM; ++c)
                        if (c == ck) {
#pragma unroll
                            for (int s = 0; s < S; ++s) A[c][s] = col[s];
                        }
                }
                if (lane == lp) rowdone |= 1u << (p >> 6);
                if (dbg) {
                    const long long t3 = clock64();
                    t_prod += t1 - t0;
                    t_wait += t2 - t1;
                    t_cons += t3 - t2;
                }
            }
        }
        __syncthreads();                                                   // every record consumed: the record area is free
        // ---------------------------------------------------------------- scaling mu = sqrt(|Z^-1|_F / |Z|_F)
        double yn2[1] = {0.0};
#pragma unroll
        for (int c = 0; c < CM; ++c)
#pragma unroll
            for (int s = 0; s < S; ++s) yn2[0] += cabs2(A[c][s]);
        pn_block_sum<1>(yn2, red);
        const double mu = unscaled ? 1.0 : sqrt(sqrt(yn2[0] / zn2));
        // ---------------------------------------------------------------- Z <- (mu Z + (mu Z)^-H) / 2
        // a_ij (row i was the pivot row of step rowstep[i], column j had pivot row prow[j]) is element (rowstep[i], prow[j]) of
        // Z^-1, i.e. conj(a_ij) is element (prow[j], rowstep[i]) of Z^-H
        const double c1 = 0.5 / mu, cz = 0.5 * mu;
#pragma unroll
        for (int c = 0; c < CM; ++c) {
            const int pj = (c < nc) ? prow[c0 + c] : 0;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int i = 64 * s + lane;
                if ((c < nc) && (i < n)) stage[(size_t)pj * n + rowstep[i]] = make_double2(c1 * A[c][s].x, -c1 * A[c][s].y);
            }
        }
        __syncthreads();
        double sums[2] = {0.0, 0.0};
#pragma unroll
        for (int c = 0; c < CM; ++c) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int i = 64 * s + lane;
                A[c][s] = make_double2(0.0, 0.0);
                if ((c < nc) && (i < n)) {
                    const double2 t = stage[(size_t)i * n + (c0 + c)];
                    const double2 zo = zs[(size_t)(c0 + c) * n + i];
                    const double2 zn = make_double2(fma(cz, zo.x, t.x), fma(cz, zo.y, t.y));
                    sums[0] += cabs2(csub(zn, zo));
                    sums[1] += cabs2(zn);
                    zs[(size_t)(c0 + c) * n + i] = zn;
                    A[c][s] = zn;
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
    for (int c = 0; c < CM; ++c) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int i = 64 * s + lane;
            if ((c < nc) && (i < n)) U[(size_t)(c0 + c) * n + i] = make_double2(A[c][s].x, -A[c][s].y);
        }
    }
    if (tid == 0) *diag = min(it, PN_MAXIT) | (n << 8);
    if (dbg && lane == 0) {
        dbg[wave * 4 + 0] = t_prod;
        dbg[wave * 4 + 1] = t_wait;
        dbg[wave * 4 + 2] = t_cons;
        dbg[wave * 4 + 3] = clock64() - t_all;
    }
}

__global__ void __launch_bounds__(PN_THREADS) k_polar_newton(const double2* __restrict__ Xall, double2* __restrict__ Uall,
                                                             const int* __restrict__ active, const int* __restrict__ xoff,
                                                             int xtot, int L, const int* __restrict__ jorder,
                                                             int* __restrict__ diag, long long* __restrict__ dbg_all, int variant) {
    HIP_DYNAMIC_SHARED(unsigned char, pn_smem)
    const int b = blockIdx.x, l = jorder[blockIdx.y];
    if (!active[l]) return;
    const int n = 2 * l + 1;
    const double2* X = Xall + (size_t)b * xtot + xoff[l];
    double2* U = Uall + (size_t)b * xtot + xoff[l];
    int* dg = diag + (size_t)b * (L + 1) + l;
    long long* dbg = dbg_all ? dbg_all + ((size_t)b * (L + 1) + l) * (PN_WAVES * 4) : nullptr;
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
    if (n <= 16) pn_body<1, 2>(X, U, n, pn_smem, dg, dbg, variant);
    else if (n <= 32) pn_body<1, 4>(X, U, n, pn_smem, dg, dbg, variant);
    else if (n <= 64) pn_body<1, 8>(X, U, n, pn_smem, dg, dbg, variant);
    else pn_body<2, 9>(X, U, n, pn_smem, dg, dbg, variant);
}

bool polar_newton_supported(const mtip_ctx* c) {
    if (!c->polar_newton) return false;
    bool any = false;
    for (int l = 0; l <= c->L; ++l) {
        if (!c->active[l]) continue;
        any = true;
        if (c->kl[l] != 2 * l + 1 || 2 * l + 1 > PN_MAXN) return false;    // square X_l up to 72 x 72 (L <= 35)
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
        (void)mtip_copy(c, c->d_jorder, ord.data(), ord.size() * sizeof(int), hipMemcpyHostToDevice);
    }
    const size_t lds = PN_FIXED + 2 * (size_t)nmax * nmax * sizeof(double2);      // records / staging + the iterate Z
    hipLaunchKernelGGL(k_polar_newton, dim3((unsigned)c->B, (unsigned)std::max(c->n_jorder, 1)), dim3(PN_THREADS), lds, c->stream,
                       (const double2*)c->d_X, c->d_U, (const int*)c->d_active, (const int*)c->d_xoff, c->xtot, c->L,
                       (const int*)c->d_jorder, c->d_sweeps, c->d_polar_dbg, c->polar_variant);
    return MTIP_OK;
}
