// Reciprocal-space (B_l / V_l) projection, rows a8, a9, a18, a19 of SURVEY section 8.
//   approximate_unknowns  xframe/projects/fxs/projectLibrary/fxs_Projections.py:752-767
//        U_l = polar unitary factor of A_l = V_l^+ D^2 I_l   (reference: u @ vh of numpy svd)
//   mtip_projection       fxs_Projections.py:832-849, 866-871
//        I'_l[mask_l] = (V_l U_l)[mask_l];  I'_0[mask_0] = V_0[mask_0];  I'_0 /= sqrt(N_particles)
//   B_l = I_l I_l^+       fxs_invariant_tools.py:915-923;  metric fxs_IO_methods.py:408-447
//
// The polar factor is computed with a one-sided (Hestenes) Jacobi SVD on X_l = A_l^+ (n_l x k_l,
// k_l <= n_l, stored column-major so that column operations are contiguous):
//   X V_r = W (orthogonal columns), sigma_c = |W_c|,  polar(X) = W Sigma^-1 V_r^+,  U_l = polar(X)^+.
#include "mtip_internal.h"
#include "k_jacobi.h"

#define JAC_TG 8            // threads cooperating on one column pair

// tournament pairing of round r: players 0..Cp-1 (Cp even), pair index pi in [0, Cp/2)
__device__ __forceinline__ void jacobi_pair(int r, int pi, int Cp, int* a, int* b) {
    const int M = Cp - 1;                 // 0 <= r < M, 0 <= pi <= M/2: one conditional subtract replaces the modulo
    if (pi == 0) {
        *a = r;
        *b = M;
    } else {
        int x = r + pi;
        if (x >= M) x -= M;
        int y = r - pi + M;
        if (y >= M) y -= M;
        *a = x;
        *b = y;
    }
}

__global__ void __launch_bounds__(256) k_polar_jacobi(double2* __restrict__ Xall, double2* __restrict__ Vrall,
                                                      double2* __restrict__ Uall, const int* __restrict__ kl,
                                                      const int* __restrict__ used, const int* __restrict__ xoff,
                                                      const int* __restrict__ roff, int xtot, int rtot) {
    __shared__ int s_rotated;
    __shared__ double s_isig[128];                        // k_l <= 2*63+1
    const int l = blockIdx.x, b = blockIdx.y;
    if (!used[l]) return;                                  // uniform per block
    const int k = kl[l], n = 2 * l + 1;
    double2* X = Xall + (size_t)b * xtot + xoff[l];
    double2* Vr = Vrall + (size_t)b * rtot + roff[l];
    double2* U = Uall + (size_t)b * xtot + xoff[l];
    const int tid = threadIdx.x;
    for (int e = tid; e < k * k; e += blockDim.x) {
        const int cc = e / k, i = e - cc * k;
        Vr[e] = make_double2(cc == i ? 1.0 : 0.0, 0.0);
    }
    __syncthreads();
    const int Cp = k + (k & 1);
    const int rounds = Cp - 1;
    const int pairs = Cp / 2;
    const int ngroups = blockDim.x / JAC_TG;
    const int group = tid / JAC_TG, t = tid - group * JAC_TG;
    const int per_group = (pairs + ngroups - 1) / ngroups;
    if (k > 1) {
        for (int sweep = 0; sweep < JAC_MAX_SWEEPS; ++sweep) {
            if (tid == 0) s_rotated = 0;
            __syncthreads();
            for (int r = 0; r < rounds; ++r) {
                for (int it = 0; it < per_group; ++it) {
                    const int pi = group + it * ngroups;
                    int ci = 0, cj = 0;
                    bool valid = pi < pairs;
                    if (valid) {
                        jacobi_pair(r, pi, Cp, &ci, &cj);
                        valid = (ci < k) && (cj < k);
                    }
                    double2* xi = X + (size_t)ci * n;
                    double2* xj = X + (size_t)cj * n;
                    double alpha = 0.0, beta = 0.0, gr = 0.0, gi = 0.0;
                    if (valid) {
                        for (int row = t; row < n; row += JAC_TG) {
                            const double2 a = xi[row], c2 = xj[row];
                            alpha += cabs2(a);
                            beta += cabs2(c2);
                            gr += a.x * c2.x + a.y * c2.y;     // conj(a) * c2
                            gi += a.x * c2.y - a.y * c2.x;
                        }
                    }
                    for (int o = JAC_TG / 2; o > 0; o >>= 1) {
                        alpha += __shfl_xor(alpha, o, JAC_TG);
                        beta += __shfl_xor(beta, o, JAC_TG);
                        gr += __shfl_xor(gr, o, JAC_TG);
                        gi += __shfl_xor(gi, o, JAC_TG);
                    }
                    const double g2 = gr * gr + gi * gi;
                    if (valid && g2 > (JAC_TOL * JAC_TOL) * alpha * beta && g2 > 0.0) {
                        const double gabs = sqrt(g2);
                        const double zeta = (beta - alpha) / (2.0 * gabs);
                        const double tt = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                        const double cs = 1.0 / sqrt(1.0 + tt * tt);
                        const double sn = cs * tt;
                        // e^{-i phi} = conj(gamma)/|gamma|
                        const double2 em = make_double2(gr / gabs, -gi / gabs);
                        for (int row = t; row < n; row += JAC_TG) {
                            const double2 a = xi[row];
                            const double2 bj = cmul(em, xj[row]);
                            xi[row] = make_double2(cs * a.x - sn * bj.x, cs * a.y - sn * bj.y);
                            xj[row] = make_double2(sn * a.x + cs * bj.x, sn * a.y + cs * bj.y);
                        }
                        double2* vi = Vr + (size_t)ci * k;
                        double2* vj = Vr + (size_t)cj * k;
                        for (int row = t; row < k; row += JAC_TG) {
                            const double2 a = vi[row];
                            const double2 bj = cmul(em, vj[row]);
                            vi[row] = make_double2(cs * a.x - sn * bj.x, cs * a.y - sn * bj.y);
                            vj[row] = make_double2(sn * a.x + cs * bj.x, sn * a.y + cs * bj.y);
                        }
                        if (t == 0) s_rotated = 1;
                    }
                }
                __syncthreads();
            }
            const int rotated = s_rotated;
            __syncthreads();
            if (!rotated) break;
        }
    }
    // U[i][j] = sum_c Vr[i][c] conj(W[j][c]) / sigma_c
    for (int cc = tid; cc < k; cc += blockDim.x) {
        const double2* wc = X + (size_t)cc * n;
        double s2 = 0.0;
        for (int row = 0; row < n; ++row) s2 += cabs2(wc[row]);
        s_isig[cc] = s2 > 1e-300 ? 1.0 / sqrt(s2) : 0.0;
    }
    __syncthreads();
    for (int e = tid; e < k * n; e += blockDim.x) {
        const int i = e / n, j = e - i * n;
        double2 acc = make_double2(0.0, 0.0);
        for (int cc = 0; cc < k; ++cc) {
            const double inv = s_isig[cc];
            const double2 p = cmulc(Vr[(size_t)cc * k + i], X[(size_t)cc * n + j]);
            acc.x = fma(inv, p.x, acc.x);
            acc.y = fma(inv, p.y, acc.y);
        }
        U[e] = acc;
    }
}

// ---- LDS-resident variant -------------------------------------------------------------------------------
// X (k columns x n rows) and V_r (k x k) live in LDS, both column-major with an odd column length so that
// the 8 lanes of a pair-group (consecutive rows) and different groups (different columns) spread over the
// banks.  One barrier per tournament round.  Warm start: the caller passes X' = X V_r_prev together with
// V_r_prev (the right singular vectors move little between phasing steps), so 1-3 sweeps suffice; the
// sweep loop stops early when the largest relative off-diagonal of a sweep predicts (quadratic
// convergence) that the next one would be below tolerance.  Output: Pn = W Sigma^-1 (overwrites X) and V_r.
#define JL_MAX_THREADS 576

// Rotation parameters of one column pair from its Gram entries (alpha, beta, gamma = gr + i gi): returns false
// when the pair is already orthogonal to tolerance.
__device__ __forceinline__ bool jl_params(double alpha, double beta, double gr, double gi, bool valid, double tabs2,
                                          double S, bool& big, double& cs, double2& w) {
    const double g2 = gr * gr + gi * gi;
    const double ab = alpha * beta;
    if (!(valid && g2 > (JAC_TOL * JAC_TOL) * ab && g2 > tabs2 * fmax(alpha, beta) * S && g2 > 0.0)) return false;
    big = big || (g2 > (JL_EARLY * JL_EARLY) * ab);
    // smaller-angle rotation [a b] <- [a b] [[c, conj(w)], [-w, c]] that annihilates gamma = a^+ b:  with
    // d = (beta - alpha)/2, h = sqrt(d^2 + |gamma|^2):  c^2 = (1 + |d|/h)/2,  w = s e = sign(d) conj(gamma) / (2 h c).
    // Only two reciprocal square roots, no division and no |gamma| (the phase e and the sine never appear alone).
    const double d = 0.5 * (beta - alpha);
    const double ih = fast_rsqrt(fma(d, d, g2));
    const double c2 = fma(0.5 * fabs(d), ih, 0.5);
    const double rc = fast_rsqrt(c2);
    cs = c2 * rc;
    const double kappa = (d >= 0.0 ? 0.5 : -0.5) * ih * rc;
    w = make_double2(kappa * gr, -kappa * gi);
    return true;
}

// [a b] <- [a b] [[c, conj(w)], [-w, c]]:  a' = c a - w b,  b' = conj(w) a + c b
__device__ __forceinline__ void jl_rotate(double cs, double2 w, double2 a, double2 b, double2& an, double2& bn) {
    an.x = fma(w.y, b.y, fma(-w.x, b.x, cs * a.x));
    an.y = fma(-w.y, b.x, fma(-w.x, b.y, cs * a.y));
    bn.x = fma(w.y, a.y, fma(w.x, a.x, cs * b.x));
    bn.y = fma(-w.y, a.x, fma(w.x, a.y, cs * b.y));
}

// One column pair of a tournament round, NR row slots per lane for X and for V_r, no branches inside the row
// loops: only the last slot can fall outside the column (xl_ok / vl_ok), its loads are issued anyway (the
// address stays inside the LDS allocation) and zeroed by a select, its stores are predicated.  The V_r loads
// are issued before the rotation parameters are known so that their latency hides behind that arithmetic.
template <int NR, int TG>
__device__ __forceinline__ void jl_pair(double2* xi, double2* xj, double2* vi, double2* vj, bool xl_ok, bool vl_ok,
                                        bool valid, double tabs2, double S, bool& big) {
    double2 ra[NR], rb[NR], va[NR], vb[NR];
#pragma unroll
    for (int u = 0; u < NR; ++u) {
        ra[u] = xi[u * TG];
        rb[u] = xj[u * TG];
    }
    if (!xl_ok) {
        ra[NR - 1] = make_double2(0.0, 0.0);
        rb[NR - 1] = make_double2(0.0, 0.0);
    }
    double alpha = 0.0, beta = 0.0, gr = 0.0, gi = 0.0;
#pragma unroll
    for (int u = 0; u < NR; ++u) {
        const double2 a = ra[u], c2 = rb[u];
        alpha += cabs2(a);
        beta += cabs2(c2);
        gr += a.x * c2.x + a.y * c2.y;                     // conj(a) * c2
        gi += a.x * c2.y - a.y * c2.x;
    }
    group_sum4<TG>(alpha, beta, gr, gi);
    double cs;
    double2 w;
    if (!jl_params(alpha, beta, gr, gi, valid, tabs2, S, big, cs, w)) return;
#pragma unroll
    for (int u = 0; u < NR; ++u) {
        va[u] = vi[u * TG];
        vb[u] = vj[u * TG];
    }
#pragma unroll
    for (int u = 0; u < NR; ++u) {
        double2 an, bn;
        jl_rotate(cs, w, ra[u], rb[u], an, bn);
        if (u < NR - 1 || xl_ok) {
            xi[u * TG] = an;
            xj[u * TG] = bn;
        }
    }
#pragma unroll
    for (int u = 0; u < NR; ++u) {
        double2 an, bn;
        jl_rotate(cs, w, va[u], vb[u], an, bn);
        if (u < NR - 1 || vl_ok) {
            vi[u * TG] = an;
            vj[u * TG] = bn;
        }
    }
}

// same with run-time row counts (k_l != 2l+1, or more row slots than the specialised bodies cover)
template <int MAXR, int TG>
__device__ __forceinline__ void jl_pair_generic(double2* xi, double2* xj, double2* vi, double2* vj, int nr, int kr, int n,
                                                int k, int pad, int t, bool valid, double tabs2, double S, bool& big) {
    double2 ra[MAXR], rb[MAXR];
    double alpha = 0.0, beta = 0.0, gr = 0.0, gi = 0.0;
#pragma unroll
    for (int u = 0; u < MAXR; ++u) {
        ra[u] = make_double2(0.0, 0.0);
        rb[u] = make_double2(0.0, 0.0);
        if (u < nr && (pad || t + u * TG < n)) {
            const double2 a = xi[u * TG], c2 = xj[u * TG];
            ra[u] = a;
            rb[u] = c2;
            alpha += cabs2(a);
            beta += cabs2(c2);
            gr += a.x * c2.x + a.y * c2.y;
            gi += a.x * c2.y - a.y * c2.x;
        }
    }
    group_sum4<TG>(alpha, beta, gr, gi);
    double cs;
    double2 w;
    if (!jl_params(alpha, beta, gr, gi, valid, tabs2, S, big, cs, w)) return;
#pragma unroll
    for (int u = 0; u < MAXR; ++u) {
        if (u < nr && (pad || t + u * TG < n)) {
            double2 an, bn;
            jl_rotate(cs, w, ra[u], rb[u], an, bn);
            xi[u * TG] = an;
            xj[u * TG] = bn;
        }
    }
#pragma unroll
    for (int u = 0; u < MAXR; ++u) {
        if (u < kr && (pad || t + u * TG < k)) {
            double2 an, bn;
            jl_rotate(cs, w, vi[u * TG], vj[u * TG], an, bn);
            vi[u * TG] = an;
            vj[u * TG] = bn;
        }
    }
}

// ---- resident-column ordering ------------------------------------------------------------------------------
// The tournament round is bound by LDS traffic (every column of X and V_r read and written once per round, all
// waves in step because of the barrier).  With the divide-and-conquer ordering below a pair-group keeps ONE column
// (X and V_r rows, NR + NR registers per lane) resident in registers over a whole phase and only the other column
// ("mover") goes through LDS: a set of columns is split into halves A (floor) and B (ceil); for |B| rounds group
// i pairs resident A_i with mover B_((i + r) mod |B|); then A and B are solved recursively side by side on
// disjoint groups (the groups of A keep their residents).  Every pair meets once per sweep, floor(k/2) groups
// (32 = 8 waves, two per SIMD, at k = 65) are busy in every round, and the LDS traffic per round halves.  The schedule (which resident / mover each group takes in each round, and when a
// resident must be written back because another group needs it next) is precomputed on the host for every
// column count (build_jacobi_schedule) and verified there.

template <int NR, int TG>
__device__ __forceinline__ void jl_sweep_resident(double2* Xs, double2* Vs, int ns, int ks, int t, int group,
                                                  const int* __restrict__ tab, int n_rounds, int ps, const int* s_perm,
                                                  bool xl_ok, bool vl_ok, double tabs2, double S, bool& big) {
    double2 rx[NR], rv[NR];
#pragma unroll
    for (int u = 0; u < NR; ++u) rx[u] = make_double2(0.0, 0.0);
#pragma unroll
    for (int u = 0; u < NR; ++u) rv[u] = make_double2(0.0, 0.0);
    int cur = -1;                                            // compact index of the resident column
    bool dirty = false;
    // table entry and physical columns (through s_perm) of a round are resolved during the previous round
    int e_next = (group < ps && n_rounds > 0) ? tab[group] : 0;
    int pr_next = s_perm[(e_next & JS_ACTIVE) ? (e_next & 255) : 0], pm_next = s_perm[(e_next & JS_ACTIVE) ? ((e_next >> 8) & 255) : 0];
    for (int r = 0; r < n_rounds; ++r) {
        const int e = e_next;
        const int pr = pr_next, pm = pm_next;
        if (r + 1 < n_rounds && group < ps) e_next = tab[(size_t)(r + 1) * ps + group];   // in flight during the round
        // (the wave-level reductions run for every group, active or not: uniform control flow around DPP)
        const bool act = (e & JS_ACTIVE) != 0;
        const int res = act ? (e & 255) : 0;
        double2* xh = Xs + (size_t)pr * ns + t;
        double2* vh = Vs + (size_t)pr * ks + t;
        double2* xm = Xs + (size_t)pm * ns + t;
        double2* vm = Vs + (size_t)pm * ks + t;
        double2 mx[NR], mv[NR];
#pragma unroll
        for (int u = 0; u < NR; ++u) mx[u] = make_double2(0.0, 0.0);
        if (act) {
#pragma unroll
            for (int u = 0; u < NR; ++u) mx[u] = xm[u * TG];
            if (res != cur) {
#pragma unroll
                for (int u = 0; u < NR; ++u) {
                    rx[u] = xh[u * TG];
                    rv[u] = vh[u * TG];
                }
                if (!xl_ok) rx[NR - 1] = make_double2(0.0, 0.0);
                cur = res;
                dirty = false;
            }
            if (!xl_ok) mx[NR - 1] = make_double2(0.0, 0.0);
        }
        double alpha = 0.0, beta = 0.0, gr = 0.0, gi = 0.0;
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const double2 a = rx[u], c2 = mx[u];
            alpha += cabs2(a);
            beta += cabs2(c2);
            gr += a.x * c2.x + a.y * c2.y;                     // conj(a) * c2
            gi += a.x * c2.y - a.y * c2.x;
        }
        group_sum4<TG>(alpha, beta, gr, gi);
        double cs = 1.0;
        double2 w = make_double2(0.0, 0.0);
        const bool rot = jl_params(alpha, beta, gr, gi, act, tabs2, S, big, cs, w);
        if (rot) {
#pragma unroll
            for (int u = 0; u < NR; ++u) mv[u] = vm[u * TG];
#pragma unroll
            for (int u = 0; u < NR; ++u) {
                double2 an, bn;
                jl_rotate(cs, w, rx[u], mx[u], an, bn);
                rx[u] = an;
                if (u < NR - 1 || xl_ok) xm[u * TG] = bn;
            }
#pragma unroll
            for (int u = 0; u < NR; ++u) {
                double2 an, bn;
                jl_rotate(cs, w, rv[u], mv[u], an, bn);
                rv[u] = an;
                if (u < NR - 1 || vl_ok) vm[u * TG] = bn;
            }
            dirty = true;
        }
        if (act && (e & JS_WB)) {                              // someone else takes this column next round
            if (dirty) {
#pragma unroll
                for (int u = 0; u < NR; ++u) {
                    if (u < NR - 1 || xl_ok) xh[u * TG] = rx[u];
                    if (u < NR - 1 || vl_ok) vh[u * TG] = rv[u];
                }
            }
            cur = -1;
        }
        pr_next = s_perm[(e_next & JS_ACTIVE) ? (e_next & 255) : 0];
        pm_next = s_perm[(e_next & JS_ACTIVE) ? ((e_next >> 8) & 255) : 0];
        __syncthreads();
    }
}

// MAXR = upper bound of the rows of a column handled by one lane of a pair-group (ceil(n / 8) <= MAXR).  Columns
// are zero padded to a multiple of 8 rows in LDS, so the row loops need no per-lane predicate; the two columns of
// the pair stay in registers between the Gram reduction and the rotation.  For odd k the tournament pair that
// contains the dummy player is skipped, so ceil(k/2) <= 32 pair-groups (one wave per SIMD at k <= 65) suffice.
// arguments of the polar-factor kernels (one struct: the Jacobi body is shared by two kernels)
struct JacobiArgs {
    const double2* Xin_all;
    double2* Pn_all;
    double2* Vr_all;
    const int *kl, *active, *xoff, *roff;
    int xtot, rtot, L, warm;
    double tabs2;
    int* sweeps_out;
    int pad;
    const int *sched, *sched_off, *sched_rounds;
    int sched_ps;
    const int* order_list;
};

// static LDS of a Jacobi workgroup
template <int MAXT>
struct JacShared {
    double gmax[MAXT / 8];
    double isig[128];
    int perm[128];
    int cont, keff;
};

// the workgroup of matrix (restart b, order_list[oy])
template <int MAXR, int TG, int MAXT>
__device__ __forceinline__ void polar_jacobi_body(const JacobiArgs& A, int b, int oy, JacShared<MAXT>& sh) {
    const double2* __restrict__ Xin_all = A.Xin_all;
    double2* __restrict__ Pn_all = A.Pn_all;
    double2* __restrict__ Vr_all = A.Vr_all;
    const int* __restrict__ kl = A.kl;
    const int* __restrict__ active = A.active;
    const int* __restrict__ xoff = A.xoff;
    const int* __restrict__ roff = A.roff;
    const int xtot = A.xtot, rtot = A.rtot, L = A.L, warm = A.warm, pad = A.pad, sched_ps = A.sched_ps;
    const double tabs2 = A.tabs2;
    int* __restrict__ sweeps_out = A.sweeps_out;
    const int* __restrict__ sched = A.sched;
    const int* __restrict__ sched_off = A.sched_off;
    const int* __restrict__ sched_rounds = A.sched_rounds;
    HIP_DYNAMIC_SHARED(double2, sm)
    double* const s_gmax = sh.gmax;
    double* const s_isig = sh.isig;
    int* const s_perm = sh.perm;
    int& s_continue = sh.cont;
    int& s_keff = sh.keff;
    const int l = A.order_list[oy];
    if (!active[l]) return;                                // uniform per block
    const int k = kl[l], n = 2 * l + 1;
    const int nr = (n + TG - 1) / TG, kr = (k + TG - 1) / TG;   // rows per lane
    // column strides: odd; padded to whole lane-groups of rows when LDS allows (pad), else row predicates
    const int ns = pad ? nr * TG + 1 : (n | 1), ks = pad ? kr * TG + 1 : (k | 1);
    double2* Xs = sm;
    double2* Vs = sm + (size_t)k * ns;
    const double2* Xin = Xin_all + (size_t)b * xtot + xoff[l];
    double2* Pn = Pn_all + (size_t)b * xtot + xoff[l];
    double2* Vr = Vr_all + (size_t)b * rtot + roff[l];
    const int tid = threadIdx.x;
    for (int e = tid; e < k * ns; e += blockDim.x) {
        const int cc = e / ns, r = e - cc * ns;
        Xs[e] = r < n ? Xin[(size_t)cc * n + r] : make_double2(0.0, 0.0);
    }
    for (int e = tid; e < k * ks; e += blockDim.x) {
        const int cc = e / ks, i = e - cc * ks;
        double2 v = make_double2(0.0, 0.0);
        if (i < k) v = warm ? Vr[(size_t)cc * k + i] : make_double2(cc == i ? 1.0 : 0.0, 0.0);
        Vs[e] = v;
    }
    __syncthreads();
    const int ngroups = blockDim.x / TG;
    const int group = tid / TG, t = tid - group * TG;
    double S = 0.0;
    if (k > 1) {
        for (int sweep = 0; sweep < JAC_MAX_SWEEPS; ++sweep) {
            // Deflation: a column that has shrunk below eps * (largest column) is a numerical zero -- it only
            // carries rounding noise of the big columns.  X = I_l^+ D^2 V_l is numerically rank deficient once
            // the density has a support (singular value ratios < 1e-17), and without this the noise columns
            // keep the relative criterion busy for ~10 extra sweeps.  Their contribution to V_l U_l is
            // O(eps) (their left vectors get sigma = 0 -> Pn = 0).
            double Sl = 0.0;
            for (int cc0 = 0; cc0 < k; cc0 += ngroups) {
                const int cc = cc0 + group;
                double s2 = 0.0;
                if (cc < k)
                    for (int u = 0; u < nr; ++u)
                        if (pad || t + u * TG < n) s2 += cabs2(Xs[(size_t)cc * ns + t + u * TG]);
                s2 = group_sum<TG>(s2);
                if (cc < k && t == 0) s_isig[cc] = s2;
                Sl = fmax(Sl, s2);
            }
            if (t == 0) s_gmax[group] = Sl;
            __syncthreads();
            S = 0.0;
            for (int g = 0; g < ngroups; ++g) S = fmax(S, s_gmax[g]);
            for (int cc0 = 0; cc0 < k; cc0 += ngroups) {
                const int cc = cc0 + group;
                if (cc < k && s_isig[cc] <= (JAC_DEFLATE * JAC_DEFLATE) * S && s_isig[cc] > 0.0)
                    for (int u = 0; u < nr; ++u)
                        if (pad || t + u * TG < n) Xs[(size_t)cc * ns + t + u * TG] = make_double2(0.0, 0.0);
            }
            // Compaction: the tournament only runs over the columns that are still non-zero (numerical rank of X;
            // about half of 2l+1 once the density has a support), through the index list s_perm.
            if (tid < 64) {                                    // wave 0: ballot + prefix popcount (k <= 128)
                const double thr = (JAC_DEFLATE * JAC_DEFLATE) * S;
                const bool a0 = tid < k && s_isig[tid] > thr;
                const bool a1 = tid + 64 < k && s_isig[tid + 64] > thr;
                const unsigned long long m0 = __ballot(a0), m1 = __ballot(a1);
                const unsigned long long below = (1ull << tid) - 1ull;
                if (a0) s_perm[__popcll(m0 & below)] = tid;
                if (a1) s_perm[__popcll(m0) + __popcll(m1 & below)] = tid + 64;
                if (tid == 0) s_keff = __popcll(m0) + __popcll(m1);
            }
            __syncthreads();
            const int ke = s_keff;
            const int Cp = ke + (ke & 1);
            const int rounds = Cp - 1;
            const int skip = ke & 1;                           // odd count: pair 0 of every round holds the dummy player
            const int pairs = Cp / 2 - skip;
            const int per_group = (pairs + ngroups - 1) / ngroups;
            bool big = false;                                  // some pair of this group was above the early-exit level
            // resident-column ordering when it applies (16-lane groups, equal row counts, enough groups)
            constexpr int RMAX = (TG == 16 && MAXR >= 7) ? 7 : 5;
            const bool resident = TG == 16 && sched != nullptr && nr == kr && nr <= RMAX && sched_ps <= ngroups;
            if (resident) {
                const bool xl_ok = pad || t + (nr - 1) * TG < n, vl_ok = pad || t + (kr - 1) * TG < k;
                const int* tab = sched + sched_off[ke];
                const int nrd = sched_rounds[ke];
#define JL_SWEEP_T(NR, TAB) jl_sweep_resident<NR, TG>(Xs, Vs, ns, ks, t, group, TAB, nrd, sched_ps, s_perm, xl_ok, vl_ok, tabs2, S, big)
#define JL_SWEEP_ALL(TAB)                              \
                switch (nr) {                              \
                case 1: JL_SWEEP_T(1, TAB); break;         \
                case 2: JL_SWEEP_T(2, TAB); break;         \
                case 3: JL_SWEEP_T(3, TAB); break;         \
                case 4: JL_SWEEP_T(4, TAB); break;         \
                case 5: JL_SWEEP_T(5, TAB); break;         \
                case 6: JL_SWEEP_T((RMAX >= 7 ? 6 : 1), TAB); break;   \
                default: JL_SWEEP_T((RMAX >= 7 ? 7 : 1), TAB); break;  \
                }
                JL_SWEEP_ALL(tab)
#undef JL_SWEEP_ALL
#undef JL_SWEEP_T
            }
            for (int r = 0; r < (resident ? 0 : rounds); ++r) {
                for (int it = 0; it < per_group; ++it) {
                    const int pi = group + it * ngroups;
                    int ci = 0, cj = 0;
                    bool valid = pi < pairs;
                    if (valid) {
                        jacobi_pair(r, pi + skip, Cp, &ci, &cj);
                        valid = (ci < ke) && (cj < ke);
                        ci = valid ? s_perm[ci] : 0;
                        cj = valid ? s_perm[cj] : 0;
                    }
                    double2* xi = Xs + (size_t)ci * ns + t;
                    double2* xj = Xs + (size_t)cj * ns + t;
                    double2* vi = Vs + (size_t)ci * ks + t;
                    double2* vj = Vs + (size_t)cj * ks + t;
                    // last row slot of this lane: inside the column?  (all earlier slots always are)
                    const bool xl_ok = pad || t + (nr - 1) * TG < n, vl_ok = pad || t + (kr - 1) * TG < k;
                    // block-uniform dispatch to a branch-free body with compile-time row counts
                    bool done = false;
                    if (nr == kr) {
                        done = true;
                        switch (nr) {
                        case 1: jl_pair<1, TG>(xi, xj, vi, vj, xl_ok, vl_ok, valid, tabs2, S, big); break;
                        case 2: jl_pair<2, TG>(xi, xj, vi, vj, xl_ok, vl_ok, valid, tabs2, S, big); break;
                        case 3: jl_pair<3, TG>(xi, xj, vi, vj, xl_ok, vl_ok, valid, tabs2, S, big); break;
                        case 4: jl_pair<4, TG>(xi, xj, vi, vj, xl_ok, vl_ok, valid, tabs2, S, big); break;
                        case 5: jl_pair<5, TG>(xi, xj, vi, vj, xl_ok, vl_ok, valid, tabs2, S, big); break;
                        default: done = false;
                        }
                        if (MAXR > 5 && !done) {
                            done = true;
                            switch (nr) {
                            case 6: jl_pair<(MAXR > 5 ? 6 : 1), TG>(xi, xj, vi, vj, xl_ok, vl_ok, valid, tabs2, S, big); break;
                            case 7: jl_pair<(MAXR > 5 ? 7 : 1), TG>(xi, xj, vi, vj, xl_ok, vl_ok, valid, tabs2, S, big); break;
                            case 8: jl_pair<(MAXR > 5 ? 8 : 1), TG>(xi, xj, vi, vj, xl_ok, vl_ok, valid, tabs2, S, big); break;
                            case 9: jl_pair<(MAXR > 5 ? 9 : 1), TG>(xi, xj, vi, vj, xl_ok, vl_ok, valid, tabs2, S, big); break;
                            default: done = false;
                            }
                        }
                    }
                    if (!done) jl_pair_generic<MAXR, TG>(xi, xj, vi, vj, nr, kr, n, k, pad, t, valid, tabs2, S, big);
                }
                __syncthreads();
            }
            if (t == 0) s_gmax[group] = big ? 1.0 : 0.0;
            __syncthreads();
            if (tid == 0) {
                double m = 0.0;
                for (int g = 0; g < ngroups; ++g) m = fmax(m, s_gmax[g]);
                // quadratic convergence: a sweep whose largest |gamma|/sqrt(alpha beta) stayed below JL_EARLY leaves
                // ~JL_EARLY^2 behind, so stop unless some pair was above that level
                s_continue = (m > 0.0) ? 1 : 0;
                sweeps_out[b * (L + 1) + l] = (sweep + 1) | (ke << 8);
            }
            __syncthreads();
            const int cont = s_continue;
            __syncthreads();
            if (!cont) break;
        }
    }
    // sigma_c and the normalised columns Pn = W Sigma^-1
    for (int cc0 = 0; cc0 < k; cc0 += ngroups) {           // uniform trip count: DPP sums need the whole group
        const int cc = cc0 + group;
        double s2 = 0.0;
        if (cc < k)
            for (int u = 0; u < nr; ++u)
                if (pad || t + u * TG < n) s2 += cabs2(Xs[(size_t)cc * ns + t + u * TG]);
        s2 = group_sum<TG>(s2);
        if (cc < k && t == 0) s_isig[cc] = s2 > 1e-300 ? 1.0 / sqrt(s2) : 0.0;
    }
    __syncthreads();
    for (int e = tid; e < k * n; e += blockDim.x) {
        const int cc = e / n, r = e - cc * n;
        Pn[e] = cscale(Xs[(size_t)cc * ns + r], s_isig[cc]);
    }
    for (int e = tid; e < k * k; e += blockDim.x) {
        const int cc = e / k, i = e - cc * k;
        Vr[e] = Vs[(size_t)cc * ks + i];
    }
}

// grid = (restart, rank of the order among the active ones): the restart index runs fastest, so the heaviest order
// of EVERY restart is dispatched first -- these workgroups are the critical path of the launch
template <int MAXR, int TG, int MAXT>
__global__ void __launch_bounds__(MAXT) k_polar_jacobi_lds(JacobiArgs A) {
    __shared__ JacShared<MAXT> sh;
    polar_jacobi_body<MAXR, TG, MAXT>(A, (int)blockIdx.x, (int)blockIdx.y, sh);
}

// ---- the four complex products around the polar factor -------------------------------------------------------------
// X_l = I_l^+ D^2 V_l, the warm start X_l V_r, U_l = V_r Pn^+ and I'_l = V_l U_l are small batched complex GEMMs
// (<= 128 x 65 x 65 per (restart, l)) on the f64 matrix cores (k_proj_mfma and the fused pairs k_proj_xw / k_proj_ua below).
struct ProjGemmArgs {
    const double2* Ilm;
    const double2* V;
    const double2* X;           // d_X: X_l, later Pn
    const double2* Xw;          // d_U: warm-start buffer, later U
    const double2* Vr;
    double2* dst;
    const double* q;
    const uint8_t* rmask;
    const int *kl, *active, *used, *voff, *xoff, *uoff;
    int N, L, xtot, utot, nlm;
    double inv_sqrt_np;
};

enum { PG_X = 0, PG_WARM = 1, PG_U = 2, PG_APPLY = 3 };

// operand views of one (restart, l) product: element (i, j) at base[i * si + j * sj], optionally conjugated
struct PgView {
    const double2* base;
    int si, sj;
};

template <int OP>
struct ProjGemm {
    static constexpr bool A_M_FAST = (OP != PG_APPLY);      // memory-contiguous index of the A operand: m (else k)
    static constexpr bool B_N_FAST = (OP != PG_WARM);
    static constexpr bool A_CONJ = (OP == PG_X), B_CONJ = (OP == PG_U);
    // dims and operand views (A: rows m, columns kk;  B: rows kk, columns nn) for order l of restart b
    static __device__ __forceinline__ void setup(const ProjGemmArgs& a, int b, int l, int& M, int& Nn, int& K, PgView& A,
                                                 PgView& Bv) {
        const int k = a.kl[l], n = 2 * l + 1;
        if (OP == PG_X) {                                   // conj(I_l[q][r]) q^2  x  V_l[q][c]
            M = n; Nn = k; K = a.N;
            A = PgView{a.Ilm + (size_t)b * a.N * a.nlm + l * l, 1, a.nlm};
            Bv = PgView{a.V + a.voff[l], k, 1};
        } else if (OP == PG_WARM) {                         // X[j][r]  x  Vr[c][j]
            M = n; Nn = k; K = k;
            A = PgView{a.X + (size_t)b * a.xtot + a.xoff[l], 1, n};
            Bv = PgView{a.Vr + (size_t)b * a.utot + a.uoff[l], 1, k};
        } else if (OP == PG_U) {                            // Vr[c][i]  x  conj(Pn[c][j])
            M = k; Nn = n; K = k;
            A = PgView{a.Vr + (size_t)b * a.utot + a.uoff[l], 1, k};
            Bv = PgView{a.X + (size_t)b * a.xtot + a.xoff[l], n, 1};
        } else {                                            // V_l[q][i]  x  U[i][j]
            M = a.N; Nn = n; K = k;
            A = PgView{a.V + a.voff[l], k, 1};
            Bv = PgView{a.Xw + (size_t)b * a.xtot + a.xoff[l], n, 1};
        }
    }
    static __device__ __forceinline__ void store(const ProjGemmArgs& a, int b, int l, int k, int n, int xo, int m, int nn,
                                                 double2 acc) {
        if (OP == PG_X || OP == PG_WARM) {                  // column-major n x k
            a.dst[(size_t)b * a.xtot + xo + (size_t)nn * n + m] = acc;
        } else if (OP == PG_U) {                            // row-major k x n
            a.dst[(size_t)b * a.xtot + xo + (size_t)m * n + nn] = acc;
        } else {                                            // I'_l[q = m][j = nn]: mask ? V_l U_l : I_l (+ l = 0 rules)
            // in place on the coefficient buffer (dst holds I_l): only used orders have tiles, only changed elements
            // are written
            if (!a.used[l]) return;
            const size_t idx = ((size_t)b * a.N + m) * a.nlm + l * l + nn;
            const bool masked = a.rmask[(size_t)l * a.N + m] != 0;
            if (l == 0) {
                const double2 v = masked ? a.V[a.voff[0] + (size_t)m * k] : a.dst[idx];    // fxs_Projections.py:840
                a.dst[idx] = cscale(v, a.inv_sqrt_np);                                      // fxs_Projections.py:870
            } else if (masked) {
                a.dst[idx] = acc;
            }
        }
    }
    static __device__ __forceinline__ bool has_product(const ProjGemmArgs& a, int l) {
        if (OP == PG_APPLY) return a.used[l] && l > 0;
        return a.active[l] != 0;
    }
};

// ---- the four products on the f64 matrix cores ------------------------------------------------------------
// The grid is the compact list of (order, tile) pairs that exist (host-built, `tiles`): a dense (max tiles) x (L+1) x B grid is
// mostly workgroups that return at once, and dispatching them costs more than the products.  They are GEMMs, so they run on v_mfma_f64_16x16x4 with the fragments read straight from global memory (no LDS,
// no barriers): a wave owns one 16 x 16 complex output tile (two accumulators), the four waves of a workgroup share
// the row tile.  Complex as real: with A read as rows [re, im, ...] over the inner index (K = 2 x inner extent),
//   Re C = sum_K A[i][K] B1[K][j],  B1[(k,re)] = Re B, B1[(k,im)] = -Im B;   Im C = sum_K A[i][K] B2[K][j],
//   B2[(k,re)] = Im B, B2[(k,im)] = Re B,
// i.e. every lane loads ONE complex element of A and one of B per k-step of 2 complex inner indices and picks its
// half.  PM_PF k-steps of branch-free loads are in flight (clamped addresses, zeros by select).
#define PM_PF 4

template <int OP>
__global__ void __launch_bounds__(256) k_proj_mfma(ProjGemmArgs a, const int* __restrict__ tiles) {
    typedef ProjGemm<OP> G;
    const int tinfo = tiles[blockIdx.x];
    const int l = tinfo & 255, tile_m = (tinfo >> 8) & 255, tile_g = tinfo >> 16;
    const int b = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool prod = G::has_product(a, l);
    if (!prod && OP != PG_APPLY) return;
    int M, Nn, K;
    PgView A, Bv;
    G::setup(a, b, l, M, Nn, K, A, Bv);
    const int k = a.kl[l], n = 2 * l + 1, xo = a.xoff[l];
    const int m0 = tile_m * 16, n0 = (tile_g * 4 + wave) * 16;
    if (m0 >= M || n0 >= Nn) return;                             // wave-uniform
    const int li = lane & 15, kk = lane >> 4;
    const int ri = kk & 1, kh = kk >> 1;                         // half of the complex number, inner index within the k-step
    v4f64 acc_re = v4f64{0.0, 0.0, 0.0, 0.0}, acc_im = acc_re;
    if (prod) {
        const int mi = m0 + li, nj = n0 + li;
        const bool m_ok = mi < M, n_ok = nj < Nn;
        const double2* ap = A.base + (size_t)(m_ok ? mi : 0) * A.si;
        const double2* bp = Bv.base + (size_t)(n_ok ? nj : 0) * Bv.sj;
        const int n_steps = (K + 1) / 2;
        double fa[PM_PF], f1[PM_PF], f2[PM_PF], ga[PM_PF], g1[PM_PF], g2[PM_PF];
        auto request = [&](int s0, double (&xa)[PM_PF], double (&x1)[PM_PF], double (&x2)[PM_PF]) {
#pragma unroll
            for (int u = 0; u < PM_PF; ++u) {
                const int kc = 2 * (s0 + u) + kh;
                const bool k_ok = kc < K;
                const int kq = k_ok ? kc : 0;
                double2 va = ap[(size_t)kq * A.sj];
                double2 vb = bp[(size_t)kq * Bv.si];
                if (OP == PG_X) {
                    const double qq = a.q[kq];
                    va = make_double2(qq * qq * va.x, -qq * qq * va.y);
                }
                if (G::B_CONJ) vb.y = -vb.y;
                const bool aok = k_ok && m_ok, bok = k_ok && n_ok;
                xa[u] = aok ? (ri ? va.y : va.x) : 0.0;
                x1[u] = bok ? (ri ? -vb.y : vb.x) : 0.0;
                x2[u] = bok ? (ri ? vb.x : vb.y) : 0.0;
            }
        };
        request(0, fa, f1, f2);
        for (int s0 = 0; s0 < n_steps; s0 += 2 * PM_PF) {
            request(s0 + PM_PF, ga, g1, g2);                     // beyond K: clamped, zeroed
#pragma unroll
            for (int u = 0; u < PM_PF; ++u) {
                acc_re = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[u], f1[u], acc_re, 0, 0, 0);
                acc_im = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[u], f2[u], acc_im, 0, 0, 0);
            }
            request(s0 + 2 * PM_PF, fa, f1, f2);
#pragma unroll
            for (int u = 0; u < PM_PF; ++u) {
                acc_re = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[u], g1[u], acc_re, 0, 0, 0);
                acc_im = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[u], g2[u], acc_im, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int mm = m0 + kk + 4 * r, nn = n0 + li;
        if (mm < M && nn < Nn) G::store(a, b, l, k, n, xo, mm, nn, make_double2(acc_re[r], acc_im[r]));
    }
}

// ---- two fused pairs of those products (round 2) ----------------------------------------------------------------
// A step of a 3-restart engine spends ~55 us in the four GEMM kernels for 0.1 GFLOP: they are short latency chains, and
// every kernel boundary of a stream costs its own start-up.  `W = X V_r` only needs the ROWS of X a workgroup has just
// computed, and `I'_l = V_l U_l` only the COLUMNS of U: so
//   k_proj_xw:  a workgroup owns 16 rows of X_l (all columns, one 16 x 16 tile per wave), keeps them in LDS and multiplies
//               them with V_r of the previous step -> W (the warm-start matrix the Jacobi kernel works on); X_l itself is
//               never stored (the cold start, every 64th call, runs the plain PG_X product instead);
//   k_proj_ua:  a workgroup owns 16 columns of U_l = V_r Pn^+ (all rows, a tile per wave), stores them (unknowns output)
//               and keeps them in LDS as the B operand of its column block of I'_l = V_l U_l (a row tile per wave).
// Same fragments, same order of the inner index as k_proj_mfma: the results are bit-identical to the four-kernel path.

// one 16 x 16 complex tile: acc += sum_kc A[kc] B[kc];  ap / bp point at inner index 0 of this lane's row of A / column of B,
// consecutive inner indices a_sk / b_sk elements apart (global or LDS), XQ: A = conj(a) q^2 (the PG_X operand)
template <bool XQ, bool B_CONJ>
__device__ __forceinline__ void pm_accumulate(const double2* ap, size_t a_sk, bool m_ok, const double2* bp, size_t b_sk, bool n_ok,
                                              int K, const double* __restrict__ q, int ri, int kh, v4f64& acc_re, v4f64& acc_im) {
    const int n_steps = (K + 1) / 2;
    double fa[PM_PF], f1[PM_PF], f2[PM_PF], ga[PM_PF], g1[PM_PF], g2[PM_PF];
    auto request = [&](int s0, double (&xa)[PM_PF], double (&x1)[PM_PF], double (&x2)[PM_PF]) {
#pragma unroll
        for (int u = 0; u < PM_PF; ++u) {
            const int kc = 2 * (s0 + u) + kh;
            const bool k_ok = kc < K;
            const int kq = k_ok ? kc : 0;
            double2 va = ap[(size_t)kq * a_sk];
            double2 vb = bp[(size_t)kq * b_sk];
            if (XQ) {
                const double qq = q[kq];
                va = make_double2(qq * qq * va.x, -qq * qq * va.y);
            }
            if (B_CONJ) vb.y = -vb.y;
            const bool aok = k_ok && m_ok, bok = k_ok && n_ok;
            xa[u] = aok ? (ri ? va.y : va.x) : 0.0;
            x1[u] = bok ? (ri ? -vb.y : vb.x) : 0.0;
            x2[u] = bok ? (ri ? vb.x : vb.y) : 0.0;
        }
    };
    request(0, fa, f1, f2);
    for (int s0 = 0; s0 < n_steps; s0 += 2 * PM_PF) {
        request(s0 + PM_PF, ga, g1, g2);                         // beyond K: clamped, zeroed
#pragma unroll
        for (int u = 0; u < PM_PF; ++u) {
            acc_re = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[u], f1[u], acc_re, 0, 0, 0);
            acc_im = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[u], f2[u], acc_im, 0, 0, 0);
        }
        request(s0 + 2 * PM_PF, fa, f1, f2);
#pragma unroll
        for (int u = 0; u < PM_PF; ++u) {
            acc_re = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[u], g1[u], acc_re, 0, 0, 0);
            acc_im = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[u], g2[u], acc_im, 0, 0, 0);
        }
    }
}

#define PF_THREADS 512          // 8 waves: up to 128 columns of X_l / 128 shells per pass of I'_l

// W[16 rows of tile_m][all k columns] = (conj(I_l)^T q^2 V_l)[rows] V_r:   tiles = order | tile_m << 8
__global__ void __launch_bounds__(PF_THREADS) k_proj_xw(ProjGemmArgs a, const int* __restrict__ tiles) {
    HIP_DYNAMIC_SHARED(double2, sm)
    const int tinfo = tiles[blockIdx.x];
    const int l = tinfo & 255, tile_m = tinfo >> 8;
    const int b = blockIdx.y;
    if (!a.active[l]) return;
    const int k = a.kl[l], n = 2 * l + 1, xo = a.xoff[l];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 15, kk = lane >> 4, ri = kk & 1, kh = kk >> 1;
    const int m0 = tile_m * 16;
    const int ld = k | 1;                                        // LDS row of the X block (odd: rows on different banks)
    double2* Xs = sm;                                            // [16][ld]
    const int ntn = (k + 15) >> 4;
    const int mi = m0 + li;
    const bool m_ok = mi < n;
    // ---- X rows: A = conj(I_l[q][m]) q^2 (inner index q: stride nlm), B = V_l[q][c] (stride k)
    for (int tn = wave; tn < ntn; tn += PF_THREADS / 64) {
        const int nj = tn * 16 + li;
        const bool n_ok = nj < k;
        v4f64 acc_re = v4f64{0.0, 0.0, 0.0, 0.0}, acc_im = acc_re;
        const double2* ap = a.Ilm + (size_t)b * a.N * a.nlm + (size_t)l * l + (m_ok ? mi : 0);
        const double2* bp = a.V + a.voff[l] + (n_ok ? nj : 0);
        pm_accumulate<true, false>(ap, (size_t)a.nlm, m_ok, bp, (size_t)k, n_ok, a.N, a.q, ri, kh, acc_re, acc_im);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ml = kk + 4 * r, nn = tn * 16 + li;
            if (nn < k) Xs[(size_t)ml * ld + nn] = make_double2(acc_re[r], acc_im[r]);     // rows beyond n hold zeros
        }
    }
    __syncthreads();
    // ---- W rows: A = X rows (LDS, inner index j: stride 1), B = V_r[c][j] (column c at c * k, stride 1)
    for (int tn = wave; tn < ntn; tn += PF_THREADS / 64) {
        const int nj = tn * 16 + li;
        const bool n_ok = nj < k;
        v4f64 acc_re = v4f64{0.0, 0.0, 0.0, 0.0}, acc_im = acc_re;
        const double2* ap = Xs + (size_t)li * ld;
        const double2* bp = a.Vr + (size_t)b * a.utot + a.uoff[l] + (size_t)(n_ok ? nj : 0) * k;
        pm_accumulate<false, false>(ap, 1, m_ok, bp, 1, n_ok, k, a.q, ri, kh, acc_re, acc_im);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int mm = m0 + kk + 4 * r, nn = tn * 16 + li;
            if (mm < n && nn < k) a.dst[(size_t)b * a.xtot + xo + (size_t)nn * n + mm] = make_double2(acc_re[r], acc_im[r]);
        }
    }
}

// U[all k rows][16 columns of tile_n] = V_r conj(Pn)^T -> a.Xw (d_U) and LDS;  I'_l[all shells][those columns] = V_l U with the
// mask / l = 0 rules of the PG_APPLY store, in place on a.dst:   tiles = order | tile_n << 8 (every used or solved order)
__global__ void __launch_bounds__(PF_THREADS) k_proj_ua(ProjGemmArgs a, double2* __restrict__ U_out, const int* __restrict__ tiles) {
    typedef ProjGemm<PG_APPLY> G;
    HIP_DYNAMIC_SHARED(double2, sm)
    const int tinfo = tiles[blockIdx.x];
    const int l = tinfo & 255, tile_n = tinfo >> 8;
    const int b = blockIdx.y;
    const bool used = a.used[l] != 0;
    if (!used && !a.active[l]) return;
    const int k = a.kl[l], n = 2 * l + 1, xo = a.xoff[l];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 15, kk = lane >> 4, ri = kk & 1, kh = kk >> 1;
    const int n0 = tile_n * 16;
    const int nj = n0 + li;
    const bool n_ok = nj < n;
    const bool prod = l > 0;
    double2* Us = sm;                                            // [k][16 + 1]: B operand of the second product, inner index i
    constexpr int LDU = 17;
    {
        const int ntm = (k + 15) >> 4;
        if (a.active[l]) {
            // ---- U rows i: A = V_r[c][i] (inner index c: stride k), B = conj(Pn[c][j]) (Pn column-major n x k: stride n)
            for (int tm = wave; tm < ntm; tm += PF_THREADS / 64) {
                const int mi = tm * 16 + li;
                const bool m_ok = mi < k;
                v4f64 acc_re = v4f64{0.0, 0.0, 0.0, 0.0}, acc_im = acc_re;
                const double2* ap = a.Vr + (size_t)b * a.utot + a.uoff[l] + (m_ok ? mi : 0);
                const double2* bp = a.X + (size_t)b * a.xtot + xo + (n_ok ? nj : 0);
                pm_accumulate<false, true>(ap, (size_t)k, m_ok, bp, (size_t)n, n_ok, k, a.q, ri, kh, acc_re, acc_im);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int mm = tm * 16 + kk + 4 * r;
                    if (mm < k) {
                        const double2 v = make_double2(acc_re[r], acc_im[r]);
                        Us[(size_t)mm * LDU + li] = v;                                       // columns beyond n hold zeros
                        if (n_ok) U_out[(size_t)b * a.xtot + xo + (size_t)mm * n + nj] = v;   // row-major k x n
                    }
                }
            }
        } else if (prod) {
            // an order that is used but not solved: its unknowns stay what they are
            for (int e = threadIdx.x; e < k * 16; e += PF_THREADS) {
                const int i = e >> 4, j = e & 15;
                Us[(size_t)i * LDU + j] = (n0 + j < n) ? a.Xw[(size_t)b * a.xtot + xo + (size_t)i * n + n0 + j] : make_double2(0.0, 0.0);
            }
        }
    }
    if (!used) return;                                           // solved but not applied: only its unknowns are wanted
    __syncthreads();
    // ---- I'_l rows q: A = V_l[q][i] (row q at q * k, stride 1), B = U[i][j] (LDS)
    const int ntq = (a.N + 15) >> 4;
    for (int tq = wave; tq < ntq; tq += PF_THREADS / 64) {
        const int mi = tq * 16 + li;
        const bool m_ok = mi < a.N;
        v4f64 acc_re = v4f64{0.0, 0.0, 0.0, 0.0}, acc_im = acc_re;
        if (prod) {
            const double2* ap = a.V + a.voff[l] + (size_t)(m_ok ? mi : 0) * k;
            const double2* bp = Us + li;
            pm_accumulate<false, false>(ap, 1, m_ok, bp, (size_t)LDU, n_ok, k, a.q, ri, kh, acc_re, acc_im);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int mm = tq * 16 + kk + 4 * r;
            if (mm < a.N && n_ok) G::store(a, b, l, k, n, xo, mm, nj, make_double2(acc_re[r], acc_im[r]));
        }
    }
}

// tile lists of the four products (order | tile_m << 8 | tile_n << 16), heavy orders first
static int build_proj_tiles(mtip_ctx* c) {
    if (c->d_pg_tiles[0] != nullptr) return MTIP_OK;
    const int tm_edge = 16, tn_edge = 64;                        // 16 x (4 x 16) outputs per workgroup
    for (int op = 0; op < 4; ++op) {
        std::vector<int> t;
        for (int l = c->L; l >= 0; --l) {
            const int k = c->kl[l], n = 2 * l + 1;
            int M, Nn;
            if (op == PG_X || op == PG_WARM) { M = n; Nn = k; }
            else if (op == PG_U) { M = k; Nn = n; }
            else { M = c->N; Nn = n; }
            if (op != PG_APPLY && !c->active[l]) continue;
            if (op == PG_APPLY && !c->used[l]) continue;        // in place: unused orders stay as they are
            for (int tm = 0; tm < div_up(M, tm_edge); ++tm)
                for (int tn = 0; tn < div_up(Nn, tn_edge); ++tn) t.push_back(l | (tm << 8) | (tn << 16));
        }
        if (t.empty()) t.push_back(0);
        c->n_pg_tiles[op] = (int)t.size();
        if (hipMalloc((void**)&c->d_pg_tiles[op], t.size() * sizeof(int)) != hipSuccess) return MTIP_ENOMEM;
        (void)mtip_copy(c, c->d_pg_tiles[op], t.data(), t.size() * sizeof(int), hipMemcpyHostToDevice);
    }
    // the fused pairs: 4 = k_proj_xw (order | row tile << 8, active orders), 5 = k_proj_ua (order | column tile << 8, used orders)
    for (int op = 4; op < 6; ++op) {
        std::vector<int> t;
        for (int l = c->L; l >= 0; --l) {
            if (op == 4 ? !c->active[l] : !(c->used[l] || c->active[l])) continue;
            for (int tt = 0; tt < div_up(2 * l + 1, 16); ++tt) t.push_back(l | (tt << 8));
        }
        c->n_pg_tiles[op] = (int)t.size();
        if (t.empty()) t.push_back(0);
        if (hipMalloc((void**)&c->d_pg_tiles[op], t.size() * sizeof(int)) != hipSuccess) return MTIP_ENOMEM;
        (void)mtip_copy(c, c->d_pg_tiles[op], t.data(), t.size() * sizeof(int), hipMemcpyHostToDevice);
    }
    return MTIP_OK;
}

template <int OP>
static void launch_proj_gemm(mtip_ctx* c, const ProjGemmArgs& a) {
    hipLaunchKernelGGL(k_proj_mfma<OP>, dim3((unsigned)c->n_pg_tiles[OP], (unsigned)c->B), dim3(256), 0, c->stream, a,
                       (const int*)c->d_pg_tiles[OP]);
}

// Divide-and-conquer pairing schedule for every column count 2..kmax (see jl_sweep_resident).  Entry
// [off[ke] + round * ps + group] = resident | mover << 8 | JS_WB | JS_ACTIVE (compact column indices).
// pair-groups a set of s columns needs: residents = the smaller half
static int js_groups(int s) {
    if (s < 2) return 0;
    const int a = s / 2;
    return std::max(a, js_groups(a) + js_groups(s - a));
}

// cols -> halves A (floor(s/2) residents, one group each) and B (movers); |B| rounds: group i takes B_((i + r) mod |B|),
// so every group is busy in every round; then A and B side by side on disjoint groups (A's groups keep residents)
int jacobi_groups(int k) { return js_groups(k); }

static void js_build(const std::vector<int>& cols, int g0, int round0, int ps, std::vector<std::vector<int>>& rounds) {
    const int s = (int)cols.size();
    if (s < 2) return;
    const int a = s / 2, b = s - a;
    if ((int)rounds.size() < round0 + b) rounds.resize(round0 + b, std::vector<int>(ps, 0));
    for (int r = 0; r < b; ++r)
        for (int i = 0; i < a; ++i) rounds[round0 + r][g0 + i] = cols[i] | (cols[a + (i + r) % b] << 8) | JS_ACTIVE;
    js_build(std::vector<int>(cols.begin(), cols.begin() + a), g0, round0 + b, ps, rounds);
    js_build(std::vector<int>(cols.begin() + a, cols.end()), g0 + js_groups(a), round0 + b, ps, rounds);
}

int build_jacobi_schedule(mtip_ctx* c, int kmax) {
    if (c->d_jsched != nullptr && c->jsched_kmax >= kmax) return MTIP_OK;
    int ps = 1;
    for (int ke = 2; ke <= kmax; ++ke) ps = std::max(ps, js_groups(ke));
    std::vector<int> all, off(kmax + 1, 0), nrd(kmax + 1, 0);
    for (int ke = 2; ke <= kmax; ++ke) {
        std::vector<int> cols(ke);
        for (int i = 0; i < ke; ++i) cols[i] = i;
        std::vector<std::vector<int>> rounds;
        js_build(cols, 0, 0, ps, rounds);
        // verification: no column twice in a round, every pair exactly once per sweep
        std::vector<char> met((size_t)ke * ke, 0);
        for (size_t r = 0; r < rounds.size(); ++r) {
            std::vector<char> used(ke, 0);
            for (int g = 0; g < ps; ++g) {
                const int e = rounds[r][g];
                if (!(e & JS_ACTIVE)) continue;
                const int x = e & 255, y = (e >> 8) & 255;
                if (x >= ke || y >= ke || x == y || used[x] || used[y] || met[(size_t)x * ke + y]) return MTIP_EINVAL;
                used[x] = used[y] = 1;
                met[(size_t)x * ke + y] = met[(size_t)y * ke + x] = 1;
            }
        }
        for (int x = 0; x < ke; ++x)
            for (int y = 0; y < ke; ++y)
                if (x != y && !met[(size_t)x * ke + y]) return MTIP_EINVAL;
        // write-back flag: the group does not keep this resident in its next active round, or the column is used by
        // anybody (this group as mover included) before that
        for (size_t r = 0; r < rounds.size(); ++r)
            for (int g = 0; g < ps; ++g) {
                int& e = rounds[r][g];
                if (!(e & JS_ACTIVE)) continue;
                const int res = e & 255;
                bool keep = false;
                for (size_t r2 = r + 1; r2 < rounds.size(); ++r2) {
                    bool elsewhere = false;
                    for (int g2 = 0; g2 < ps; ++g2) {
                        const int e2 = rounds[r2][g2];
                        if (!(e2 & JS_ACTIVE)) continue;
                        const int x = e2 & 255, y = (e2 >> 8) & 255;
                        if (y == res || (x == res && g2 != g)) elsewhere = true;
                    }
                    if (elsewhere) break;
                    const int e2 = rounds[r2][g];
                    if (e2 & JS_ACTIVE) {
                        keep = (e2 & 255) == res;
                        break;
                    }
                }
                if (!keep) e |= JS_WB;
            }
        off[ke] = (int)all.size();
        nrd[ke] = (int)rounds.size();
        for (auto& rd : rounds) all.insert(all.end(), rd.begin(), rd.end());
    }
    if (all.empty()) all.push_back(0);
    if (c->d_jsched) { (void)hipFree(c->d_jsched); (void)hipFree(c->d_jsched_off); (void)hipFree(c->d_jsched_rounds); }
    if (hipMalloc((void**)&c->d_jsched, all.size() * sizeof(int)) != hipSuccess) return MTIP_ENOMEM;
    if (hipMalloc((void**)&c->d_jsched_off, off.size() * sizeof(int)) != hipSuccess) return MTIP_ENOMEM;
    if (hipMalloc((void**)&c->d_jsched_rounds, nrd.size() * sizeof(int)) != hipSuccess) return MTIP_ENOMEM;
    (void)mtip_copy(c, c->d_jsched, all.data(), all.size() * sizeof(int), hipMemcpyHostToDevice);
    (void)mtip_copy(c, c->d_jsched_off, off.data(), off.size() * sizeof(int), hipMemcpyHostToDevice);
    (void)mtip_copy(c, c->d_jsched_rounds, nrd.data(), nrd.size() * sizeof(int), hipMemcpyHostToDevice);
    c->jsched_kmax = kmax;
    c->jsched_ps = ps;
    c->jsched_nrd = nrd;                                         // host copy: rounds per sweep by column count
    c->jsched_off_h = off;
    return MTIP_OK;
}

static void fill_gemm_args(mtip_ctx* c, ProjGemmArgs& ga, const double2* Ilm) {
    ga.Ilm = Ilm; ga.V = c->d_V; ga.X = c->d_X; ga.Xw = c->d_U; ga.Vr = c->d_Vr; ga.dst = c->d_X;
    ga.q = c->d_q; ga.rmask = c->d_rmask; ga.kl = c->d_kl; ga.active = c->d_active; ga.used = c->d_used;
    ga.voff = c->d_voff; ga.xoff = c->d_xoff; ga.uoff = c->d_uoff;
    ga.N = c->N; ga.L = c->L; ga.xtot = c->xtot; ga.utot = c->utot; ga.nlm = c->nlm;
    ga.inv_sqrt_np = 1.0 / std::sqrt(c->n_particles);
}

// I'_l = V_l U_l (+ mask / l = 0 rules) with the U_l currently in c->d_U: fxs_Projections.py:832-849, 866-871
int launch_apply_unknowns(mtip_ctx* c, const double2* Ilm, double2* out) {
    ProjGemmArgs ga;
    fill_gemm_args(c, ga, Ilm);
    if (build_proj_tiles(c) != MTIP_OK) {
        c->err = "projection tile lists: out of device memory";
        return MTIP_ENOMEM;
    }
    if (out != Ilm)
        (void)hipMemcpyAsync(out, Ilm, (size_t)c->B * c->C * sizeof(double2), hipMemcpyDeviceToDevice, c->stream);
    ga.dst = out;
    launch_proj_gemm<PG_APPLY>(c, ga);
    return MTIP_OK;
}

// largest k_l over the orders the apply product touches (LDS of k_proj_ua)
static int kmax_used(const mtip_ctx* c) {
    int k = 1;
    for (int l = 0; l <= c->L; ++l)
        if (c->used[l] || c->active[l]) k = std::max(k, c->kl[l]);
    return k;
}

// SO_freedom (fxs_Projections.py:768-780): `u_SO[4, 2] = u_SO[4, 2].real` on the unknowns of the chosen order after the polar
// factors, before they are applied -- here as a correction after the fact, the same for the real and the complex kernels: with
// y = Im U[4][2], U[4][2] -= i y and I'_l(q)[2] -= V_l[q][4] i y on the masked shells (column 2 of the order is m = 2 - l).
__global__ void __launch_bounds__(256) k_so_freedom(double2* __restrict__ U, double2* __restrict__ coef, const double2* __restrict__ V,
                                                    const uint8_t* __restrict__ rmask, int l, int kmax, int N, int nlm, int xtot,
                                                    int xoff_l, int voff_l) {
    const int b = blockIdx.x, n = 2 * l + 1;
    double2* u = U + (size_t)b * xtot + xoff_l + 4 * n + 2;
    const double y = u->y;
    __syncthreads();
    if (threadIdx.x == 0) u->y = 0.0;
    for (int q = threadIdx.x; q < N; q += blockDim.x) {
        if (!rmask[(size_t)l * N + q]) continue;
        const double2 v = V[(size_t)voff_l + (size_t)q * kmax + 4];
        double2* cq = coef + ((size_t)b * N + q) * nlm + l * l + 2;
        cq->x += v.y * y;                                         // - (v.x + i v.y) (i y)
        cq->y -= v.x * y;
    }
}

static int launch_project_coefficients_impl(mtip_ctx* c, const double2* Ilm, double2* out, bool real_intensity);

int launch_project_coefficients(mtip_ctx* c, const double2* Ilm, double2* out, bool real_intensity) {
    const int rc = launch_project_coefficients_impl(c, Ilm, out, real_intensity);
    if (rc == MTIP_OK && c->so_order >= 0 && c->active[c->so_order] && c->kl[c->so_order] >= 5) {
        const int l = c->so_order;
        hipLaunchKernelGGL(k_so_freedom, dim3((unsigned)c->B), dim3(256), 0, c->stream, c->d_U, out, (const double2*)c->d_V,
                           (const uint8_t*)c->d_rmask, l, std::min(2 * l + 1, c->N), c->N, c->nlm, c->xtot, c->xoff[l], c->voff[l]);
    }
    return rc;
}

static int launch_project_coefficients_impl(mtip_ctx* c, const double2* Ilm, double2* out, bool real_intensity) {
    ProfScope ps(c, "proj");
    if (real_intensity && rproj_supported(c)) {
        // real V_l and coefficients of a real intensity: the whole projection is one kernel in real arithmetic (k_projr.hip)
        if (out != Ilm)
            (void)hipMemcpyAsync(out, Ilm, (size_t)c->B * c->C * sizeof(double2), hipMemcpyDeviceToDevice, c->stream);
        c->vr_valid = false;
        return launch_rproj(c, out);
    }
    // general case (complex V_l, or coefficients without the symmetry of a real intensity): complex one-sided Jacobi between
    // the fused product pairs
    c->vr_kind = 0;
    int kmax = 1, nmax = 1;
    for (int l = 0; l <= c->L; ++l) {
        if (!c->active[l]) continue;
        kmax = std::max(kmax, c->kl[l]);
        nmax = std::max(nmax, 2 * l + 1);
    }
    ProjGemmArgs ga;
    fill_gemm_args(c, ga, Ilm);
    if (build_proj_tiles(c) != MTIP_OK) {
        c->err = "projection tile lists: out of device memory";
        return MTIP_ENOMEM;
    }
    // fused pairs (k_proj_xw, k_proj_ua) when X_l fits the 8 waves of k_proj_xw
    const bool fuse = c->proj_fuse && kmax <= 16 * (PF_THREADS / 64);
    const size_t lds = ((size_t)kmax * (nmax | 1) + (size_t)kmax * (kmax | 1)) * sizeof(double2);   // unpadded minimum
    const bool sched_ok = c->jac_resident && kmax <= 255 && kmax >= 2 && build_jacobi_schedule(c, kmax) == MTIP_OK;
    const bool lds_path = lds <= 158 * 1024;                    // X_l and V_r of the largest order share one CU's LDS (2l+1 <= 71)
    // cold start every 64 calls bounds the accumulated rounding drift of the carried V_r
    const int warm = (lds_path && c->vr_valid && (c->proj_calls % 64) != 0) ? 1 : 0;
    if (!(fuse && warm)) launch_proj_gemm<PG_X>(c, ga);         // (fused warm start: X_l never leaves the workgroups of k_proj_xw)
    if (lds_path) {
        const double2* src = c->d_X;
        if (warm) {
            ga.dst = c->d_U;
            if (fuse)                                           // X_l rows stay in LDS, W = X_l V_r straight into d_U
                hipLaunchKernelGGL(k_proj_xw, dim3((unsigned)std::max(c->n_pg_tiles[4], 1), (unsigned)c->B), dim3(PF_THREADS),
                                   (size_t)16 * (kmax | 1) * sizeof(double2), c->stream, ga, (const int*)c->d_pg_tiles[4]);
            else
                launch_proj_gemm<PG_WARM>(c, ga);
            src = c->d_U;
        }
        const int pairs_max = kmax / 2;                         // valid pairs per round (odd k: dummy pair skipped)
        // 16 lanes per pair when all pairs of a round still fit one workgroup, else 8
        const int tg = (c->jac_tg == 16 && pairs_max * 16 <= JL_MAX_THREADS && nmax <= 5 * 16) ? 16 : 8;
        const size_t lds_pad = ((size_t)kmax * (div_up(nmax, tg) * tg + 1) + (size_t)kmax * (div_up(kmax, tg) * tg + 1)) * sizeof(double2);
        const int pad = lds_pad <= 158 * 1024 ? 1 : 0;
        const size_t lds_use = (pad ? lds_pad : lds) + 16 * sizeof(double2);
        const bool use_sched = tg == 16 && sched_ok && c->jsched_ps * 16 <= JL_MAX_THREADS;
        int threads = (((use_sched ? c->jsched_ps : pairs_max) * tg + 63) / 64) * 64;
        threads = std::min(std::max(threads, 64), JL_MAX_THREADS);
        if (c->d_jorder == nullptr) {                           // active orders, heaviest (largest k_l) first
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
        // the launch's LDS must hold the padded (or tight) X_l and V_r of EVERY active order: they are sized by the largest
        // k_l and 2l+1 above, which bound every order's
        const dim3 gj((unsigned)c->B, (unsigned)std::max(c->n_jorder, 1));
        JacobiArgs ja;
        ja.Xin_all = src; ja.Pn_all = c->d_X; ja.Vr_all = c->d_Vr;
        ja.kl = c->d_kl; ja.active = c->d_active; ja.xoff = c->d_xoff; ja.roff = c->d_uoff;
        ja.xtot = c->xtot; ja.rtot = c->utot; ja.L = c->L; ja.warm = warm;
        ja.tabs2 = c->polar_abs_tol * c->polar_abs_tol;
        ja.sweeps_out = c->d_sweeps; ja.pad = pad;
        ja.sched = use_sched ? (const int*)c->d_jsched : (const int*)nullptr;
        ja.sched_off = c->d_jsched_off; ja.sched_rounds = c->d_jsched_rounds; ja.sched_ps = c->jsched_ps;
        ja.order_list = c->d_jorder;
        {
            ProfScope pp(c, "polar");                            // the polar-factor kernel alone (nested in "proj")
            if (tg == 16) hipLaunchKernelGGL((k_polar_jacobi_lds<5, 16, JL_MAX_THREADS>), gj, dim3(threads), lds_use, c->stream, ja);
            else if (nmax <= 9 * 8) hipLaunchKernelGGL((k_polar_jacobi_lds<9, 8, JL_MAX_THREADS>), gj, dim3(threads), lds_use, c->stream, ja);
            else hipLaunchKernelGGL((k_polar_jacobi_lds<16, 8, JL_MAX_THREADS>), gj, dim3(threads), lds_use, c->stream, ja);
        }
        c->vr_valid = true;
        c->proj_calls += 1;
        if (fuse) {
            // U_l = V_r Pn^+ by column blocks, each block at once the operand of its part of I'_l = V_l U_l
            if (out != Ilm)
                (void)hipMemcpyAsync(out, Ilm, (size_t)c->B * c->C * sizeof(double2), hipMemcpyDeviceToDevice, c->stream);
            ga.dst = out;
            hipLaunchKernelGGL(k_proj_ua, dim3((unsigned)std::max(c->n_pg_tiles[5], 1), (unsigned)c->B), dim3(PF_THREADS),
                               (size_t)kmax_used(c) * 17 * sizeof(double2), c->stream, ga, c->d_U, (const int*)c->d_pg_tiles[5]);
            return MTIP_OK;
        }
        ga.dst = c->d_U;
        launch_proj_gemm<PG_U>(c, ga);
    } else {
        // X_l and V_r do not share a CU's LDS (complex 2l+1 > 71): global-memory Jacobi, cold start
        ProfScope pp(c, "polar");
        hipLaunchKernelGGL(k_polar_jacobi, dim3((unsigned)(c->L + 1), (unsigned)c->B), dim3(256), 0, c->stream, c->d_X,
                           c->d_Vr, c->d_U, (const int*)c->d_kl, (const int*)c->d_active, (const int*)c->d_xoff,
                           (const int*)c->d_uoff, c->xtot, c->utot);
        c->vr_valid = false;
        c->proj_calls += 1;
    }
    // I'_l = V_l U_l in place on the coefficient buffer; a separate output first receives a copy of I_l
    if (out != Ilm)
        (void)hipMemcpyAsync(out, Ilm, (size_t)c->B * c->C * sizeof(double2), hipMemcpyDeviceToDevice, c->stream);
    ga.dst = out;
    launch_proj_gemm<PG_APPLY>(c, ga);
    return MTIP_OK;
}

// ---- B_l = I_l I_l^+ ------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_deg2(const double2* __restrict__ Ilm, double2* __restrict__ Bl, int N, int L) {
    const int l = blockIdx.y, b = blockIdx.z;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * N) return;
    const int nlm = (L + 1) * (L + 1);
    const int i = e / N, j = e - i * N;
    const double2* Ii = Ilm + ((size_t)b * N + i) * nlm + l * l;
    const double2* Ij = Ilm + ((size_t)b * N + j) * nlm + l * l;
    double2 acc = make_double2(0.0, 0.0);
    for (int m = 0; m < 2 * l + 1; ++m) acc = cadd(acc, cmulc(Ii[m], Ij[m]));
    Bl[(((size_t)b * (L + 1) + l) * N + i) * N + j] = acc;
}

// B_l = I_l I_l^+ on the f64 matrix cores (this IS a GEMM: N x (2l+1) x N per (restart, l)).  With the coefficients read
// as real rows [re, im, re, im, ...] of length K = 2 (2l+1):  Re B[q][q'] = sum_K A[q][K] A[q'][K]  and
// Im B[q][q'] = sum_K A'[q][K] A[q'][K]  with A'[(m,re)] = Im, A'[(m,im)] = -Re -- the neighbouring K index, i.e. the
// lane 16 further (v_mfma_f64_16x16x4: A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15], D reg r =
// D[(lane>>4) + 4r][lane&15]).  A wave owns one 16 x 16 output tile (two accumulators), the four waves of a workgroup
// share the row tile.  METRIC: instead of storing B_l the tile adds up |ref rs - mask B|^2 (deg2_invariant_l2_diff,
// fxs_IO_methods.py:408-447) into a per-tile partial sum; k_deg2_metric_finish adds the tiles in a fixed order.
template <bool METRIC>
__global__ void __launch_bounds__(256) k_deg2_mfma(const double* __restrict__ Ilm, double2* __restrict__ Bl,
                                                   const double2* __restrict__ Bref, const uint8_t* __restrict__ rmask,
                                                   const int* __restrict__ used, double* __restrict__ part, int N, int L,
                                                   double inv_np) {
    const int l = blockIdx.y, b = blockIdx.z;
    const int nt16 = (N + 15) / 16, ng = (nt16 + 3) / 4;
    const int ti = blockIdx.x / ng, tg = blockIdx.x - ti * ng;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tj = tg * 4 + wave;
    double* my_part = METRIC ? part + (((size_t)b * (L + 1) + l) * nt16 + ti) * nt16 + tj : nullptr;
    if (tj >= nt16) return;                                      // wave-uniform
    if (METRIC && !used[l]) {
        if (lane == 0) *my_part = 0.0;
        return;
    }
    const int nlm2 = 2 * (L + 1) * (L + 1);
    const int K = 2 * (2 * l + 1);
    const int li = lane & 15, kk = lane >> 4;
    const int qi = ti * 16 + li, qj = tj * 16 + li;
    const double* ai = Ilm + ((size_t)b * N + (qi < N ? qi : 0)) * nlm2 + 2 * l * l;
    const double* aj = Ilm + ((size_t)b * N + (qj < N ? qj : 0)) * nlm2 + 2 * l * l;
    v4f64 acc_re = v4f64{0.0, 0.0, 0.0, 0.0}, acc_im = acc_re;
    for (int k0 = 0; k0 < K; k0 += 4) {
        const int kx = k0 + kk;
        const double a = (kx < K && qi < N) ? ai[kx] : 0.0;
        const double bv = (kx < K && qj < N) ? aj[kx] : 0.0;
        const double partner = __shfl_xor(a, 16, 64);            // the other half (re <-> im) of the same m
        const double ap = (kk & 1) ? -partner : partner;
        acc_re = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc_re, 0, 0, 0);
        acc_im = __builtin_amdgcn_mfma_f64_16x16x4f64(ap, bv, acc_im, 0, 0, 0);
    }
    const int col = tj * 16 + li;
    double sum = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = ti * 16 + kk + 4 * r;
        if (row < N && col < N) {
            if (!METRIC) {
                Bl[(((size_t)b * (L + 1) + l) * N + row) * N + col] = make_double2(acc_re[r], acc_im[r]);
            } else {
                const bool m = rmask[(size_t)l * N + row] && rmask[(size_t)l * N + col];
                const double rs = (l == 0) ? inv_np : 1.0;
                const double2 ref = Bref[((size_t)l * N + row) * N + col];
                const double dx = ref.x * rs - (m ? acc_re[r] : 0.0), dy = ref.y * rs - (m ? acc_im[r] : 0.0);
                sum += dx * dx + dy * dy;
            }
        }
    }
    if (METRIC) {
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        if (lane == 0) *my_part = sum;
    }
}

// one workgroup per (order, restart): the tile sums in a fixed order, then / norm
__global__ void __launch_bounds__(256) k_deg2_metric_finish(const double* __restrict__ part, const double* __restrict__ Bnorm,
                                                            const int* __restrict__ used, double* __restrict__ out, int n_tiles,
                                                            int L) {
    __shared__ double red[256];
    const int l = blockIdx.x, b = blockIdx.y;
    const double* p = part + ((size_t)b * (L + 1) + l) * n_tiles;
    double acc = 0.0;
    for (int e = threadIdx.x; e < n_tiles; e += blockDim.x) acc += p[e];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double nrm = Bnorm[l];
        out[(size_t)b * (L + 1) + l] = (used[l] && nrm != 0.0) ? red[0] / nrm : -1.0;
    }
}

void launch_deg2(mtip_ctx* c, const double2* Ilm, double2* Bl) {
    ProfScope ps(c, "deg2");
    if (!c->deg2_simple) {
        const int nt16 = div_up(c->N, 16);
        hipLaunchKernelGGL(k_deg2_mfma<false>, dim3((unsigned)(nt16 * div_up(nt16, 4)), (unsigned)(c->L + 1), (unsigned)c->B),
                           dim3(256), 0, c->stream, reinterpret_cast<const double*>(Ilm), Bl, (const double2*)nullptr,
                           (const uint8_t*)nullptr, (const int*)nullptr, (double*)nullptr, c->N, c->L, 1.0);
        return;
    }
    hipLaunchKernelGGL(k_deg2, dim3((unsigned)div_up((long long)c->N * c->N, 256), (unsigned)(c->L + 1), (unsigned)c->B),
                       dim3(256), 0, c->stream, Ilm, Bl, c->N, c->L);
}

// deg2_invariant_l2_diff: sum |ref - mask*B|^2 / norm per order (fxs_IO_methods.py:408-447)
__global__ void __launch_bounds__(256) k_deg2_metric(const double2* __restrict__ Ilm, const double2* __restrict__ Bref,
                                                     const double* __restrict__ Bnorm, const uint8_t* __restrict__ rmask,
                                                     const int* __restrict__ used, double* __restrict__ out, int N, int L,
                                                     double inv_np) {
    __shared__ double red[256];
    const int l = blockIdx.x, b = blockIdx.y;
    const int nlm = (L + 1) * (L + 1);
    double acc = 0.0;
    if (used[l]) {
        const double rs = (l == 0) ? inv_np : 1.0;
        for (int e = threadIdx.x; e < N * N; e += blockDim.x) {
            const int i = e / N, j = e - i * N;
            double2 bij = make_double2(0.0, 0.0);
            if (rmask[(size_t)l * N + i] && rmask[(size_t)l * N + j]) {
                const double2* Ii = Ilm + ((size_t)b * N + i) * nlm + l * l;
                const double2* Ij = Ilm + ((size_t)b * N + j) * nlm + l * l;
                for (int m = 0; m < 2 * l + 1; ++m) bij = cadd(bij, cmulc(Ii[m], Ij[m]));
            }
            const double2 r = Bref[((size_t)l * N + i) * N + j];
            const double dx = r.x * rs - bij.x, dy = r.y * rs - bij.y;
            acc += dx * dx + dy * dy;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double nrm = Bnorm[l];
        out[(size_t)b * (L + 1) + l] = (used[l] && nrm != 0.0) ? red[0] / nrm : -1.0;
    }
}

void launch_deg2_metric(mtip_ctx* c, const double2* Ilm, double* out) {
    ProfScope ps(c, "deg2_metric");
    if (!c->deg2_simple && c->d_deg2_part != nullptr) {
        const int nt16 = div_up(c->N, 16);
        hipLaunchKernelGGL(k_deg2_mfma<true>, dim3((unsigned)(nt16 * div_up(nt16, 4)), (unsigned)(c->L + 1), (unsigned)c->B),
                           dim3(256), 0, c->stream, reinterpret_cast<const double*>(Ilm), (double2*)nullptr,
                           (const double2*)c->d_Bref, (const uint8_t*)c->d_rmask, (const int*)c->d_used, c->d_deg2_part, c->N,
                           c->L, 1.0 / c->n_particles);
        hipLaunchKernelGGL(k_deg2_metric_finish, dim3((unsigned)(c->L + 1), (unsigned)c->B), dim3(256), 0, c->stream,
                           (const double*)c->d_deg2_part, (const double*)c->d_Bnorm, (const int*)c->d_used, out, nt16 * nt16,
                           c->L);
        return;
    }
    hipLaunchKernelGGL(k_deg2_metric, dim3((unsigned)(c->L + 1), (unsigned)c->B), dim3(256), 0, c->stream, Ilm,
                       (const double2*)c->d_Bref, (const double*)c->d_Bnorm, (const uint8_t*)c->d_rmask,
                       (const int*)c->d_used, out, c->N, c->L, 1.0 / c->n_particles);
}
