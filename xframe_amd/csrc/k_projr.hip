// Reciprocal-space projection for REAL projection matrices (rows a8, a9 of SURVEY section 8), one kernel per call.
//   approximate_unknowns  xframe/projects/fxs/projectLibrary/fxs_Projections.py:752-767   U_l = u @ vh of svd(V_l^+ D^2 I_l)
//   mtip_projection       fxs_Projections.py:832-849, 866-871                             I'_l[mask] = (V_l U_l)[mask], l = 0 rules
//
// B_l(q,q') = sum_m I_lm(q) conj(I_lm(q')) of a real intensity is real, so V_l = eigvecs sqrt(eigvals) of it is real
// (fxs_invariant_tools.py:1114-1141: scipy eigh, real eigenvectors for a matrix without imaginary part; stored
// `astype(complex)`, 1207).  The reference types B_l complex: from cross-correlation data (extract.py:136, real data through
// real Legendre matrices, 460-515) its imaginary part is exactly zero and V_l exactly real; on the `density` route
// (extract.py:288, `Il @ Il.T.conj()`) rounding residue remains (measured with the reference here: <= 1e-17 |B_l| at even l,
// up to 1e-7 max|V_l| in the null-space columns of V_l) -- the host takes this kernel only when Im V_l == 0 exactly for
// every used order (or below MTIP_PROJ_REAL_TOL when that opt-in is set), otherwise the general kernels of k_proj.hip.
// I_lm are the coefficients of the REAL intensity |F|^2, so I_{l,-m} = (-1)^m conj(I_{l,m}).  Then M = V_l^T D^2 I_l (k x (2l+1), complex) is unitarily equivalent
// to a REAL matrix: with the unitary T that maps the column pair (m, -m) to (sqrt2 Re, sqrt2 Im) of column m,
//   M~ = M T,   M~[:, rho'] = sum_q V[q, :] q^2 I~[q, rho'],   I~[q, .] = (Re I_l0, 0, sqrt2 Re I_l1, sqrt2 Im I_l1, ...)
// (rho' = 2m + part; the slot rho' = 1, "Im I_l0", is identically zero and kept so that (Re, Im) pairs sit in lane pairs),
// and polar(M) = polar(M~) T^+ exactly: the polar factor is unique and T unitary.  Everything is real arithmetic:
// a third of the flops of the complex one-sided Jacobi and half its LDS, and the four products around it are real GEMMs.
//
// One workgroup per (restart, slot); a slot is a short list of orders solved one after the other (host-packed so that every
// slot costs about as much as the largest order alone: 9 workgroups per restart at L = 32 instead of 18 CU-filling ones).
// Per order, all in LDS:
//   A  X~^T[j][rho'] = sum_q (q^2 V)[q][j] I~[q][rho']         f64 MFMA, fragments straight from L2
//   W  warm start: X~ <- X~ V_r(previous step)                  f64 MFMA from LDS (rows of X~ transform independently)
//   J  one-sided Jacobi on the columns of X~ (n' x k) and V_r  (resident-column ordering, schedule of k_proj.hip)
//   U  U~^T[rho'][i] = sum_c X~[rho'][c] / sigma_c V_r[i][c]    (-> complex U_l for the unknowns output)
//   E  I~'[q][rho'] = sum_i V[q][i] U~^T[rho'][i] -> I'_{l,+-m}(q) on the masked shells, in place on the coefficients
#include "mtip_internal.h"
#include "k_jacobi.h"

#define RP_MAX_THREADS 1024
// The last sweep of the classic loop only confirms convergence: its rotations are below JL_EARLY and it takes the columns from
// ~1e-6 of non-orthogonality to ~1e-12.  With the padded layout the loop stops one level earlier -- after a sweep whose largest
// |gamma| / sqrt(alpha beta) stayed below RP_EARLY_CORR, which by quadratic convergence leaves E ~ RP_EARLY_CORR^2 -- and the rest is
// done by ONE first-order step on the matrix pipe: with G = W^T W = D (1 + E) D (D = diag sigma), the polar factor of W is
// W G^-1/2 and G^1/2 = D + Delta + O(E^2), Delta_ij = G_ij / (sigma_i + sigma_j) (the Sylvester equation D Delta + Delta D = G - D^2
// is diagonal in this basis; no division by sigma_i - sigma_j: clustered spectra are harmless), so
//   W G^-1/2 = W D^-1 (1 - Delta D^-1) + O(E^2),   (D^-1 (1 - Delta D^-1))_ij = delta_ij / sigma_i - G_ij / (sigma_i sigma_j (sigma_i + sigma_j)),
// a symmetric k x k matrix: one Gram product and one k x k x n' product (about a third of a sweep at k = 65) leave E^2 <= 1e-11.
// The Gram matrix is also the check: if its largest |E_ij| is above RP_CORR_MAX the sweeps go on with the classic criterion.
// Second order (taken when the Gram matrix shows RP_CORR_MAX < max |E_ij| <= RP_CORR2_MAX), with A = D^-1 Delta D^-1 and A~ = A D:
//   G^-1/2 = D^-1 - A + (A~ A~^T) o S + A~ A + O(E^3),   S_ij = 1 / (sigma_i + sigma_j)
// (the second Newton correction of the square root, Delta2 = -(Delta^2) o S, and the second term of the Neumann series of the
// inverse): two more k x k x k products, measured error 0.3 E^3 -- so the sweeps may stop after one with rotations up to
// RP_EARLY_CORR = 3e-2, which leaves E ~ 1e-6 .. 1e-4 (when the Gram matrix says more, one more sweep runs and the check repeats;
// measured 1e-2 / 3e-2 / 1e-1: 365 / 352 / 348 us per projection in the HIO blocks).
#define RP_EARLY_CORR 3e-2
#define RP_CORR_MAX 3e-6
#define RP_CORR_SKIP 1e-12    // below this the plain finish (error ~ |E|) is as good as the step: no step needed
#define RP_CORR2_MAX 1.5e-4
#define RP_SLACK 128            // doubles behind the matrices: predicated-off lanes of the last row slot still form addresses
#define RP_ACC_MAX 5            // 16 x 16 output tiles a wave works on at a time (and holds across a barrier: in-place products)

struct RpShared {
    double gmax[RP_MAX_THREADS / 16];
    double isig[128];
    double sig[128];
    int perm[128];
    int cont, keff;
    double red[RP_MAX_THREADS / 64];
    int pad_;
};

// rotation [a b] <- [a b] [[c, w], [-w, c]] that annihilates gamma = a.b (smaller angle); false: already orthogonal
__device__ __forceinline__ bool rp_params(double alpha, double beta, double g, bool valid, double tabs2, double S, double early2, bool& big,
                                          double& cs, double& w) {
    const double g2 = g * g;
    const double ab = alpha * beta;
    if (!(valid && g2 > (JAC_TOL * JAC_TOL) * ab && g2 > tabs2 * fmax(alpha, beta) * S && g2 > 0.0)) return false;
    big = big || (g2 > early2 * ab);
    // d = (beta - alpha)/2, h = sqrt(d^2 + g^2):  c^2 = (1 + |d|/h)/2,  w = sign(d) g / (2 h c)
    const double d = 0.5 * (beta - alpha);
    const double ih = fast_rsqrt(fma(d, d, g2));
    const double c2 = fma(0.5 * fabs(d), ih, 0.5);
    const double rc = fast_rsqrt(c2);
    cs = c2 * rc;
    w = ((d >= 0.0 ? 0.5 : -0.5) * ih * rc) * g;
    return true;
}

// One sweep in the resident-column ordering (see jl_sweep_resident in k_proj.hip): 16 lanes per column pair, NR row slots per
// lane for X~ and for V_r; the resident column of a group stays in registers over a phase, the mover goes through LDS.
// TIMED (diagnostic instance, mtip_debug_polar_timing): cycles of the segments of a round summed into tacc[0..4]
// (operands arrived | Gram sums + lane sums | rotation parameters | rotations + stores | barrier)
template <int NR, bool TIMED = false>
__device__ __forceinline__ void rp_sweep(double* Xs, double* Vs, int ns, int ks, int t, int group, const int* tab, int n_rounds,
                                         int ps, const int* s_perm, bool xl_ok, bool vl_ok, double tabs2, double S, double early2, bool& big,
                                         long long* tacc = nullptr) {
    long long tq = 0;
#define RP_SEG(I)                            \
    if (TIMED) {                             \
        const long long tn_ = clock64();     \
        tacc[I] += tn_ - tq;                 \
        tq = tn_;                            \
    }
    double rx[NR], rv[NR];
#pragma unroll
    for (int u = 0; u < NR; ++u) {
        rx[u] = 0.0;
        rv[u] = 0.0;
    }
    int cur = -1;
    bool dirty = false;
    int e_next = (group < ps && n_rounds > 0) ? tab[group] : 0;
    int pr_next = s_perm[(e_next & JS_ACTIVE) ? (e_next & 255) : 0], pm_next = s_perm[(e_next & JS_ACTIVE) ? ((e_next >> 8) & 255) : 0];
    if (TIMED) tq = clock64();
    for (int r = 0; r < n_rounds; ++r) {
        const int e = e_next;
        const int pr = pr_next, pm = pm_next;
        if (r + 1 < n_rounds && group < ps) e_next = tab[(size_t)(r + 1) * ps + group];
        const bool act = (e & JS_ACTIVE) != 0;
        const int res = act ? (e & 255) : 0;
        double* xh = Xs + (size_t)pr * ns + t;
        double* vh = Vs + (size_t)pr * ks + t;
        double* xm = Xs + (size_t)pm * ns + t;
        double* vm = Vs + (size_t)pm * ks + t;
        double mx[NR], mv[NR];
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            mx[u] = 0.0;
            mv[u] = 0.0;
        }
        if (act) {
#pragma unroll
            for (int u = 0; u < NR; ++u) {
                mx[u] = xm[u * 16];
                mv[u] = vm[u * 16];
            }
            if (res != cur) {
#pragma unroll
                for (int u = 0; u < NR; ++u) {
                    rx[u] = xh[u * 16];
                    rv[u] = vh[u * 16];
                }
                if (!xl_ok) rx[NR - 1] = 0.0;
                cur = res;
                dirty = false;
            }
            if (!xl_ok) mx[NR - 1] = 0.0;
        }
        if (TIMED) {
            double sink = 0.0;
#pragma unroll
            for (int u = 0; u < NR; ++u) sink += mx[u] + mv[u] + rx[u];
            asm volatile("" ::"v"(sink));                      // operands have arrived
        }
        RP_SEG(0)
        double alpha = 0.0, beta = 0.0, g = 0.0, zero = 0.0;
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            alpha = fma(rx[u], rx[u], alpha);
            beta = fma(mx[u], mx[u], beta);
            g = fma(rx[u], mx[u], g);
        }
        group_sum4<16>(alpha, beta, g, zero);          // (run by every group, active or not: uniform control flow around DPP)
        if (TIMED) asm volatile("" ::"v"(alpha), "v"(beta), "v"(g));
        RP_SEG(1)
        double cs = 1.0, w = 0.0;
        const bool rot = rp_params(alpha, beta, g, act, tabs2, S, early2, big, cs, w);
        if (TIMED) asm volatile("" ::"v"(cs), "v"(w));
        RP_SEG(2)
        if (rot) {
#pragma unroll
            for (int u = 0; u < NR; ++u) {
                const double a = rx[u], bq = mx[u];
                rx[u] = fma(-w, bq, cs * a);
                const double bn = fma(w, a, cs * bq);
                if (u < NR - 1 || xl_ok) xm[u * 16] = bn;
            }
#pragma unroll
            for (int u = 0; u < NR; ++u) {
                const double a = rv[u], bq = mv[u];
                rv[u] = fma(-w, bq, cs * a);
                const double bn = fma(w, a, cs * bq);
                if (u < NR - 1 || vl_ok) vm[u * 16] = bn;
            }
            dirty = true;
        }
        if (act && (e & JS_WB)) {                      // someone else takes this column next round
            if (dirty) {
#pragma unroll
                for (int u = 0; u < NR; ++u) {
                    if (u < NR - 1 || xl_ok) xh[u * 16] = rx[u];
                    if (u < NR - 1 || vl_ok) vh[u * 16] = rv[u];
                }
            }
            cur = -1;
        }
        pr_next = s_perm[(e_next & JS_ACTIVE) ? (e_next & 255) : 0];
        pm_next = s_perm[(e_next & JS_ACTIVE) ? ((e_next >> 8) & 255) : 0];
        RP_SEG(3)
        __syncthreads();
        RP_SEG(4)
    }
#undef RP_SEG
}

// ---- the same sweep without branches: zero-padded columns ---------------------------------------------------------------------
// A round of rp_sweep is bound by the NUMBER of instructions a wave issues (one per ~4 cycles, whatever their kind), and two
// thirds of them were not arithmetic: exec-mask regions around every predicate (inactive group, last row slot, rotation or not,
// write-back), zero initialisations, address arithmetic through the column permutation.  Up to 5 row slots (k <= 79) the matrices
// are laid out so that none of that is needed:
//   * X~ and V_r share one column stride NS = 16 NR + 1 and V_r sits a compile-time distance behind X~ (16 NR columns): every
//     access of a round is `ds_read / ds_write base + immediate`, base = column offset from the table + lane;
//   * rows beyond the matrix are zero and stay zero under rotations: no predicate on the last row slot;
//   * column 16 NR - 1 is all zero: the groups without a pair in a round "rotate" it with itself (identity, zeros written back);
//   * the pairing table is translated once per sweep into {resident offset << 2 | write-back << 1 | load, mover offset} (through
//     the compaction of the deflated columns), flags made self-contained: a resident is written back before an idle round;
//   * a pair that is orthogonal already takes the identity rotation through the same instructions.
// nc = blocks of 16 columns (= row slots of 16 rows: 2l+2 <= 16 nc); TG = lanes per column pair (16; a 32-lane variant -- twice
// the waves per round with shorter chains each -- measured 3520 against 2440 ticks per round and was removed in round 4)
#define RP_PAD_MAX_NR 5
__host__ __device__ __forceinline__ constexpr int rp_pad_nrt(int nc, int tg) { return (16 * nc + tg - 1) / tg; }      // row slots per lane
__host__ __device__ __forceinline__ constexpr int rp_pad_ns(int nc, int tg) { return rp_pad_nrt(nc, tg) * tg + 1; }   // column stride (odd)
__host__ __device__ __forceinline__ constexpr int rp_pad_voff(int nc, int tg) { return 16 * nc * rp_pad_ns(nc, tg); } // doubles from X~ to V_r

// sums over the TG lanes of a pair-group, every lane receives them
// The carve-up of a workgroup's dynamic LDS for one order (k columns, n2 = 2l+2 rows): the ONE definition the launcher sizes the
// block from and the kernel takes its pointers from (gfx950 drops stores beyond the allocation silently -- two rounds in a row a
// hand-kept second copy of this arithmetic went out of step).  Offsets in doubles from the start of the block.
struct RpLayout {
    bool pad;                 // zero-padded columns, one stride (up to RP_PAD_MAX_NR row slots); else tight columns
    int ns, ks;               // column strides of X~ and V_r (odd)
    size_t v_off;             // V_r
    size_t tab_off;           // padded layout: raw pairing table (ints), behind it the per-sweep translation (int2 per round and
                              // group); the Gram matrix / symmetric factor of the closing step overlays both.  Tight layout: end of
                              // the matrices (+ slack); the Gram matrix overlays V_r there
    size_t end_bytes;         // bytes of dynamic LDS this order needs
};
__host__ __device__ __forceinline__ RpLayout rp_layout(int k, int n2, int tg, int tab_ints, int tab2_entries) {
    RpLayout y;
    const int nc = (n2 + 15) >> 4;
    y.pad = nc <= RP_PAD_MAX_NR;
    y.ns = y.pad ? rp_pad_ns(nc, tg) : (n2 | 1);
    y.ks = y.pad ? y.ns : (k | 1);
    y.v_off = y.pad ? (size_t)rp_pad_voff(nc, tg) : (size_t)k * y.ns;
    y.tab_off = (y.pad ? 2 * (size_t)rp_pad_voff(nc, tg) : (size_t)k * y.ns + (size_t)k * y.ks) + RP_SLACK;
    size_t end = y.tab_off * sizeof(double);
    if (y.pad) {
        const size_t tables = end + (size_t)tab_ints * sizeof(int) + (size_t)tab2_entries * sizeof(int2);
        const size_t gram = end + (size_t)k * (k | 1) * sizeof(double);
        end = tables > gram ? tables : gram;
    }
    y.end_bytes = end;
    return y;
}
#define RP_LAYOUT_ERROR 0x7fffffff     // sweeps_out marker: the kernel's layout did not fit the launch's LDS (order skipped)

template <int TG>
__device__ __forceinline__ void rp_sum3(double& a, double& b, double& g) {
    static_assert(TG == 16, "16 lanes per column pair (the 32-lane variant measured slower and was removed)");
    double z = 0.0;
    group_sum4<16>(a, b, g, z);
}
template <int TG>
__device__ __forceinline__ double rp_sum1(double v) {
    v = group_sum<16>(v);
    return v;
}

template <int NC, int TG, bool TIMED = false>
__device__ __forceinline__ void rp_sweep_pad(double* xt /* X~ + lane of the group */, const int2* tab, int n_rounds, int ngroups,
                                             int group, double tabs2, double S, double early2, bool& big, long long* tacc = nullptr) {
    constexpr int NR = rp_pad_nrt(NC, TG);
    constexpr int VOFF = rp_pad_voff(NC, TG);
    long long tq = 0;
#define RP_SEG(I)                            \
    if (TIMED) {                             \
        const long long tn_ = clock64();     \
        tacc[I] += tn_ - tq;                 \
        tq = tn_;                            \
    }
    double rx[NR], rv[NR];
#pragma unroll
    for (int u = 0; u < NR; ++u) {
        rx[u] = 0.0;
        rv[u] = 0.0;
    }
    int2 e_next = tab[group];
    if (TIMED) tq = clock64();
    for (int r = 0; r < n_rounds; ++r) {
        const int2 e = e_next;
        if (r + 1 < n_rounds) e_next = tab[(r + 1) * ngroups + group];     // in flight during the round
        double* xr = xt + (e.x >> 2);
        double* xm = xt + e.y;
        double mx[NR], mv[NR];
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            mx[u] = xm[u * TG];
            mv[u] = xm[VOFF + u * TG];
        }
        if (e.x & 1) {                                           // a new resident for this group
#pragma unroll
            for (int u = 0; u < NR; ++u) {
                rx[u] = xr[u * TG];
                rv[u] = xr[VOFF + u * TG];
            }
        }
        if (TIMED) asm volatile("" : "+v"(mx[0]), "+v"(mx[NR - 1]), "+v"(mv[NR - 1]), "+v"(rx[NR - 1]));
        RP_SEG(0)
        double alpha = 0.0, beta = 0.0, g = 0.0;
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            alpha = fma(rx[u], rx[u], alpha);
            beta = fma(mx[u], mx[u], beta);
            g = fma(rx[u], mx[u], g);
        }
        if (TIMED) asm volatile("" : "+v"(alpha), "+v"(beta), "+v"(g));
        RP_SEG(1)
        rp_sum3<TG>(alpha, beta, g);
        if (TIMED) asm volatile("" : "+v"(alpha), "+v"(beta), "+v"(g));
        RP_SEG(2)
        // rotation [a b] <- [a b] [[c, w], [-w, c]] (see rp_params), the identity for a pair that is orthogonal already
        const double g2 = g * g, ab = alpha * beta;
        const bool rot = g2 > (JAC_TOL * JAC_TOL) * ab && g2 > tabs2 * fmax(alpha, beta) * S && g2 > 0.0;
        big = big || (rot && g2 > early2 * ab);
        const double d = 0.5 * (beta - alpha);
        const double h2 = fma(d, d, g2);
        const double ih = fast_rsqrt(rot ? h2 : 1.0);
        const double c2 = fma(0.5 * fabs(d), ih, 0.5);
        const double rc = fast_rsqrt(c2);
        const double cs = rot ? c2 * rc : 1.0;
        const double w = rot ? ((d >= 0.0 ? 0.5 : -0.5) * ih * rc) * g : 0.0;
        if (TIMED) asm volatile("" : "+v"(mx[0]) : "v"(cs), "v"(w));
        RP_SEG(3)
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const double a = rx[u], bq = mx[u];
            rx[u] = fma(-w, bq, cs * a);
            xm[u * TG] = fma(w, a, cs * bq);
        }
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const double a = rv[u], bq = mv[u];
            rv[u] = fma(-w, bq, cs * a);
            xm[VOFF + u * TG] = fma(w, a, cs * bq);
        }
        if (e.x & 2) {                                           // someone else takes the resident next round
#pragma unroll
            for (int u = 0; u < NR; ++u) {
                xr[u * TG] = rx[u];
                xr[VOFF + u * TG] = rv[u];
            }
        }
        RP_SEG(4)
        if (TIMED) {
            MTIP_WAIT_LDS();
            RP_SEG(5)
        }
        __syncthreads();
        RP_SEG(6)
    }
#undef RP_SEG
}

struct RProjArgs {
    double* coef;                 // (B, N, nlm) complex128 coefficients viewed as doubles; projected in place
    const double* DV;             // per order at voff[l]: (N x k) row-major, q^2 V_l[q][j]
    const double* Vt;             // per order at voff[l]: (k x N) row-major, V_l[q][i] transposed
    double* Vr;                   // (B, utot) right singular vectors, column-major k x k per order at uoff[l] (warm start)
    double2* U;                   // (B, xtot) complex unknowns U_l (k x (2l+1) row-major) at xoff[l]: the output dict's fxs_unknowns
    const uint8_t* rmask;         // (L+1, N)
    const int *kl, *voff, *uoff, *xoff;
    const int* slots;             // (n_slots, slot_len): order | kind << 8, -1 = none
    int slot_len;
    const int *sched, *sched_off, *sched_rounds;
    int sched_ps, tab_ints;       // tab_ints: ints reserved in LDS for the raw pairing table (even)
    int tab2_entries, lds_bytes;  // int2 entries reserved behind it for the per-sweep translation; dynamic LDS of the launch
    int N, L, nlm, utot, xtot, warm, corr;
    double rp_early, rp_corr2_max;   // closing step: stop sweeping below this rotation size / accept the second-order step below this |E|
    double tabs2, inv_sqrt_np;
    int* sweeps_out;
    long long* dbg;               // mtip_debug_polar_timing: 32 slots per (restart, order): cycles of the phases A, W, J, U, E of wave 0,
                                  // rounds, start and end s_memtime, hardware id (nullptr: off)
};

enum { RP_SOLVE = 0, RP_ZERO = 1, RP_L0 = 2 };

// acc[u] = sum_{kk < K} A_u[row][kk] B_u[kk][col] for the nu <= T 16 x 16 tiles of this wave; v_mfma_f64_16x16x4:
// A[i = lane & 15][kk = lane >> 4], B[kk = lane >> 4][j = lane & 15], D reg r = D[(lane >> 4) + 4 r][lane & 15].  pa[u] / pb[u] point
// at inner index 0 of this lane's row of A_u / column of B_u, consecutive inner indices sa / sb doubles apart.  The operands of
// a chunk of UNR inner steps are requested for ALL tiles first (T * UNR * 2 values in flight), then multiplied: independent
// accumulators back to back on the matrix pipe while the partner wave of the SIMD loads.  No branches and no selects on the
// loads (a `cond ? load : 0` compiles to an exec-masked block with a full wait at its join): rows and columns outside the
// matrices are clamped by the caller -- they only reach outputs that are never stored -- and the inner indices beyond K are
// clamped to K - 1 and multiplied by zero.
template <int T, int UNR>
__device__ __forceinline__ void rp_tiles(v4f64 (&acc)[T], const double* const (&pa)[T], int sa, const double* const (&pb)[T], int sb,
                                         int K, int lk, int nu) {
#pragma unroll
    for (int u = 0; u < T; ++u) acc[u] = v4f64{0.0, 0.0, 0.0, 0.0};
    // two register sets: the operands of the next chunk are requested before the current chunk is multiplied (the two waves of a
    // SIMD run this in lock step after a barrier: without the prefetch both load, then both queue on the matrix pipe).  Running
    // pointers: an integer multiply per operand (inner index x run-time stride) costs as much issue time as the MFMA it feeds.
    double a0[T][UNR], b0[T][UNR], a1[T][UNR], b1[T][UNR];
    const int n_chunks = (K + 4 * UNR - 1) / (4 * UNR), n_full = K / (4 * UNR);
    const int sa4 = 4 * sa, sb4 = 4 * sb;
    const double *qa[T], *qb[T];
#pragma unroll
    for (int u = 0; u < T; ++u) {
        qa[u] = pa[u] + lk * sa;
        qb[u] = pb[u] + lk * sb;
    }
    int next = 0;                                                // chunk the next request loads (requests come in order)
    auto request = [&](double (&a)[T][UNR], double (&bb)[T][UNR]) {
        if (next < n_full) {                                     // (uniform) whole chunk: no clamps, no masks
#pragma unroll
            for (int u = 0; u < T; ++u)
                if (u < nu) {
#pragma unroll
                    for (int x = 0; x < UNR; ++x) {
                        a[u][x] = qa[u][x * sa4];
                        bb[u][x] = qb[u][x * sb4];
                    }
                    qa[u] += UNR * sa4;
                    qb[u] += UNR * sb4;
                }
        } else {                                                 // the last, partial chunk (inner indices beyond K: clamped, times zero) or past the end
            const int k0 = next * 4 * UNR;
#pragma unroll
            for (int u = 0; u < T; ++u)
                if (u < nu) {
#pragma unroll
                    for (int x = 0; x < UNR; ++x) {
                        const bool in = k0 + 4 * x + lk < K;
                        const int xo = in ? x : 0;               // (step 0 of the last chunk exists for the lanes that matter: clamp to it)
                        const bool in0 = k0 + lk < K;
                        a[u][x] = (in0 ? qa[u][xo * sa4] : 0.0) * (in ? 1.0 : 0.0);
                        bb[u][x] = in0 ? qb[u][xo * sb4] : 0.0;
                    }
                }
        }
        ++next;
    };
    auto multiply = [&](const double (&a)[T][UNR], const double (&bb)[T][UNR]) {
#pragma unroll
        for (int x = 0; x < UNR; ++x)
#pragma unroll
            for (int u = 0; u < T; ++u)
                if (u < nu) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][x], bb[u][x], acc[u], 0, 0, 0);
    };
    request(a0, b0);
    for (int ch = 0; ch < n_chunks; ch += 2) {
        if (ch + 1 < n_chunks) request(a1, b1);
        multiply(a0, b0);
        if (ch + 2 < n_chunks) request(a0, b0);
        if (ch + 1 < n_chunks) multiply(a1, b1);
    }
}

// value of the partner lane (lane ^ 1): the (Re, Im) parts of one m sit in neighbouring lanes
__device__ __forceinline__ double rp_partner(double v) { return dpp_mov<0xB1>(v); }

// the active order l of restart b: products, Jacobi, apply.  UNR: inner steps of the products requested at a time (register budget
// of the launch bound: 4 with two waves per SIMD, 2 with three)
template <int UNR, int RP_ACC, bool BIG, int TG>
__device__ __forceinline__ void rp_solve(const RProjArgs& A, int b, int l, RpShared& sh, double* sm) {
    const int tid = threadIdx.x, nthreads = blockDim.x;
    // (the wave index through readfirstlane: everything derived from it -- tile lists, tile counts -- is then scalar, and the
    // guards around the products are branches, not exec-mask regions with a full wait at every join)
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = nthreads >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int N = A.N, k = A.kl[l], n = 2 * l + 1, n2 = 2 * l + 2;
    const int nc = (n2 + 15) >> 4;                       // blocks of 16 columns / rows
    const int nr = TG == 16 ? nc : rp_pad_nrt(nc, TG);   // row slots per lane (the same for the k rows of V_r: k = 2l+1)
    // up to RP_PAD_MAX_NR row slots: zero-padded columns, one stride, V_r a fixed distance behind X~ (rp_sweep_pad); beyond
    // (config 5) the matrices fill the CU: tight columns, predicates on the last row slot (rp_sweep)
    const RpLayout lay = rp_layout(k, n2, TG, A.tab_ints, A.tab2_entries);
    if (lay.end_bytes > (size_t)A.lds_bytes) {           // never with the launcher of this file: host and device share rp_layout
        if (tid == 0) A.sweeps_out[b * (A.L + 1) + l] = RP_LAYOUT_ERROR;
        return;
    }
    const bool pad = lay.pad;
    const int ns = lay.ns, ks = lay.ks;                  // odd column strides
    double* Xs = sm;
    double* Vs = sm + lay.v_off;
    // behind the matrices: the pairing table of the current column count (raw, and translated per sweep in pad mode)
    int* s_tab = reinterpret_cast<int*>(sm + lay.tab_off);
    int2* s_tab2 = reinterpret_cast<int2*>(s_tab + A.tab_ints);
    const double* DV = A.DV + A.voff[l];
    const double* Vt = A.Vt + A.voff[l];
    double* coef = A.coef + ((size_t)b * N * A.nlm + (size_t)l * (l + 1)) * 2;    // (Re, Im) of m = 0 on shell 0
    const size_t cstride = (size_t)A.nlm * 2;                                   // doubles between shells
    double* Vr = A.Vr + (size_t)b * A.utot + A.uoff[l];
    const int ntm_k = (k + 15) >> 4, ntn = (n2 + 15) >> 4;
    long long* dbg = A.dbg ? A.dbg + ((size_t)b * (A.L + 1) + l) * MTIP_POLAR_DBG_SLOTS : nullptr;
    long long t0 = 0, t1 = 0;
    long long n_rounds_done = 0;
    if (dbg) t0 = clock64();
#define RP_STAMP(SLOT)                                  \
    if (dbg) {                                          \
        t1 = clock64();                                 \
        if (tid == 0) dbg[SLOT] = t1 - t0;              \
        t0 = t1;                                        \
    }
    static_assert(MTIP_POLAR_DBG_SLOTS >= 40, "slots 10..39 hold the per-wave and per-segment sums");
    if (dbg && tid < 30) dbg[10 + tid] = 0;
    if (dbg && tid == 0) {
        dbg[6] = t0;
        dbg[8] = (long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));      // HW_ID
    }
    if (pad) {                                           // zero padding (rows beyond the matrices, unused columns) once
        for (int e = tid; e < 2 * rp_pad_voff(nc, TG); e += nthreads) Xs[e] = 0.0;
        __syncthreads();
    }
    // ---- A: X~^T[j][rho'] = sum_q DV[q][j] I~[q][rho'] -> Xs[j * ns + rho'] ------------------------------------------
    // (q^2 V)[q][j]: q stride k;  I~[q][rho']: q stride = one shell of coefficients
    for (int base = 0; base < ntm_k * ntn; base += nwaves * RP_ACC) {
        v4f64 acc[RP_ACC];
        const double *pa[RP_ACC], *pb[RP_ACC];
        int nu = 0;
#pragma unroll
        for (int u = 0; u < RP_ACC; ++u) {
            const int tile = base + wave + u * nwaves;
            const bool ok = tile < ntm_k * ntn;
            const int tq = ok ? tile : 0;
            const int tm = tq / ntn, tn = tq - tm * ntn;
            const int j = tm * 16 + li, rho = tn * 16 + li;
            pa[u] = DV + (j < k ? j : k - 1);
            pb[u] = coef + (rho < n2 ? rho : n2 - 1);
            if (ok) nu = u + 1;
        }
        rp_tiles<RP_ACC, UNR>(acc, pa, k, pb, (int)cstride, N, lk, nu);
#pragma unroll
        for (int u = 0; u < RP_ACC; ++u) {
            const int tile = base + wave + u * nwaves;
            if (tile < ntm_k * ntn) {
                const int tm = tile / ntn, tn = tile - tm * ntn;
                const int rho = tn * 16 + li;
                const double f = rho == 1 ? 0.0 : (rho >= 2 ? 1.4142135623730951 : 1.0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int jj = tm * 16 + lk + 4 * r;
                    if (jj < k && rho < n2) Xs[(size_t)jj * ns + rho] = f * acc[u][r];
                }
            }
        }
    }
    // V_r: the previous step's right singular vectors (warm start) or the identity
    for (int e = tid; e < k * ks; e += nthreads) {
        const int cc = e / ks, i = e - cc * ks;
        double v = 0.0;
        if (i < k) v = A.warm ? Vr[(size_t)cc * k + i] : (cc == i ? 1.0 : 0.0);
        if (i < k || !pad) Vs[e] = v;
    }
    __syncthreads();
    RP_STAMP(0)
    // ---- W: X~ <- X~ V_r:  D[c][rho'] = sum_j V_r[j][c] X~[rho'][j]  (held in registers until every wave has read X~) --
    if (A.warm) {
        // V_r[j][c] = Vs[c * ks + j]: j stride 1;  X~[rho'][j] = Xs[j * ns + rho']: j stride ns
        v4f64 acc[RP_ACC];
        {
            const double *pa[RP_ACC], *pb[RP_ACC];
            int nu = 0;
#pragma unroll
            for (int u = 0; u < RP_ACC; ++u) {
                const int tile = wave + u * nwaves;
                const bool ok = tile < ntm_k * ntn;
                const int tq = ok ? tile : 0;
                const int tm = tq / ntn, tn = tq - tm * ntn;
                const int cc = tm * 16 + li, rho = tn * 16 + li;
                pa[u] = Vs + (cc < k ? cc : k - 1) * ks;
                pb[u] = Xs + (rho < n2 ? rho : n2 - 1);
                if (ok) nu = u + 1;
            }
            rp_tiles<RP_ACC, UNR>(acc, pa, 1, pb, ns, k, lk, nu);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < RP_ACC; ++u) {
            const int tile = wave + u * nwaves;
            if (tile < ntm_k * ntn) {
                const int tm = tile / ntn, tn = tile - tm * ntn;
                const int rho = tn * 16 + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int cc = tm * 16 + lk + 4 * r;
                    if (cc < k && rho < n2) Xs[(size_t)cc * ns + rho] = acc[u][r];
                }
            }
        }
        __syncthreads();
    }
    RP_STAMP(1)
    // ---- J: one-sided Jacobi sweeps ------------------------------------------------------------------------------------
    const int ngroups = nthreads / TG;
    const int group = tid / TG, t = tid % TG;
    const bool xl_ok = t + (nr - 1) * TG < n2, vl_ok = t + (nr - 1) * TG < k;
    double S = 0.0;
    const bool corr = (pad || BIG) && A.corr != 0;
    double early2 = corr ? A.rp_early * A.rp_early : JL_EARLY * JL_EARLY;
    bool use_corr = false, corr2 = false;
    // k x kp: Gram matrix, then D^-1 (1 - Delta D^-1).  Padded layout: in place of the pairing tables (done with by then).  Tight layout
    // (config 5: the matrices fill the CU): in place of V_r, which goes to global memory first -- where the next call wants it anyway --
    // and is read back for one more sweep or for the U product (`v_saved`)
    double* Pm = pad ? reinterpret_cast<double*>(s_tab) : Vs;
    const int kp = k | 1;
    bool v_saved = false;
    if (k > 1) {
        int tab_ke = -1;
        for (int sweep = 0; sweep < JAC_MAX_SWEEPS; ++sweep) {
            // column norms; deflation of numerical-zero columns (k_proj.hip: X is numerically rank deficient once the
            // density has a support) and compaction of the tournament to the columns that are left
            double Sl = 0.0;
            for (int cc0 = 0; cc0 < k; cc0 += ngroups) {
                const int cc = cc0 + group;
                double s2 = 0.0;
                if (cc < k)
                    for (int u = 0; u < nr; ++u)
                        if (t + u * TG < n2) {
                            const double x = Xs[(size_t)cc * ns + t + u * TG];
                            s2 = fma(x, x, s2);
                        }
                s2 = rp_sum1<TG>(s2);
                if (cc < k && t == 0) sh.isig[cc] = s2;
                Sl = fmax(Sl, s2);
            }
            if (t == 0) sh.gmax[group] = Sl;
            __syncthreads();
            S = 0.0;
            for (int gq = 0; gq < ngroups; ++gq) S = fmax(S, sh.gmax[gq]);
            for (int cc0 = 0; cc0 < k; cc0 += ngroups) {
                const int cc = cc0 + group;
                if (cc < k && sh.isig[cc] <= (JAC_DEFLATE * JAC_DEFLATE) * S && sh.isig[cc] > 0.0)
                    for (int u = 0; u < nr; ++u)
                        if (t + u * TG < n2) Xs[(size_t)cc * ns + t + u * TG] = 0.0;
            }
            if (tid < 64) {                                    // wave 0: ballot + prefix popcount (k <= 128)
                const double thr = (JAC_DEFLATE * JAC_DEFLATE) * S;
                const bool a0 = tid < k && sh.isig[tid] > thr;
                const bool a1 = tid + 64 < k && sh.isig[tid + 64] > thr;
                const unsigned long long m0 = __ballot(a0), m1 = __ballot(a1);
                const unsigned long long below = (1ull << tid) - 1ull;
                if (a0) sh.perm[__popcll(m0 & below)] = tid;
                if (a1) sh.perm[__popcll(m0) + __popcll(m1 & below)] = tid + 64;
                if (tid == 0) sh.keff = __popcll(m0) + __popcll(m1);
            }
            __syncthreads();
            const int ke = sh.keff;
            bool big = false;
            if (ke >= 2) {
                const int* gtab = A.sched + A.sched_off[ke];
                const int nrd = A.sched_rounds[ke];
                if (pad) {
                    // the raw pairing table of this column count in LDS once, translated for rp_sweep_pad every sweep (the
                    // compaction of the deflated columns may have changed): offsets of the resident / mover columns, flags
                    if (tab_ke != ke) {
                        for (int e = tid; e < nrd * A.sched_ps; e += nthreads) s_tab[e] = gtab[e];
                        tab_ke = ke;
                        __syncthreads();
                    }
                    const int ps = A.sched_ps;
                    const int dummy = (16 * nc - 1) * ns;        // the all-zero column
                    for (int e = tid; e < nrd * ngroups; e += nthreads) {
                        const int r = e / ngroups, gq = e - r * ngroups;
                        const int raw = gq < ps ? s_tab[r * ps + gq] : 0;
                        int x = (dummy << 2) | 1, y = dummy;
                        if (raw & JS_ACTIVE) {
                            const int prev = r > 0 ? s_tab[(r - 1) * ps + gq] : 0;
                            const int next = r + 1 < nrd ? s_tab[(r + 1) * ps + gq] : 0;
                            // a resident stays in registers only from one active round to the next one with the same resident
                            const int load = (!(prev & JS_ACTIVE) || (prev & JS_WB) || (prev & 255) != (raw & 255)) ? 1 : 0;
                            const int wb = ((raw & JS_WB) || !(next & JS_ACTIVE)) ? 2 : 0;
                            x = ((sh.perm[raw & 255] * ns) << 2) | wb | load;
                            y = sh.perm[(raw >> 8) & 255] * ns;
                        }
                        s_tab2[e] = make_int2(x, y);
                    }
                    __syncthreads();
                    double* xt = Xs + t;
                    if (dbg != nullptr && nc == 5) {             // diagnostic instance with segment timers
                        long long tacc[7] = {0, 0, 0, 0, 0, 0, 0};
                        rp_sweep_pad<5, TG, true>(xt, s_tab2, nrd, ngroups, group, A.tabs2, S, early2, big, tacc);
                        if ((tid & 63) == 0 && (tid >> 6) < 8) {   // per wave: busy, LDS drain + barrier; wave 0 and 5 also the segments
                            dbg[10 + (tid >> 6)] += tacc[0] + tacc[1] + tacc[2] + tacc[3] + tacc[4];
                            dbg[18 + (tid >> 6)] += tacc[5] + tacc[6];
                            if (tid == 0 || tid == 320)
                                for (int i = 0; i < 7; ++i) dbg[(tid ? 33 : 26) + i] += tacc[i];
                        }
                    } else {
                        switch (nc) {
                        case 1: rp_sweep_pad<1, TG>(xt, s_tab2, nrd, ngroups, group, A.tabs2, S, early2, big); break;
                        case 2: rp_sweep_pad<2, TG>(xt, s_tab2, nrd, ngroups, group, A.tabs2, S, early2, big); break;
                        case 3: rp_sweep_pad<3, TG>(xt, s_tab2, nrd, ngroups, group, A.tabs2, S, early2, big); break;
                        case 4: rp_sweep_pad<4, TG>(xt, s_tab2, nrd, ngroups, group, A.tabs2, S, early2, big); break;
                        default: rp_sweep_pad<5, TG>(xt, s_tab2, nrd, ngroups, group, A.tabs2, S, early2, big); break;
                        }
                    }
                } else if (BIG) {
                    // 6 or 7 row slots: the table is read from L2 one round ahead
                    if (nr == 6) rp_sweep<6>(Xs, Vs, ns, ks, t, group, gtab, nrd, A.sched_ps, sh.perm, xl_ok, vl_ok, A.tabs2, S, early2, big);
                    else rp_sweep<7>(Xs, Vs, ns, ks, t, group, gtab, nrd, A.sched_ps, sh.perm, xl_ok, vl_ok, A.tabs2, S, early2, big);
                }
                n_rounds_done += nrd;
            }
            if (t == 0) sh.gmax[group] = big ? 1.0 : 0.0;
            __syncthreads();
            if (tid == 0) {
                double m = 0.0;
                for (int gq = 0; gq < ngroups; ++gq) m = fmax(m, sh.gmax[gq]);
                sh.cont = (m > 0.0) ? 1 : 0;                   // 1: some rotation was above `early2`: sweep on (quadratic convergence)
                A.sweeps_out[b * (A.L + 1) + l] = (sweep + 1) | (ke << 8);
            }
            __syncthreads();
            const int cont = sh.cont;
            __syncthreads();
            if (cont) continue;
            if (!corr) break;                                    // classic exit: the sweep just done was the confirming one
            if (!pad) {
                for (int e = tid; e < k * k; e += nthreads) {
                    const int cc = e / k, i = e - cc * k;
                    Vr[e] = Vs[(size_t)cc * ks + i];
                }
                v_saved = true;
                __syncthreads();
            }
            // ---- Gram matrix G = W^T W on the matrix pipe, its diagonal (sigma^2) and the largest |E_ij| ----
            for (int base = 0; base < ntm_k * ntm_k; base += nwaves * RP_ACC) {
                v4f64 acc[RP_ACC];
                const double *pa[RP_ACC], *pb[RP_ACC];
                int nu = 0;
#pragma unroll
                for (int u = 0; u < RP_ACC; ++u) {
                    const int tile = base + wave + u * nwaves;
                    const bool ok = tile < ntm_k * ntm_k;
                    const int tq = ok ? tile : 0;
                    const int tm = tq / ntm_k, tn = tq - tm * ntm_k;
                    const int i = tm * 16 + li, j = tn * 16 + li;
                    pa[u] = Xs + (size_t)(i < k ? i : k - 1) * ns;
                    pb[u] = Xs + (size_t)(j < k ? j : k - 1) * ns;
                    if (ok) nu = u + 1;
                }
                rp_tiles<RP_ACC, UNR>(acc, pa, 1, pb, 1, n2, lk, nu);
#pragma unroll
                for (int u = 0; u < RP_ACC; ++u) {
                    const int tile = base + wave + u * nwaves;
                    if (tile < ntm_k * ntm_k) {
                        const int tm = tile / ntm_k, tn = tile - tm * ntm_k;
                        const int j = tn * 16 + li;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = tm * 16 + lk + 4 * r;
                            if (i < k && j < k) Pm[(size_t)i * kp + j] = acc[u][r];
                        }
                    }
                }
            }
            __syncthreads();
            if (tid < k) {
                const double g = Pm[(size_t)tid * kp + tid];
                sh.sig[tid] = sqrt(fmax(g, 0.0));
                sh.isig[tid] = g > 1e-300 ? 1.0 / sqrt(g) : 0.0;
            }
            __syncthreads();
            double em = 0.0;
            for (int e = tid; e < k * k; e += nthreads) {
                const int i = e / k, j = e - i * k;
                const double g = Pm[(size_t)i * kp + j], gi = Pm[(size_t)i * kp + i], gj = Pm[(size_t)j * kp + j];
                // (pairs the sweeps leave alone -- below the absolute tolerance, rp_params -- are left alone here too)
                const bool live = i != j && g * g > A.tabs2 * fmax(gi, gj) * S && g * g > (JAC_TOL * JAC_TOL) * gi * gj;
                if (live) em = fmax(em, fabs(g) * sh.isig[i] * sh.isig[j]);
            }
            for (int o = 32; o > 0; o >>= 1) em = fmax(em, __shfl_xor(em, o, 64));
            if ((tid & 63) == 0) sh.red[tid >> 6] = em;
            __syncthreads();
            em = 0.0;
            for (int wv = 0; wv < nwaves; ++wv) em = fmax(em, sh.red[wv]);
            __syncthreads();
            if (em <= RP_CORR_SKIP) break;                       // already orthogonal to rounding: the plain finish
            if (em <= A.rp_corr2_max) {
                use_corr = true;
                corr2 = em > RP_CORR_MAX;
                // (bits 24, 25 of the sweep record: closed by the first / second order step)
                if (tid == 0) A.sweeps_out[b * (A.L + 1) + l] |= corr2 ? (2 << 24) : (1 << 24);
                break;
            }
            tab_ke = -1;                                         // not there yet: one more sweep (the Gram matrix overwrote the
                                                                 // pairing table: staged again), then the check again
            if (!pad) {                                          // ... or V_r: read back
                for (int e = tid; e < k * k; e += nthreads) {
                    const int cc = e / k, i = e - cc * k;
                    Vs[(size_t)cc * ks + i] = *reinterpret_cast<const volatile double*>(Vr + e);   // (written by this kernel: not through a stale L1 line)
                }
                v_saved = false;
                __syncthreads();
            }
        }
    } else if (tid == 0) {
        A.sweeps_out[b * (A.L + 1) + l] = 0 | (k << 8);
    }
    RP_STAMP(2)
    // ---- sigma_c; V_r for the next call -----------------------------------------------------------------------------------
    if (!use_corr) {
        for (int cc0 = 0; cc0 < k; cc0 += ngroups) {           // uniform trip count: DPP sums need the whole group
            const int cc = cc0 + group;
            double s2 = 0.0;
            if (cc < k)
                for (int u = 0; u < nr; ++u)
                    if (t + u * TG < n2) {
                        const double x = Xs[(size_t)cc * ns + t + u * TG];
                        s2 = fma(x, x, s2);
                    }
            s2 = rp_sum1<TG>(s2);
            if (cc < k && t == 0) sh.isig[cc] = s2 > 1e-300 ? 1.0 / sqrt(s2) : 0.0;
        }
    }
    if (!v_saved)
        for (int e = tid; e < k * k; e += nthreads) {
            const int cc = e / k, i = e - cc * k;
            Vr[e] = Vs[(size_t)cc * ks + i];
        }
    __syncthreads();
    if (!use_corr) {
        // columns of X~ <- left singular vectors W / sigma (numerical-zero columns: 0)
        for (int e = tid; e < k * ns; e += nthreads) Xs[e] *= sh.isig[e / ns];
        if (v_saved)                                           // (tight layout, Gram matrix orthogonal to rounding: V_r back in place)
            for (int e = tid; e < k * k; e += nthreads) {
                const int cc = e / k, i = e - cc * k;
                Vs[(size_t)cc * ks + i] = *reinterpret_cast<const volatile double*>(Vr + e);   // (written by this kernel: not through a stale L1 line)
            }
        __syncthreads();
    } else {
        // columns of X~ <- W G^-1/2 to first or second order in E: the symmetric factor P in place of the Gram matrix, then one
        // product held in registers across the barrier (it overwrites its own operand).  V_r is in global memory by now: its LDS
        // block serves as scratch for A~ and is read back before the U product.
        // (tight layout: P sits in V_r's block already; A~ goes through the order's slot of the output U, written only after this)
        double* Ab = pad ? Vs : reinterpret_cast<double*>(A.U + (size_t)b * A.xtot + A.xoff[l]);
        for (int e = tid; e < k * k; e += nthreads) {
            const int i = e / k, j = e - i * k;
            const double g = Pm[(size_t)i * kp + j], gi = sh.sig[i] * sh.sig[i], gj = sh.sig[j] * sh.sig[j];
            const bool live = i != j && g * g > A.tabs2 * fmax(gi, gj) * S && g * g > (JAC_TOL * JAC_TOL) * gi * gj;
            const double ss = sh.sig[i] + sh.sig[j];
            const double a = live && ss > 0.0 ? g * sh.isig[i] * sh.isig[j] / ss : 0.0;      // A_ij
            if (corr2) {
                Pm[(size_t)i * kp + j] = a;
                Ab[(size_t)i * kp + j] = a * sh.sig[j];
            } else {
                Pm[(size_t)i * kp + j] = i == j ? sh.isig[i] : -a;
            }
        }
        __syncthreads();
        if (corr2) {
            // T1 = A~ A~^T, T2 = A~ A (A symmetric): all tiles in registers, then P = D^-1 - A + T1 o S + T2 in place of A
            v4f64 t1[RP_ACC], t2[RP_ACC];
            const double *pa[RP_ACC], *pb[RP_ACC], *pc[RP_ACC];
            int nu = 0;
#pragma unroll
            for (int u = 0; u < RP_ACC; ++u) {
                const int tile = wave + u * nwaves;
                const bool ok = tile < ntm_k * ntm_k;
                const int tq = ok ? tile : 0;
                const int tm = tq / ntm_k, tn = tq - tm * ntm_k;
                const int i = tm * 16 + li, j = tn * 16 + li;
                pa[u] = Ab + (size_t)(i < k ? i : k - 1) * kp;
                pb[u] = Ab + (size_t)(j < k ? j : k - 1) * kp;
                pc[u] = Pm + (size_t)(j < k ? j : k - 1) * kp;
                if (ok) nu = u + 1;
            }
            rp_tiles<RP_ACC, UNR>(t1, pa, 1, pb, 1, k, lk, nu);
            rp_tiles<RP_ACC, UNR>(t2, pa, 1, pc, 1, k, lk, nu);
            __syncthreads();
#pragma unroll
            for (int u = 0; u < RP_ACC; ++u) {
                const int tile = wave + u * nwaves;
                if (tile < ntm_k * ntm_k) {
                    const int tm = tile / ntm_k, tn = tile - tm * ntm_k;
                    const int j = tn * 16 + li;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = tm * 16 + lk + 4 * r;
                        if (i < k && j < k) {
                            const double ss = sh.sig[i] + sh.sig[j];
                            Pm[(size_t)i * kp + j] = (i == j ? sh.isig[i] : 0.0) - Pm[(size_t)i * kp + j] + (ss > 0.0 ? t1[u][r] / ss : 0.0) + t2[u][r];
                        }
                    }
                }
            }
            __syncthreads();
        }
        v4f64 acc[RP_ACC];
        {
            const double *pa[RP_ACC], *pb[RP_ACC];
            int nu = 0;
#pragma unroll
            for (int u = 0; u < RP_ACC; ++u) {
                const int tile = wave + u * nwaves;
                const bool ok = tile < ntm_k * ntn;
                const int tq = ok ? tile : 0;
                const int tm = tq / ntn, tn = tq - tm * ntn;
                const int j = tm * 16 + li, rho = tn * 16 + li;
                pa[u] = Pm + (size_t)(j < k ? j : k - 1) * kp;
                pb[u] = Xs + (rho < n2 ? rho : n2 - 1);
                if (ok) nu = u + 1;
            }
            rp_tiles<RP_ACC, UNR>(acc, pa, 1, pb, ns, k, lk, nu);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < RP_ACC; ++u) {
            const int tile = wave + u * nwaves;
            if (tile < ntm_k * ntn) {
                const int tm = tile / ntn, tn = tile - tm * ntn;
                const int rho = tn * 16 + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = tm * 16 + lk + 4 * r;
                    if (j < k && rho < n2) Xs[(size_t)j * ns + rho] = acc[u][r];
                }
            }
        }
        if (corr2 || v_saved) {                                // V_r back into its LDS block (its zero padding was never an operand)
            for (int e = tid; e < k * k; e += nthreads) {
                const int cc = e / k, i = e - cc * k;
                Vs[(size_t)cc * ks + i] = *reinterpret_cast<const volatile double*>(Vr + e);   // (written by this kernel: not through a stale L1 line)
            }
        }
        __syncthreads();
    }
    // ---- U: U~^T[rho'][i] = sum_c (X~[rho'][c] / sigma_c) V_r[i][c]:  D[i][rho'] -> Xs[i * ns + rho'], complex U_l -> A.U ----
    {
        double2* Uo = A.U + (size_t)b * A.xtot + A.xoff[l];
        // V_r[i][c] = Vs[c * ks + i]: c stride ks;  (X~ / sigma)[rho'][c] = Xs[c * ns + rho']: c stride ns
        v4f64 acc[RP_ACC];
        {
            const double *pa[RP_ACC], *pb[RP_ACC];
            int nu = 0;
#pragma unroll
            for (int u = 0; u < RP_ACC; ++u) {
                const int tile = wave + u * nwaves;
                const bool ok = tile < ntm_k * ntn;
                const int tq = ok ? tile : 0;
                const int tm = tq / ntn, tn = tq - tm * ntn;
                const int i = tm * 16 + li, rho = tn * 16 + li;
                pa[u] = Vs + (i < k ? i : k - 1);
                pb[u] = Xs + (rho < n2 ? rho : n2 - 1);
                if (ok) nu = u + 1;
            }
            rp_tiles<RP_ACC, UNR>(acc, pa, ks, pb, ns, k, lk, nu);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < RP_ACC; ++u) {
            const int tile = wave + u * nwaves;
            const bool t_ok = tile < ntm_k * ntn;
            const int tm = t_ok ? tile / ntn : 0, tn = t_ok ? tile - tm * ntn : 0;
            const int rho = tn * 16 + li, m = rho >> 1;
            const double sg = (m & 1) ? -1.0 : 1.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = tm * 16 + lk + 4 * r;
                const double v = acc[u][r];
                const double p = rp_partner(v);                  // (every lane: uniform control flow around DPP)
                if (t_ok && i < k && rho < n2) {
                    Xs[(size_t)i * ns + rho] = v;
                    // U[i][l + m] = (re + i im) / sqrt2,  U[i][l - m] = (-1)^m conj(.),  U[i][l] = re
                    if (!(rho & 1)) {
                        Uo[(size_t)i * n + l + m] = m ? make_double2(0.7071067811865476 * v, 0.7071067811865476 * p) : make_double2(v, 0.0);
                    } else if (m) {
                        Uo[(size_t)i * n + l - m] = make_double2(sg * 0.7071067811865476 * p, -sg * 0.7071067811865476 * v);
                    }
                }
            }
        }
        __syncthreads();
    }
    RP_STAMP(3)
    // ---- E: I~'[q][rho'] = sum_i V[q][i] U~^T[rho'][i] -> I'_{l, +-m}(q) on the masked shells ----------------------------------
    {
        const int ntq = (N + 15) >> 4;
        const uint8_t* rm = A.rmask + (size_t)l * N;
        // V[q][i] = Vt[i * N + q]: i stride N;  U~^T[rho'][i] = Xs[i * ns + rho']: i stride ns
        for (int base = 0; base < ntq * ntn; base += nwaves * RP_ACC) {
            v4f64 acc[RP_ACC];
            const double *pa[RP_ACC], *pb[RP_ACC];
            int nu = 0;
#pragma unroll
            for (int u = 0; u < RP_ACC; ++u) {
                const int tile = base + wave + u * nwaves;
                const bool ok = tile < ntq * ntn;
                const int tq = ok ? tile : 0;
                const int tm = tq / ntn, tn = tq - tm * ntn;
                const int q = tm * 16 + li, rho = tn * 16 + li;
                pa[u] = Vt + (q < N ? q : N - 1);
                pb[u] = Xs + (rho < n2 ? rho : n2 - 1);
                if (ok) nu = u + 1;
            }
            rp_tiles<RP_ACC, UNR>(acc, pa, N, pb, ns, k, lk, nu);
#pragma unroll
            for (int u = 0; u < RP_ACC; ++u) {
                const int tile = base + wave + u * nwaves;
                const bool t_ok = tile < ntq * ntn;
                const int tq = t_ok ? tile : 0;
                const int tm = tq / ntn, tn = tq - tm * ntn;
                const int rho = tn * 16 + li, m = rho >> 1;
                const double sg = (m & 1) ? -1.0 : 1.0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qq = tm * 16 + lk + 4 * r;
                    const double v = acc[u][r];
                    const double p = rp_partner(v);              // (every lane: uniform control flow around DPP)
                    if (t_ok && qq < N && rho < n2 && rm[qq]) {
                        double2* dst = reinterpret_cast<double2*>(coef + (size_t)qq * cstride);     // I_{l,0}(qq)
                        if (!(rho & 1)) {
                            dst[m] = m ? make_double2(0.7071067811865476 * v, 0.7071067811865476 * p) : make_double2(v, 0.0);
                        } else if (m) {
                            dst[-m] = make_double2(sg * 0.7071067811865476 * p, -sg * 0.7071067811865476 * v);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();                                           // LDS is reused by the next order of the slot
    RP_STAMP(4)
    if (dbg && tid == 0) {
        dbg[5] = n_rounds_done;
        dbg[7] = t1;
    }
#undef RP_STAMP
}

// grid = (restart, slot); slots are listed heaviest first, the restart index runs fastest: the workgroups that set the
// duration of the launch are dispatched first
template <int MAXT, int UNR, int ACC, bool BIG, int TG>
__global__ void __launch_bounds__(MAXT) k_rproj(RProjArgs A) {
    HIP_DYNAMIC_SHARED(double, sm)
    __shared__ RpShared sh;
    const int b = (int)blockIdx.x;
    const int tid = threadIdx.x, nthreads = blockDim.x;
    for (int it = 0; it < A.slot_len; ++it) {
        const int e = A.slots[(size_t)blockIdx.y * A.slot_len + it];
        if (e < 0) break;                                      // uniform per block
        const int l = e & 255, kind = e >> 8;
        if (kind == RP_SOLVE) {
            rp_solve<UNR, ACC, BIG, TG>(A, b, l, sh, sm);
        } else if (kind == RP_ZERO) {
            // used order with V_l = 0 (odd_orders_to_0): I'_l = 0 on the masked shells, its unknowns stay 0
            const int n = 2 * l + 1;
            for (int x = tid; x < A.N * n; x += nthreads) {
                const int q = x / n, mm = x - q * n;
                if (A.rmask[(size_t)l * A.N + q])
                    reinterpret_cast<double2*>(A.coef)[((size_t)b * A.N + q) * A.nlm + (size_t)l * l + mm] = make_double2(0.0, 0.0);
            }
        } else {
            // l = 0 (fxs_Projections.py:840, 870): unknown = sign of V_0^T D^2 Re I_00, I'_00 = V_0 on the masked shells, all / sqrt(N_p)
            const double* DV = A.DV + A.voff[0];
            const double* Vt = A.Vt + A.voff[0];
            double2* c0 = reinterpret_cast<double2*>(A.coef) + (size_t)b * A.N * A.nlm;
            double x = 0.0;
            for (int q = tid; q < A.N; q += nthreads) x = fma(DV[(size_t)q * A.kl[0]], c0[(size_t)q * A.nlm].x, x);
            for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
            if ((tid & 63) == 0) sh.red[tid >> 6] = x;
            __syncthreads();
            if (tid == 0) {
                double s = 0.0;
                for (int wv = 0; wv < (nthreads >> 6); ++wv) s += sh.red[wv];
                const double s2 = s * s;
                A.U[(size_t)b * A.xtot + A.xoff[0]] = make_double2(s2 > 1e-300 ? s / sqrt(s2) : 0.0, 0.0);
                A.sweeps_out[b * (A.L + 1)] = 0 | (1 << 8);
            }
            for (int q = tid; q < A.N; q += nthreads) {
                const double2 v = A.rmask[q] ? make_double2(Vt[q], 0.0) : c0[(size_t)q * A.nlm];
                c0[(size_t)q * A.nlm] = cscale(v, A.inv_sqrt_np);
            }
            __syncthreads();
        }
    }
}

// ---- host side --------------------------------------------------------------------------------------------------------
// lanes per column pair, threads and dynamic LDS of a k_rproj launch.  LDS: the largest order's matrices -- padded layout up to
// RP_PAD_MAX_NR blocks of 16 columns, behind them the raw pairing table and its per-sweep translation (one int2 per round and
// group); tight layout beyond (table read from L2) -- whichever needs more; tab_ints = ints reserved for the raw table
struct RpGeom {
    int tg = 16, threads = 256, tab_ints = 0, tab2_entries = 0, acc = RP_ACC_MAX;
    bool big = false;                  // some order has the tight layout (k_rproj<768, ...>)
    size_t lds = 0;
};
static RpGeom rp_launch_geometry(const mtip_ctx* c) {
    RpGeom g;
    int kpad = 0, kbig = 0;
    for (int l = 1; l <= c->L; ++l) {
        if (!c->active[l]) continue;
        if ((c->kl[l] + 1 + 15) / 16 <= RP_PAD_MAX_NR) kpad = std::max(kpad, c->kl[l]);
        else kbig = std::max(kbig, c->kl[l]);
    }
    const int kmax = std::max(kpad, kbig);
    const int ps = kmax >= 2 ? std::max(c->jsched_ps, 1) : 1;        // row length of the pairing table (it may have been built for more columns)
    const int groups = kmax >= 2 ? std::max(jacobi_groups(kmax), 1) : 1; // pair-groups the largest order keeps busy
    g.tg = 16;
    g.big = kbig > 0;
    g.threads = std::max(256, (groups * g.tg + 63) / 64 * 64);
    if (kpad >= 2) {
        int nrd = 1;                                             // (the schedule has more rounds than columns: 70 at k = 65)
        for (int ke = 2; ke <= kpad && ke < (int)c->jsched_nrd.size(); ++ke) nrd = std::max(nrd, c->jsched_nrd[ke]);
        g.tab_ints = (nrd * ps + 1) & ~1;
        g.tab2_entries = nrd * (g.threads / g.tg);
    }
    // the block holds the largest layout among the solved orders (rp_layout: the definition the kernel takes its pointers from)
    g.lds = RP_SLACK * sizeof(double);
    for (int l = 1; l <= c->L; ++l)
        if (c->active[l] && c->kl[l] >= 2)
            g.lds = std::max(g.lds, rp_layout(c->kl[l], 2 * l + 2, g.tg, g.tab_ints, g.tab2_entries).end_bytes);
    // the in-place products hold all their 16 x 16 tiles in registers across a barrier
    const int nt16 = (kmax + 1 + 15) / 16;
    g.acc = div_up(nt16 * nt16, g.threads / 64);
    return g;
}

// every solved order square (k_l = 2l+1), V_l real, 2l+2 <= 7 row slots of 16, the schedule and the matrices fit
bool rproj_supported(mtip_ctx* c) {
    if (!c->proj_real) return false;
    int kmax = 0;
    for (int l = 0; l <= c->L; ++l) {
        if (!c->used[l]) continue;
        if (!c->v_real[l]) return false;
        if (c->active[l] && l > 0) {
            if (c->kl[l] != 2 * l + 1) return false;
            kmax = std::max(kmax, c->kl[l]);
        }
    }
    if (kmax > 111) return false;
    if (kmax >= 2) {
        if (build_jacobi_schedule(c, kmax) != MTIP_OK) return false;
        const RpGeom g = rp_launch_geometry(c);
        // the instantiations of launch_rproj: <512, ...> up to 512 threads, <768, ...> (tight layout, config 5) up to 768
        if (g.threads > 768) return false;             // (l = 49: 49 pair-groups = 832 threads -> the general kernels)
        if (g.lds + sizeof(RpShared) + 256 > 160 * 1024) return false;
        if (g.acc > (g.threads <= 512 ? 4 : RP_ACC_MAX)) return false;    // tiles per wave of those instantiations
    }
    return true;
}

// cost model of one order inside a slot, in cycles, from the in-kernel timers at 128 x L32 (profiles/r03_rproj_round_timers.txt):
// the four products and the bookkeeping around them 38 k + 1.55 k per column, a sweep 1.08 k rounds of 2000 + 8 k cycles, three
// sweeps per call.  (Round 3 first packed with k (40 + k): the fixed part was missing, and the slots of three or four small
// orders -- {20, 12, 8, 4}: 413 k cycles against 315 k for l = 32 alone -- set the duration of the launch.)
static double rp_cost(int k) { return 38e3 + 1550.0 * k + 3.0 * (1.08 * k) * (2000.0 + 8.0 * k); }

static int build_rproj_tables(mtip_ctx* c) {
    if (c->d_rp_slots != nullptr) return MTIP_OK;
    const int L = c->L, N = c->N;
    // real tables: q^2 V (N x k) and V^T (k x N) per order, from the host copy of V
    std::vector<double> q(N);
    if (mtip_copy(c, q.data(), c->d_q, N * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return MTIP_EHIP;
    std::vector<double> DV((size_t)c->vtot, 0.0), Vt((size_t)c->vtot, 0.0);
    for (int l = 0; l <= L; ++l) {
        const int k = c->kl[l];
        const double2* V = c->h_V.data() + c->voff[l];
        for (int qi = 0; qi < N; ++qi)
            for (int j = 0; j < k; ++j) {
                DV[(size_t)c->voff[l] + (size_t)qi * k + j] = q[qi] * q[qi] * V[(size_t)qi * k + j].x;
                Vt[(size_t)c->voff[l] + (size_t)j * N + qi] = V[(size_t)qi * k + j].x;
            }
    }
    // slots: solved orders by first-fit decreasing into bins of the heaviest order's cost, then the cheap items
    std::vector<int> solve;
    for (int l = L; l >= 1; --l)
        if (c->active[l]) solve.push_back(l);
    std::stable_sort(solve.begin(), solve.end(), [&](int x, int y) { return c->kl[x] > c->kl[y]; });
    std::vector<std::vector<int>> slots;
    std::vector<double> load;
    const double cap = solve.empty() ? 1.0 : rp_cost(c->kl[solve[0]]) * 1.02;
    const int slot_len = 8;
    for (int l : solve) {
        const double w = rp_cost(c->kl[l]);
        size_t s = 0;
        for (; s < slots.size(); ++s)
            if (load[s] + w <= cap && (int)slots[s].size() < slot_len - 2) break;
        if (s == slots.size()) {
            slots.emplace_back();
            load.push_back(0.0);
        }
        slots[s].push_back(l | (RP_SOLVE << 8));
        load[s] += w;
    }
    std::vector<int> cheap;
    if (c->used[0]) cheap.push_back(0 | (RP_L0 << 8));
    for (int l = 1; l <= L; ++l)
        if (c->used[l] && !c->active[l]) cheap.push_back(l | (RP_ZERO << 8));
    if (slots.empty()) {
        slots.emplace_back();
        load.push_back(0.0);
    }
    {
        // cheap items go to the lightest slots, several per slot; more slots are opened when the lists are full
        size_t s = slots.size() - 1;
        for (int item : cheap) {
            size_t tries = 0;
            while ((int)slots[s].size() >= slot_len && tries < slots.size()) {
                s = s == 0 ? slots.size() - 1 : s - 1;
                ++tries;
            }
            if ((int)slots[s].size() >= slot_len) {
                slots.emplace_back();
                load.push_back(0.0);
                s = slots.size() - 1;
            }
            slots[s].push_back(item);
            s = s == 0 ? slots.size() - 1 : s - 1;
        }
    }
    std::vector<int> flat(slots.size() * slot_len, -1);
    for (size_t s = 0; s < slots.size(); ++s)
        for (size_t i = 0; i < slots[s].size(); ++i) flat[s * slot_len + i] = slots[s][i];
    if (hipMalloc((void**)&c->d_rp_DV, DV.size() * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&c->d_rp_Vt, Vt.size() * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&c->d_rp_slots, flat.size() * sizeof(int)) != hipSuccess)
        return MTIP_ENOMEM;
    (void)mtip_copy(c, c->d_rp_DV, DV.data(), DV.size() * sizeof(double), hipMemcpyHostToDevice);
    (void)mtip_copy(c, c->d_rp_Vt, Vt.data(), Vt.size() * sizeof(double), hipMemcpyHostToDevice);
    (void)mtip_copy(c, c->d_rp_slots, flat.data(), flat.size() * sizeof(int), hipMemcpyHostToDevice);
    c->rp_n_slots = (int)slots.size();
    c->rp_slot_len = slot_len;
    return MTIP_OK;
}

void free_rproj_tables(mtip_ctx* c) {
    if (c->d_rp_slots == nullptr && c->d_rp_DV == nullptr) return;
    (void)hipStreamSynchronize(c->stream);
    if (c->d_rp_DV) (void)hipFree(c->d_rp_DV);
    if (c->d_rp_Vt) (void)hipFree(c->d_rp_Vt);
    if (c->d_rp_slots) (void)hipFree(c->d_rp_slots);
    c->d_rp_DV = nullptr;
    c->d_rp_Vt = nullptr;
    c->d_rp_slots = nullptr;
}

// in place on `coef` (the caller has copied I_lm there when its output is a different buffer)
int launch_rproj(mtip_ctx* c, double2* coef) {
    if (int rc = build_rproj_tables(c)) {
        c->err = "real projection tables: out of device memory";
        return rc;
    }
    int kmax = 1;
    for (int l = 1; l <= c->L; ++l)
        if (c->active[l]) kmax = std::max(kmax, c->kl[l]);
    RProjArgs a;
    a.coef = reinterpret_cast<double*>(coef);
    a.DV = c->d_rp_DV; a.Vt = c->d_rp_Vt;
    a.Vr = reinterpret_cast<double*>(c->d_Vr);
    a.U = c->d_U;
    a.rmask = c->d_rmask;
    a.kl = c->d_kl; a.voff = c->d_voff; a.uoff = c->d_uoff; a.xoff = c->d_xoff;
    a.slots = c->d_rp_slots; a.slot_len = c->rp_slot_len;
    a.sched = c->d_jsched; a.sched_off = c->d_jsched_off; a.sched_rounds = c->d_jsched_rounds;
    a.sched_ps = std::max(c->jsched_ps, 1);
    a.N = c->N; a.L = c->L; a.nlm = c->nlm; a.utot = c->utot; a.xtot = c->xtot;
    // cold start every 64 calls bounds the accumulated rounding drift of the carried V_r
    a.warm = (c->vr_kind == 2 && (c->proj_calls % 64) != 0) ? 1 : 0;
    a.tabs2 = c->polar_abs_tol * c->polar_abs_tol;
    a.corr = c->rp_corr ? 1 : 0;
    a.rp_early = c->rp_early;
    a.rp_corr2_max = c->rp_corr2_max;
    a.inv_sqrt_np = 1.0 / std::sqrt(c->n_particles);
    a.sweeps_out = c->d_sweeps;
    a.dbg = c->d_polar_dbg;
    const RpGeom g = rp_launch_geometry(c);
    a.tab_ints = g.tab_ints;
    a.tab2_entries = g.tab2_entries;
    a.lds_bytes = (int)g.lds;
    ProfScope pp(c, "polar");                                    // (the whole projection is this one kernel)
    const dim3 grid((unsigned)c->B, (unsigned)c->rp_n_slots), block((unsigned)g.threads);
    // registers by launch bound: 512 threads = two waves per SIMD, 256 registers each; 768 (the 97-column orders of config 5) =
    // three, 168 each
    if (g.threads <= 512)
        hipLaunchKernelGGL((k_rproj<512, 2, 4, false, 16>), grid, block, g.lds, c->stream, a);
    else
        hipLaunchKernelGGL((k_rproj<768, 1, RP_ACC_MAX, true, 16>), grid, block, g.lds, c->stream, a);
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) {
        c->err = std::string("k_rproj launch: ") + hipGetErrorString(le);
        return MTIP_EHIP;
    }
    c->vr_kind = 2;
    c->proj_calls += 1;
    return MTIP_OK;
}
