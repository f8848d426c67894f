// Legendre synthesis of the rows of a shell into LDS spectra by the in-register three-term recurrence -- the first half of
// the inverse spherical-harmonic transform (sh.inverse_d, shtns_plugin.py:250-261), shared by k_sht_inv_wide (k_sht_reg.hip)
// and k_sht_chain (k_sht_chain.hip).
//
// The wave is the unit of work: an item = (pair of orders m = 2p, 2p + 1; chunk of 32 thetas).  Lanes 0-31 run the recurrence
// P_lm = a_lm (x P_l-1,m - b_lm P_l-2,m) of m = 2p for their theta, lanes 32-63 that of m = 2p + 1, and every lane accumulates
// the rows of +m AND -m (P_lm is the same for both); a_lm, b_lm sit in LDS, the two start values P_mm, P_m+1,m come from the
// table and are prefetched one item ahead, so the loop touches no global memory.  Even and odd l - m accumulate separately
// (north = E + O, south = E - O).  Items are dealt to the waves in snake order of decreasing length.
#pragma once
#include "mtip_internal.h"

struct LegendreStart {
    double pmm = 0.0, pm1 = 0.0;       // start values of the wave's next item
};

// item index of round kk for this wave (snake order)
__device__ __forceinline__ int legendre_item(int kk, int nw, int wave) { return kk * nw + ((kk & 1) ? nw - 1 - wave : wave); }

__device__ __forceinline__ void legendre_load_start(LegendreStart& s, const double* __restrict__ P, int nt, int L, int nth, int j0,
                                                    int n_chunks, int item, int lane) {
    const int jj = lane & 31, half_id = lane >> 5;
    const int mp = item / n_chunks, ch = item - mp * n_chunks;
    const int m = min(2 * mp + half_id, L);
    const int j = ch * 32 + jj;
    const int jc = j0 + (j < nth ? j : nth - 1);
    const double* pcol = P + (size_t)(m * (L + 1) - m * (m - 1) / 2) * nt + jc;
    s.pmm = pcol[0];
    s.pm1 = m < L ? pcol[nt] : 0.0;
}

// start values of the wave's first item: issued before the tables are staged, so that they are in flight meanwhile
// (nth theta pairs of this workgroup, the first one j0)
__device__ __forceinline__ void legendre_prefetch_first(LegendreStart& s, const double* __restrict__ P, int nt, int L, int nth, int j0,
                                                        int wave, int lane) {
    const int n_chunks = (nth + 31) >> 5;
    const int n_items = ((L + 2) >> 1) * n_chunks;
    if (wave < n_items) legendre_load_start(s, P, nt, L, nth, j0, n_chunks, wave, lane);
}

// Gs: spectra, row 2j = theta_(j0+j), row 2j+1 = its mirror, 2L+1 entries per row (m = -L..L); cl: the shell's coefficients
// (index l(l+1)+m); ABs: recurrence coefficients (a_lm, b_lm), (l,m)-major; cost: cos(theta_j).  All of Gs that belongs to
// the workgroup's rows is written; the caller synchronises afterwards.
__device__ __forceinline__ void legendre_synthesis_rows(LegendreStart& s, double2* __restrict__ Gs, const double2* __restrict__ cl,
                                                        const double2* __restrict__ ABs, const double* __restrict__ P,
                                                        const double* __restrict__ cost, int nt, int L, int nth, int j0,
                                                        int wave, int nw, int lane) {
    const int nm = 2 * L + 1;
    const int jj = lane & 31, half_id = lane >> 5;
    const int n_chunks = (nth + 31) >> 5;
    const int n_items = ((L + 2) >> 1) * n_chunks;
    for (int kk = 0;; ++kk) {
        const int i = legendre_item(kk, nw, wave);
        if (i >= n_items) break;
        const int mp = i / n_chunks, ch = i - mp * n_chunks;
        const int m_a = 2 * mp;                                  // the smaller order of the pair: sets the trip count of the wave
        const bool m_ok = m_a + half_id <= L;
        const int m = min(m_a + half_id, L);                     // this lane's order (clamped: an odd L + 1 has no partner)
        const int j = ch * 32 + jj;
        const bool act = (j < nth) && m_ok;
        const int jc = j0 + (j < nth ? j : nth - 1);
        const double x = cost[jc];
        double p2 = s.pmm, p1 = s.pm1;
        {   // prefetch the start values of this wave's next item
            const int i2 = legendre_item(kk + 1, nw, wave);
            if (i2 < n_items) legendre_load_start(s, P, nt, L, nth, j0, n_chunks, i2, lane);
        }
        const double2* cp = cl + m;                              // c_l,+m at cp[l (l + 1)]
        const double2* cm = cl - m;                              // c_l,-m
        const double2* abm = ABs + (m * (L + 1) - m * (m - 1) / 2) - m;   // abm[l]
        double2 Ep, Em, Op = make_double2(0.0, 0.0), Om = make_double2(0.0, 0.0);
        {
            const double2 a = cp[m * (m + 1)], b = cm[m * (m + 1)];
            Ep = make_double2(p2 * a.x, p2 * a.y);
            Em = make_double2(p2 * b.x, p2 * b.y);
        }
        if (m < L) {
            const double2 a = cp[(m + 1) * (m + 2)], b = cm[(m + 1) * (m + 2)];
            Op = make_double2(p1 * a.x, p1 * a.y);
            Om = make_double2(p1 * b.x, p1 * b.y);
        }
        // The recurrence is a dependent chain and there are only two waves per SIMD: with the operands read at the top of
        // each iteration an LDS round trip per iteration was most of the loop.  Two operand sets, A and B, alternate; a set is
        // requested before the other one is used (clamped indices, branch-free) and the empty asm pins it there.  The wave runs
        // the iterations of its smaller order; a lane whose order is one larger sits out the last one when its l runs out.
        int l = m + 2;                                           // per lane
        const int n_it = m_a + 2 <= L ? (L - m_a - 1) >> 1 : 0;  // double steps of the wave (uniform)
        double2 Aab0, Aab1, Acp, Aop, Acm, Aom, Bab0, Bab1, Bcp, Bop, Bcm, Bom;
#define LEG_LOAD(S, lq_)                                             \
        {                                                            \
            const int q_ = max(min((lq_), L - 1), 0);                \
            const int o_ = min((q_ + 1) * (q_ + 2), L * (L + 1));    \
            S##ab0 = abm[q_];                                        \
            S##ab1 = abm[q_ + 1];                                    \
            S##cp = cp[q_ * (q_ + 1)];                               \
            S##op = cp[o_];                                          \
            S##cm = cm[q_ * (q_ + 1)];                               \
            S##om = cm[o_];                                          \
        }
#define LEG_STEP(S)                                                         \
        if (l + 1 <= L) {                                                   \
            const double pa = S##ab0.x * (x * p1 - S##ab0.y * p2);          \
            const double pb = S##ab1.x * (x * pa - S##ab1.y * p1);          \
            Ep.x = fma(pa, S##cp.x, Ep.x); Ep.y = fma(pa, S##cp.y, Ep.y);   \
            Op.x = fma(pb, S##op.x, Op.x); Op.y = fma(pb, S##op.y, Op.y);   \
            Em.x = fma(pa, S##cm.x, Em.x); Em.y = fma(pa, S##cm.y, Em.y);   \
            Om.x = fma(pb, S##om.x, Om.x); Om.y = fma(pb, S##om.y, Om.y);   \
            p2 = pa;                                                        \
            p1 = pb;                                                        \
            l += 2;                                                         \
        }
#define LEG_PIN(S)                                                  \
        MTIP_PIN_VGPRS4(S##ab0.x, S##ab0.y, S##ab1.x, S##ab1.y)     \
        MTIP_PIN_VGPRS4(S##cp.x, S##cp.y, S##op.x, S##op.y)         \
        MTIP_PIN_VGPRS4(S##cm.x, S##cm.y, S##om.x, S##om.y)
        LEG_LOAD(A, l)
        for (int it = 0; it < n_it;) {
            LEG_LOAD(B, l + 2)
            LEG_STEP(A)
            LEG_PIN(B)
            if (++it >= n_it) break;
            LEG_LOAD(A, l + 2)
            LEG_STEP(B)
            LEG_PIN(A)
            ++it;
        }
#undef LEG_LOAD
#undef LEG_STEP
#undef LEG_PIN
        if (l <= L) {
            const double2 ab0 = abm[l];
            const double2 a = cp[l * (l + 1)], b = cm[l * (l + 1)];
            const double pa = ab0.x * (x * p1 - ab0.y * p2);
            Ep.x = fma(pa, a.x, Ep.x); Ep.y = fma(pa, a.y, Ep.y);
            Em.x = fma(pa, b.x, Em.x); Em.y = fma(pa, b.y, Em.y);
        }
        if (act) {
            double2* g_p = Gs + (size_t)(2 * j) * nm + L + m;
            g_p[0] = make_double2(Ep.x + Op.x, Ep.y + Op.y);
            g_p[nm] = make_double2(Ep.x - Op.x, Ep.y - Op.y);
            if (m > 0) {
                const double sg = (m & 1) ? -1.0 : 1.0;            // Y_l,-m = (-1)^m conj(Y_lm)
                double2* g_m = Gs + (size_t)(2 * j) * nm + L - m;
                g_m[0] = make_double2(sg * (Em.x + Om.x), sg * (Em.y + Om.y));
                g_m[nm] = make_double2(sg * (Em.x - Om.x), sg * (Em.y - Om.y));
            }
        }
    }
}
