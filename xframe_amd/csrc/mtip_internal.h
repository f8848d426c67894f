// Internal declarations of libmtip_hip.so (gfx950 only).  Public ABI: include/mtip_hip.h
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <map>
#include <string>
#include <vector>
#include "../../include/mtip_hip.h"

// Scheduling fence for values in vector registers: the (empty) statement reads and writes its operands, so loads that
// produce them stay above it and are waited for here, not where the compiler would fold them to.
#ifndef MTIP_PIN_VGPRS4
#define MTIP_PIN_VGPRS4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#endif
// all LDS operations of this wave have completed (diagnostic timers: separates the store drain from the barrier wait)
#ifndef MTIP_WAIT_LDS
#define MTIP_WAIT_LDS() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#endif


// LDS data handed from some lanes of a wave to others of the SAME wave: LDS operations of one wave complete in the order they were
// issued, so only the compiler has to keep them in order (the emulation runs the lanes one after the other and needs a real
// rendezvous: tests/emul defines its own)
#ifndef MTIP_WAVE_LDS_SYNC
#define MTIP_WAVE_LDS_SYNC()                        \
    do {                                            \
        asm volatile("" ::: "memory");              \
        __builtin_amdgcn_wave_barrier();            \
        asm volatile("" ::: "memory");              \
    } while (0)
#endif

typedef double v4f64 __attribute__((vector_size(32)));   // accumulator of v_mfma_f64_16x16x4_f64

// ---- complex helpers (complex128 = double2, interleaved like numpy) ----------------------------
__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 cmulc(double2 a, double2 b) {   // a * conj(b)
    return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 cscale(double2 a, double s) { return make_double2(a.x * s, a.y * s); }
__device__ __forceinline__ double cabs2(double2 a) { return a.x * a.x + a.y * a.y; }
__device__ __forceinline__ int isqrt_lm(int lm) {   // l of index l(l+1)+m, exact integer version
    int l = (int)sqrt((double)lm);
    while (l * l > lm) --l;
    while ((l + 1) * (l + 1) <= lm) ++l;
    return l;
}
// multiply by (-i)^l (sign=-1) or (+i)^l (sign=+1)
__device__ __forceinline__ double2 cmul_ipow(double2 a, int l, int sign) {
    int r = l & 3;
    if (sign < 0) r = (4 - r) & 3;              // (-i)^l = i^(-l)
    switch (r) {
        case 0: return a;
        case 1: return make_double2(-a.y, a.x);   // * i
        case 2: return make_double2(-a.x, -a.y);
        default: return make_double2(a.y, -a.x);  // * -i
    }
}

#define MTIP_CHAIN_DBG_SLOTS 18
#define MTIP_POLAR_DBG_SLOTS MTIP_POLAR_TIMING_SLOTS   // int64 per (restart, order) of the k_rproj timers (include/mtip_hip.h)
// ---- per-restart slot table (device ints), see DESIGN.md "state" ---------------------------------
// SL_HIST: the pair the reference's stale local `hist` ends with (reconstruct.py:859, 913): the input pair of the most
// recent step, or the latest pair when no step has run in the current sub-loop call; read by SW_center (893) and by the
// fixed amplitudes of the *_non_FXS variants (901)
enum { SL_CUR = 0, SL_OUT = 1, SL_BEST = 2, SL_SUP = 3, SL_SUP_BEST = 4, SL_ENFORCE = 5, SL_HAS_ERR = 6, SL_HIST = 7, SL_N = 8 };

// real-space constraint flags (mtip_set_real_constraints)
enum { RC_SUPPORT = 1, RC_VALUE_LO = 2, RC_VALUE_HI = 4, RC_LIMIT_IMAG = 8 };

struct RealParams {
    uint32_t flags, hio_flags;
    double lo, hi, imag_thr;
};

// One grid point of the real-space stage: P = real_projection(w) (support, value bounds, imaginary-part limit;
// fxs_Projections.py:72-130), new density = P (error reduction) or prev - beta (w - P) where a constraint counted
// by the HIO mask was violated (fxs_IO_methods.py:40-64).  Returns the new density, P through P_out.
__device__ __forceinline__ double2 real_update_point(const RealParams& rp, int method, double beta, double2 w, double2 pv,
                                                     bool S, double2& P_out) {
    double2 P = w;
    uint32_t viol = 0;
    if (rp.flags & RC_SUPPORT) {
        if (!S) {
            P = make_double2(0.0, 0.0);
            viol |= RC_SUPPORT;
        }
    }
    if ((rp.flags & RC_VALUE_LO) && (rp.flags & RC_VALUE_HI)) {
        if (P.x < rp.lo) { P.x = rp.lo; viol |= RC_VALUE_LO; }
        if (P.x > rp.hi) { P.x = rp.hi; viol |= RC_VALUE_LO; }
    } else if (rp.flags & RC_VALUE_LO) {
        if (P.x < rp.lo) { P.x = rp.lo; viol |= RC_VALUE_LO; }
    } else if (rp.flags & RC_VALUE_HI) {
        if (P.x > rp.hi) { P.x = rp.hi; viol |= RC_VALUE_LO; }
    }
    if (rp.flags & RC_LIMIT_IMAG) {
        if (fabs(P.y) >= rp.imag_thr) { P.y = 0.0; viol |= RC_LIMIT_IMAG; }
    }
    double2 nw = P;
    if (method == MTIP_HIO || method == MTIP_HIO_NON_FXS) {
        uint32_t hm = rp.hio_flags;
        if (hm & (RC_VALUE_LO | RC_VALUE_HI)) hm |= RC_VALUE_LO;      // both bounds share one mask
        if (viol & hm) {
            nw.x = pv.x - beta * (w.x - P.x);
            nw.y = pv.y - beta * (w.y - P.y);
        }
    }
    P_out = P;
    return nw;
}

// The same point update with the settings decoded once (uniform booleans) and no short-circuit logic: in the chained kernel the
// flag tests of real_update_point became scalar branches around every grid point (157 branches in the kernel, eight points per
// lane and row group).  Bit-identical results: the same comparisons and the same arithmetic in the same order.
struct RealFlags {
    bool support, lo, hi, both, imag, hio, h_support, h_value, h_imag;
};
__device__ __forceinline__ RealFlags real_flags(const RealParams& rp, int method) {
    RealFlags f;
    f.support = (rp.flags & RC_SUPPORT) != 0;
    f.lo = (rp.flags & RC_VALUE_LO) != 0;
    f.hi = (rp.flags & RC_VALUE_HI) != 0;
    f.both = f.lo && f.hi;
    f.imag = (rp.flags & RC_LIMIT_IMAG) != 0;
    f.hio = method == MTIP_HIO || method == MTIP_HIO_NON_FXS;
    uint32_t hm = rp.hio_flags;
    if (hm & (RC_VALUE_LO | RC_VALUE_HI)) hm |= RC_VALUE_LO;          // both bounds share one mask
    f.h_support = (hm & RC_SUPPORT) != 0;
    f.h_value = (hm & RC_VALUE_LO) != 0;
    f.h_imag = (hm & RC_LIMIT_IMAG) != 0;
    return f;
}
__device__ __forceinline__ double2 real_update_point_flat(const RealParams& rp, const RealFlags& f, double beta, double2 w, double2 pv,
                                                          bool S, double2& P_out) {
    double2 P = w;
    const bool v_s = f.support & !S;
    P.x = v_s ? 0.0 : P.x;
    P.y = v_s ? 0.0 : P.y;
    // (with both bounds the upper one is tested on the value the lower one may have replaced, as in real_update_point)
    const bool c_lo = f.lo & (P.x < rp.lo);
    P.x = c_lo ? rp.lo : P.x;
    const bool c_hi = f.hi & (P.x > rp.hi);
    P.x = c_hi ? rp.hi : P.x;
    const bool v_i = f.imag & (fabs(P.y) >= rp.imag_thr);
    P.y = v_i ? 0.0 : P.y;
    const bool take = f.hio & ((f.h_support & v_s) | (f.h_value & (c_lo | c_hi)) | (f.h_imag & v_i));
    double2 nw;
    nw.x = take ? pv.x - beta * (w.x - P.x) : P.x;
    nw.y = take ? pv.y - beta * (w.y - P.y) : P.y;
    P_out = P;
    return nw;
}

// real-space stage fused into the last inverse SHT of a step (EPI_REAL_UPDATE): w = iSHT value (+ previous density on
// shells > 0: the ft_stab add-back), densities and support read / written through the slot table, error partial
// sums per shell
struct RealEpi {
    const double2* prev = nullptr;     // slot base (3,B,G)
    double2* out = nullptr;            // slot base (3,B,G)
    const uint8_t* sup = nullptr;      // support slot base (3,B,G)
    const uint8_t* S0 = nullptr;       // (G)
    const double* wr = nullptr;        // (Nq) radial error weights
    const double* wt = nullptr;        // (nt) polar error weights
    double* partial = nullptr;         // (B, Nq, 2)
    RealParams rp{};
    int method = 0, err_use_mask = 0, add_prev = 0;
    const uint8_t* add_mask = nullptr; // (B) per-restart ft_stab (the add-back only where set), or nullptr: every restart
    double beta = 0.0;
};

// epilogues of the inverse SHT (grid-side fused elementwise stages)
enum { EPI_STORE = 0, EPI_MODULUS = 1, EPI_SCALE_SHELL = 2, EPI_MODULUS_FIXED = 3, EPI_REAL_UPDATE = 4 };

struct ProfEntry { double ms = 0; long long n = 0; };

struct mtip_ctx {
    mtip_cfg cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    int N = 0, L = 0, nt = 0, np = 0, B = 0, nlm = 0, nm = 0, Np = 0;
    size_t G = 0, C = 0;
    // tables
    double *d_cost = nullptr, *d_gw = nullptr, *d_P = nullptr, *d_r = nullptr, *d_q = nullptr;
    int* d_poff = nullptr;
    double2* d_AB = nullptr;                          // (npairs) three-term recurrence coefficients (a_lm, b_lm), (l,m)-major
    double* d_PT = nullptr;                           // (nt/2, npairs) theta-major Legendre table (fused SHT)
    double* d_PTc = nullptr;                          // (nt/2, 768) the same table in the chunk layout of k_sht_chain's Legendre sums
    int* d_lmc = nullptr;                             // (768) l | m << 8 of slot u * 256 + t, -1: none
    uint8_t* d_ftmask = nullptr;                      // (B) per-restart ft_stab of the next runs (mtip_set_ft_stab_mask)
    bool ftmask_mixed = false;                        // the mask has both values: steps with ft_stab take it per restart
    hipEvent_t turn_ev = nullptr;                     // mtip_run_group_async: end of this context's latest transform block
    int chain_chunks = 0;                             // chunks (threads of an accumulation group with work) in that layout; 0: it does not fit 256
    int* d_lmtab = nullptr;                           // (npairs) l | m << 8
    int npairs = 0;
    bool sht_unfused = false;                         // MTIP_SHT_MODE=0: two-kernel SHT (A/B testing)
    int sht_mode = 2;                                 // env MTIP_SHT_MODE: 0 unfused, 1 LDS-Stockham fused, 2 register FFT
    double2* d_twN = nullptr;                         // exp(-2 pi i j / n_phi), j < n_phi
    double2* d_tw = nullptr;
    double* d_W = nullptr;
    int n_cu = 256;                                   // compute units of the device (persistent-grid sizing)
    bool fuse_real_update = true;                     // env MTIP_FUSE_REAL=0: separate coefficient-difference / real-space kernels
    bool sht_wide = true;                             // env MTIP_SHT_WIDE=0: pass-wise inverse Legendre synthesis
    bool jac_resident = true;                         // env MTIP_JAC_RESIDENT=0: round-robin ordering, both columns via LDS
    int *d_jsched = nullptr, *d_jsched_off = nullptr, *d_jsched_rounds = nullptr;   // resident-column pairing schedule
    int jsched_kmax = 0, jsched_ps = 0;
    std::vector<int> jsched_nrd;                      // rounds of a sweep for every column count (host copy of d_jsched_rounds)
    double rp_early = 3e-2, rp_corr2_max = 1.5e-4;     // thresholds of the closing step (k_projr.hip RP_EARLY_CORR, RP_CORR2_MAX; env MTIP_RP_EARLY, MTIP_RP_CORR2_MAX)
    bool rp_corr = true;                              // k_rproj: close the Jacobi sweeps with the first-order polar step (MTIP_RP_CORR=0: classic)
    // non-default reciprocal metrics (k_metrics.hip): flags 1 II_error | 2 ccd_diff | 4 fqc_error, their constant tables, history rows
    uint32_t im_which = 0;
    uint8_t* d_im_zmask = nullptr;
    double2 *d_im_IIref = nullptr, *d_im_ccdref = nullptr;
    double *d_rl2_wr = nullptr, *d_rl2_wt = nullptr, *d_rl2_part = nullptr, *d_rl2_hist = nullptr;   // reciprocal l2_projection_diff (k_metrics.hip)
    double* d_im_fq = nullptr;                         // (B, Nq, Nq) fqc values of a step (k_metric_fqc -> k_metric_fqc_fold)
    double2* d_im_part = nullptr;                      // (B, IM_BLOCKS, 4) partial sums of II_error / ccd_diff
    double *d_im_qq = nullptr, *d_im_ccdT = nullptr, *d_im_P = nullptr, *d_im_refavg = nullptr, *d_im_refw = nullptr, *d_im_hist = nullptr;
    double im_ccd_inv_norm = 0.0;
    int so_order = -1;                                // SO_freedom: order whose unknown [4][2] is made real after every projection (-1: off)
    std::vector<int> jsched_off_h;                    // offsets of the per-column-count tables in d_jsched (host copy of d_jsched_off)
    int* d_jorder = nullptr;                          // active orders, heaviest first (grid of the polar-factor kernel)
    int n_jorder = 0;
    int* d_pg_tiles[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // (order, tile) lists of the projection GEMMs (4, 5: fused pairs)
    int n_pg_tiles[6] = {0, 0, 0, 0, 0, 0};
    long long* d_polar_dbg = nullptr;                 // (B, L+1, MTIP_POLAR_DBG_SLOTS) phase / round timers of k_rproj, allocated by mtip_debug_polar_timing
    int jac_tg = 16;                                  // env MTIP_JAC_TG=8|16: lanes per Jacobi column pair
    bool sht_fwd_pair = true;                         // env MTIP_SHT_FWD_PAIR=0: k_sht_fwd_reg (table loads inside the accumulation loop)
    bool sht_chain_lc = true;                         // env MTIP_SHT_CHAIN_LC=0: run-time L also at L = 32 (A/B of the compile-time instantiation)
    bool sht_chain = true;                            // env MTIP_SHT_CHAIN=0: separate inverse / forward SHT kernels in the fused step (k_sht_chain.hip)
    double2* d_c0n = nullptr;                         // (B, C) SHT of the current density, written by the chained last kernel of a step
    long long* d_chain_dbg = nullptr;                 // (3 kinds, B * Nq, MTIP_CHAIN_DBG_SLOTS) phase stamps of k_sht_chain, allocated by mtip_debug_chain_timing
    bool c0n_valid = false;                           // d_c0n holds SHT(rho[SL_CUR]) of every restart
    void* d_htiles32 = nullptr;                       // workgroup tiles (order, first column) of k_hankel_tile
    int htile_force = 0;                              // env MTIP_HANKEL_CT=1|2|3|5: tile width of k_hankel_tile instead of the occupancy rule (A/B)
    int n_htiles32 = 0, htile_ct = 5;                 // 16-column MFMA tiles per workgroup
    double fwd_scale = 0, inv_scale = 0;
    bool have_angular = false, have_radial = false, have_weights = false, have_support = false, have_errw = false;
    // projection data
    std::vector<int> kl, used, active, voff, xoff, uoff;     // host copies (active = used and V_l != 0)
    int* d_active = nullptr;
    int* d_sweeps = nullptr;                          // (B, L+1) Jacobi sweeps of the last projection (diagnostic)
    bool vr_valid = false;                            // d_Vr holds (complex) right singular vectors of the previous call
    int vr_kind = 0;                                  // 2: d_Vr holds the REAL right singular vectors of the previous k_rproj call
    bool proj_real = true;                            // env MTIP_PROJ_REAL=0: never take the real form of the projection (k_projr.hip)
    std::vector<char> v_real;                         // per order: V_l has no imaginary part
    std::vector<double2> h_V;                         // host copy of the concatenated V_l (tables of the real projection)
    double *d_rp_DV = nullptr, *d_rp_Vt = nullptr;    // q^2 V_l (N x k) and V_l^T (k x N), real, at voff[l]
    int* d_rp_slots = nullptr;                        // (rp_n_slots, rp_slot_len) order | kind << 8 lists of the k_rproj workgroups
    int rp_n_slots = 0, rp_slot_len = 0;
    long long proj_calls = 0;
    double polar_abs_tol = 0.0;                       // 0 = purely relative Jacobi criterion (env MTIP_POLAR_ABS_TOL)
    int *d_kl = nullptr, *d_used = nullptr, *d_voff = nullptr, *d_xoff = nullptr, *d_uoff = nullptr;
    int vtot = 0, xtot = 0, utot = 0;                 // per-restart element counts
    double2* d_V = nullptr;                           // concatenated V_l, (Nq, k_l) row-major each
    uint8_t* d_rmask = nullptr;                       // (L+1, Nq)
    std::vector<char> have_V;
    double n_particles = 1.0;
    // deg2 metric
    int deg2_enable = 0;
    double2* d_Bref = nullptr;                        // (L+1, Nq, Nq) masked reference B_l
    double* d_Bnorm = nullptr;                        // (L+1)
    double* d_deg2_part = nullptr;                    // (B, L+1, (Nq/16)^2) per-tile partial sums of the B_l metric
    bool proj_fuse = true;                            // env MTIP_PROJ_FUSE=0: four separate projection products instead of the two fused pairs
    bool deg2_simple = false;                         // env MTIP_DEG2_SIMPLE=1: one thread per B_l element instead of MFMA tiles
    bool bref_dirty = true;
    // real-space constraints and error metric
    RealParams rp{RC_SUPPORT | RC_VALUE_LO, RC_SUPPORT | RC_VALUE_LO, 0.0, 0.0, 0.0};
    uint8_t *d_S0 = nullptr, *d_sup = nullptr;        // (G), (3, B, G)
    uint16_t* d_mk = nullptr;                         // (3, B, Nq, nt, R2) the same masks packed for k_sht_chain: bit n1 = support, bit 8 + n1 = S0 of point R2 n1 + n2 of the row
    double *d_err_wr = nullptr, *d_err_wt = nullptr;
    int err_use_mask = 1;
    // state
    double2 *d_rho = nullptr, *d_Fp = nullptr;        // (3, B, G) each
    int* d_slot = nullptr;                            // (B, SL_N)
    double *d_best_err = nullptr, *d_last_err = nullptr;   // (B)
    double* d_op_err = nullptr;                       // (B) error of the single-operator entry point (never the loop's)
    double* d_gq = nullptr;                           // (Nq) shrink-wrap Gaussian G_sigma(q)
    double* d_err_hist = nullptr;                     // (cap, B) real l2 metric per step
    double* d_main_hist = nullptr;                    // (cap, B) main error per step when it is not the real metric (main_mode 1)
    int main_mode = 0, main_type = 0;                 // mtip_set_main_error
    double* d_deg2_hist = nullptr;                    // (cap, B, L+1)
    long long err_cap = 0, n_steps_done = 0;
    bool state_ready = false, fixed_valid = false;
    // work buffers
    double2 *d_F = nullptr, *d_T1 = nullptr, *d_T2 = nullptr;   // grids (B, G)
    double* d_fixed = nullptr;                        // real grid (B, G)
    double2* d_g = nullptr;                           // (B, Nq, nt, 2L+1)
    double2* d_c[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // coefficient arrays (B, C)
    double2 *d_X = nullptr, *d_Vr = nullptr, *d_U = nullptr;   // projection workspaces
    double* d_partial = nullptr;                      // (B, nblk, 2) error partial sums
    int n_partial_blocks = 0;
    double* d_minmax = nullptr;                       // (B, nblk, 2)
    double2* d_Bl = nullptr;                          // (B, L+1, Nq, Nq) scratch (lazy)
    // rotational alignment (k_align.hip): Wigner table d^l_mn(beta_b), DFT twiddles, work arrays
    double* d_so3_d = nullptr;
    double2 *d_so3_tw = nullptr, *d_so3_T = nullptr, *d_so3_S = nullptr, *d_so3_P = nullptr, *d_so3_D = nullptr;
    double* d_so3_C = nullptr;
    int so3_bw = 0;
    // host staging
    void* h_stage = nullptr;
    // profiling
    int prof = 0;
    std::map<std::string, ProfEntry> prof_data;
    // asynchronous family timers: event pairs recorded on the ctx stream, resolved when the numbers are read
    std::vector<hipEvent_t> prof_events;
    std::vector<std::pair<const char*, int>> prof_pending;   // (family, index of the first event of the pair)
    int prof_next = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

// ---- launchers (defined next to their kernels) ------------------------------------------------------
// SHT
// in_slot >= 0: grid is a (3,B,G) slot array read through slot[b][in_slot]
void launch_sht_forward(mtip_ctx* c, const double2* grid, double2* coeff, int prologue, int in_slot = -1);
struct InvEpilogue {
    int mode = EPI_STORE;
    const double2* F = nullptr;        // EPI_MODULUS*: reciprocal density to rescale
    const double* fixed = nullptr;     // EPI_MODULUS_FIXED
    const double* shell_scale = nullptr;   // EPI_SCALE_SHELL (Nq)
    const double2* coeff_sub = nullptr;    // coefficients to subtract on shells > 0 before the synthesis (ft_stab)
    RealEpi real;                          // EPI_REAL_UPDATE
    int out_slot = -1;                 // >= 0: grid is a (3,B,G) slot array written through slot[b][out_slot]
};
void launch_sht_inverse(mtip_ctx* c, const double2* coeff, double2* grid, const InvEpilogue& epi);
void build_legendre_tables(mtip_ctx* c, const double* cos_theta);
bool sht_fused_supported(const mtip_ctx* c);
void launch_sht_forward_fused(mtip_ctx* c, const double2* grid, double2* coeff, int prologue, int in_slot);
void launch_sht_inverse_fused(mtip_ctx* c, const double2* coeff, double2* grid, const InvEpilogue& epi);
bool sht_reg_supported(const mtip_ctx* c);
bool sht_inverse_fuses_real_update(const mtip_ctx* c);    // EPI_REAL_UPDATE / coeff_sub available (wide inverse kernel)
int sht_inverse_real_update_blocks(const mtip_ctx* c);   // error partial sums per restart written by that epilogue
void launch_sht_forward_reg(mtip_ctx* c, const double2* grid, double2* coeff, int prologue, int in_slot);
void launch_sht_inverse_reg(mtip_ctx* c, const double2* coeff, double2* grid, const InvEpilogue& epi);
// k_sht_chain.hip: grid = epilogue(iSHT(coeff)) and coeff_out = SHT(prologue(grid)) in one kernel (EPI_STORE with
// MTIP_PRE_NONE / MTIP_PRE_SQUARE, EPI_MODULUS, EPI_REAL_UPDATE without coeff_sub)
bool sht_chain_supported(const mtip_ctx* c);
void launch_sht_chain(mtip_ctx* c, const double2* coeff, double2* grid, const InvEpilogue& epi, int prologue, double2* coeff_out);
// Hankel
void launch_hankel(mtip_ctx* c, const double2* in, double2* out, int inverse);
bool hankel_has_difference(const mtip_ctx* c);       // launch_hankel_mfma_sub is available (workgroup-tiled kernel)
void launch_hankel_mfma_sub(mtip_ctx* c, const double2* in, const double2* in_sub, double2* out, int inverse, const uint8_t* sub_mask = nullptr);
int build_jacobi_schedule(mtip_ctx* c, int kmax);    // k_proj.hip: resident-column pairing schedule, verified on the host
int jacobi_groups(int k);                            // pair-groups a round of that schedule keeps busy for k columns (<= jsched_ps: the table's row length)
int build_hankel_tiles(mtip_ctx* c);
int launch_invariant_metrics(mtip_ctx* c, const double2* Ilm, long long step);
void free_invariant_metrics(mtip_ctx* c);
int launch_reciprocal_l2_metric(mtip_ctx* c, const double2* F, const double2* Fp, long long step);
void launch_deg2(mtip_ctx* c, const double2* Ilm, double2* Bl);
void launch_coeff_diff(mtip_ctx* c, const double2* a, const double2* b, double2* out);
// reciprocal projection
// real_intensity: the caller guarantees I_{l,-m} = (-1)^m conj(I_{l,m}) (coefficients of a real grid, as in the phasing
// loop): with real V_l the projection then takes its real form (k_projr.hip)
int launch_project_coefficients(mtip_ctx* c, const double2* Ilm, double2* out, bool real_intensity = false);   // MTIP_OK or an error code (c->err set)
bool rproj_supported(mtip_ctx* c);                    // k_projr.hip
int launch_rproj(mtip_ctx* c, double2* coef);         // in place
void free_rproj_tables(mtip_ctx* c);
int launch_apply_unknowns(mtip_ctx* c, const double2* Ilm, double2* out);
void launch_deg2(mtip_ctx* c, const double2* Ilm, double2* Bl);
void launch_deg2_metric(mtip_ctx* c, const double2* Ilm, double* out /*(B, L+1)*/);
// elementwise / reductions
// rho_p = IFT(F') (B,G); prev/out: slot arrays (3,B,G) when use_slots else plain (B,G); rho_rt may be null
void launch_real_update(mtip_ctx* c, const double2* rho_p, const double2* prev, const double2* rho_rt, double2* out,
                        int method, double beta, int use_slots);
// step_index >= 0: loop step (history, best tracking, slot rotation); < 0: only the error, written to c->d_op_err
void launch_finish_step(mtip_ctx* c, long long step_index, int nblk = 0);   // nblk partial sums per restart (0: grid blocks)
void launch_abs_to_fixed(mtip_ctx* c);
void launch_modulus_plain(mtip_ctx* c, const double2* F, const double2* Inew, double2* out);
void launch_modulus_fixed_slots(mtip_ctx* c, const double2* F);
void launch_copy_to_slot(mtip_ctx* c, const double2* src, double2* dst_slots, int which);
void launch_sw_clamp(mtip_ctx* c, const double2* conv, double* tmp_real);
void launch_sw_threshold(mtip_ctx* c, const double* tmp_real, double threshold, double error_limit);
void launch_pack_masks(mtip_ctx* c);                  // d_sup, d_S0 -> d_mk (all slots; after every writer of the two)
void launch_apply_matrix(mtip_ctx* c, const double* M, const double* x, double* y, int nr, int nc, int nv);

// ---- small utilities ---------------------------------------------------------------------------------
#define MTIP_HIP_CHECK(c, call)                                                              \
    do {                                                                                     \
        hipError_t e__ = (call);                                                             \
        if (e__ != hipSuccess) {                                                             \
            (c)->err = std::string(#call) + ": " + hipGetErrorString(e__);                   \
            return MTIP_EHIP;                                                                \
        }                                                                                    \
    } while (0)

void prof_flush(mtip_ctx* c);       // mtip_api.hip: synchronise the stream and fold the pending event pairs into prof_data

// hipEvent bracket of one kernel family launch.  Nothing waits here: the pair is resolved by prof_flush (called when
// the numbers are read, or when 4096 pairs are pending), so the brackets can stay on during a timed run.
struct ProfScope {
    mtip_ctx* c;
    const char* name;
    int idx = -1;
    ProfScope(mtip_ctx* c_, const char* n) : c(c_), name(n) {
        if (!c->prof) return;
        if (c->prof_next + 2 > 8192) prof_flush(c);
        while ((int)c->prof_events.size() < c->prof_next + 2) {
            hipEvent_t e;
            // no system-scope fence at the record: a default event releases to the host (cache write-back) every time it is
            // recorded -- eighteen of those made a bracketed step twice as long; the timestamps do not need it
            if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) return;
            c->prof_events.push_back(e);
        }
        idx = c->prof_next;
        c->prof_next += 2;
        (void)hipEventRecord(c->prof_events[idx], c->stream);
    }
    ~ProfScope() {
        if (idx < 0) return;
        (void)hipEventRecord(c->prof_events[idx + 1], c->stream);
        c->prof_pending.emplace_back(name, idx);
    }
};

// The context's stream is created hipStreamNonBlocking: it does not synchronise with the null stream, so a blocking copy of one
// engine no longer stalls the kernels of the other engines of the process (measured in the worker: -20 % loop time).  The price:
// nothing orders a null-stream copy against the stream's kernels any more -- every blocking copy therefore first waits for the
// stream (free when it is idle, as in all setters), and memsets / device-to-device copies are enqueued ON the stream.
// The caller's side of a copy (`kind` names the direction the ABI documents) may itself be device memory -- the averaging keeps
// its batch in HBM between operator calls -- so the direction is left to the runtime (unified addressing); a device-to-device
// hipMemcpy is ordered on the null stream, which this context's stream does not wait for: wait for it here.
static inline hipError_t mtip_copy(mtip_ctx* c, void* dst, const void* src, size_t n, hipMemcpyKind kind) {
    (void)kind;
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return e;
    e = hipMemcpy(dst, src, n, hipMemcpyDefault);
    return e != hipSuccess ? e : hipStreamSynchronize(nullptr);
}

static inline int div_up(long long a, long long b) { return (int)((a + b - 1) / b); }
