/* mtip_hip.h -- C ABI of libmtip_hip.so, the MI355X (gfx950) MTIP phasing engine.
 *
 * This is the drop-in boundary for the hot path of European-XFEL/xFrame's
 * `xframe/projects/fxs/reconstruct.py` (reference file:line cited per entry point; paths are
 * relative to the reference checkout).  The reference binds its GPU code through PyOpenCL
 * (`xframe/externalLibraries/openCL_plugin.py:154-226`) behind
 * `Multiprocessing.comm_module.add_gpu_process` (`xframe/control/communicators.py:79-82`);
 * a maintainer replaces that with a ctypes binding of the functions below.
 *
 * Conventions
 *   - every function returns 0 on success or a negative MTIP_E* code; nothing throws across the ABI;
 *     `mtip_last_error(ctx)` returns a human readable message (ctx may be NULL for create errors: the message of the
 *     last failed mtip_create on the calling thread);
 *   - host buffers are caller-owned, C-contiguous; complex128 = interleaved (re,im) doubles
 *     (numpy complex128); masks are uint8 (numpy bool); device memory is library-owned;
 *   - a ctx is bound to one device and one hipStream_t; it is not thread-safe; several ctxs may
 *     be used concurrently (one per GPU / per process, or several per GPU from different host threads);
 *     the stream is created hipStreamNonBlocking: it does NOT synchronise with the null stream, so a caller's own
 *     null-stream work is not ordered against a ctx's kernels (call mtip_synchronize); every entry point that moves
 *     data to or from the host waits for the ctx's own stream itself and returns with the copy complete;
 *   - array shapes: grid  (n_batch, Nq, n_theta, n_phi);  coeff 'direct' (n_batch, Nq, (L+1)^2)
 *     with index l(l+1)+m  (`xframe/externalLibraries/shtns_plugin.py:24,105-114,250-261`).
 */
#ifndef MTIP_HIP_H
#define MTIP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mtip_ctx mtip_ctx;
typedef struct { double re, im; } mtip_cdouble;

enum {
    MTIP_OK = 0,
    MTIP_EINVAL = -1,      /* bad argument / shape                        */
    MTIP_ENODEV = -2,      /* no usable HIP device                        */
    MTIP_EHIP = -3,        /* a HIP runtime call or kernel launch failed  */
    MTIP_ENOMEM = -4,
    MTIP_ESTATE = -5       /* call order violated (e.g. run before setup) */
};

/* phasing methods: the sketches of reconstruct.py:565-596 */
enum { MTIP_HIO = 0, MTIP_ER = 1, MTIP_HIO_NON_FXS = 2, MTIP_ER_NON_FXS = 3 };

/* SHT load-side prologues (fused elementwise stages) */
enum { MTIP_PRE_NONE = 0, MTIP_PRE_SQUARE = 1 /* |x|^2, misk.py:159-168 */, MTIP_PRE_ABS = 2 /* |x|, misk.py:221-225 */ };

typedef struct {
    int32_t n_radial;      /* Nq: grid.n_radial_points                                   */
    int32_t l_max;         /* L : grid.max_order                                         */
    int32_t n_theta;       /* Gauss-Legendre nodes (shtns_plugin.py:94-101 for defaults) */
    int32_t n_phi;         /* power of two, > 2L                                         */
    int32_t n_batch;       /* restarts resident in this ctx (>=1)                        */
    int32_t hankel_trapz;  /* 0: midpoint/gauss rule (hankel_transforms.py:702-731);
                              1: trapz/Zernike (671-700: weights have Nq-1 rows, reads f[p+1]) */
    int32_t fused;         /* 1: algebraically fused step, 0: reference operator order */
    int32_t reserved;
} mtip_cfg;

/* ---- life cycle ------------------------------------------------------------------------------
 * replaces Multiprocessing.get_number_of_gpus (xframe/Multiprocessing.py:892-898) and the
 * OpenCL context/buffer creation of openCL_plugin.py:28-61,118-152 */
int mtip_device_count(void);
mtip_ctx* mtip_create(const mtip_cfg* cfg, int device);
void mtip_destroy(mtip_ctx* ctx);
const char* mtip_last_error(const mtip_ctx* ctx);
int mtip_get_cfg(const mtip_ctx* ctx, mtip_cfg* out);
int mtip_synchronize(mtip_ctx* ctx);

/* ---- one-off setup (host side of rows a1, a2, a4, a7 of SURVEY section 8) -------------------- */
/* cos(theta) north->south and Gauss weights, n_theta each (shtns_plugin.py:130-133). The library
 * builds the orthonormal Condon-Shortley Legendre tables itself. */
int mtip_set_angular_grid(mtip_ctx* ctx, const double* cos_theta, const double* gauss_weights);
/* radial points r_p, q_k (ft_grid_pairs.py:282-291), Nq each */
int mtip_set_radial_grid(mtip_ctx* ctx, const double* r, const double* q);
/* raw real Hankel weights w[l][p][k] exactly as calc_spherical_mid_weights / _trapz_weights return
 * them (hankel_transforms.py:399-410 / 322-333), shape (L+1, Np, Nq) with Np = Nq (midpoint) or Nq-1
 * (trapz); fwd_scale=(r_max/N)^3 sqrt(2/pi), inv_scale=(q_max/N)^3 sqrt(2/pi) (assemble_weights_mid,
 * 426-452); the (-i)^l / (+i)^l phases are applied in the kernel epilogue. */
int mtip_set_hankel_weights(mtip_ctx* ctx, const double* w_raw, double fwd_scale, double inv_scale);
/* modified projection matrix V_l (fxs_Projections.py:679-714), shape (Nq, k_l) row-major, its radial
 * mask row (578-629, Nq bytes) and whether order l takes part in the projection (used_orders, 492). */
int mtip_set_projection_matrix(mtip_ctx* ctx, int l, const mtip_cdouble* V, int k_l,
                               const uint8_t* radial_mask, int used);
int mtip_set_number_of_particles(mtip_ctx* ctx, double n_particles);   /* fxs_Projections.py:497,870 */
/* projections.reciprocal.SO_freedom (fxs_Projections.py:493, 768-780): after every approximate_unknowns the element [4][2] of the
 * unknowns of `order` loses its imaginary part before the unknowns are applied (the host ranks the orders,
 * fxs_invariant_tools.py:1467-1486); order = -1 switches it off (default) */
int mtip_set_so_freedom(mtip_ctx* ctx, int order);
/* reference B_l = V_l V_l^+ masks for the deg2_invariant_l2_diff metric are derived internally
 * (fxs_IO_methods.py:408-447); enable = 1 evaluates it every step. */
int mtip_set_deg2_metric(mtip_ctx* ctx, int enable);
/* generate_main_error_routine (fxs_IO_methods.py:746-765): the main error of a step is `type` (0 mean, 1 min, 2 max,
 * 3 prod) over the last values of the chosen metrics.  use_reciprocal_deg2 = 0: the real l2_projection_diff metric (a
 * scalar: every type returns it); 1: the per-order values of deg2_invariant_l2_diff (one entry per used order, -1 where
 * the reference invariant vanishes).  Mixing both makes the reference itself raise (np.array of a scalar and a vector). */
int mtip_set_main_error(mtip_ctx* ctx, int use_reciprocal_deg2, int type);
/* the non-default reciprocal metrics II_error / ccd_diff / fqc_error (fxs_IO_methods.py:587-627, 651-683, 507-550), evaluated in
 * every FXS step from B_l of the current intensity coefficients.  which: 1 | 2 | 4 (0 switches them off); zero_mask (L+1, Nq, Nq):
 * 1 = entry of B_l outside the invariant mask (zeroed); II: masked reference sum_{l>=1} B_l^ref (Nq, Nq), qq = (q q')^2;
 * ccd: P^C_l(q) P^C_l(q') / (2l+1) per order with the rows of order 0 and of orders < C_order zero (L+1, Nq, Nq), reference C (Nq, Nq),
 * its squared norm; fqc: P (L+1, Nq, Nq, L+1) = P^m_l(q) P^m_l(q') / (2l+1) indexed [l][q][q'][m], reference average (Nq, Nq),
 * reference weights (L+1, Nq, Nq).  The host prepares the tables (xframe_amd/fxs/hostsetup.py). */
int mtip_set_invariant_metrics(mtip_ctx* ctx, uint32_t which, const uint8_t* zero_mask, const mtip_cdouble* II_reference, const double* qq,
                               const double* ccd_weights, const mtip_cdouble* ccd_reference, double ccd_norm, const double* fqc_P,
                               const double* fqc_reference_average, const double* fqc_reference_weights);
/* their values for the steps [first, first + n_steps): II (n, n_batch), ccd (n, n_batch), fqc (n, n_batch, Nq); NULL skips one */
int mtip_fetch_invariant_metrics(mtip_ctx* ctx, int64_t first_step, int64_t n_steps, double* II, double* ccd, double* fqc);
/* the same metrics of given intensity coefficients (n_batch, Nq, (L+1)^2): II (n_batch), ccd (n_batch), fqc (n_batch, Nq) */
int mtip_op_invariant_metrics(mtip_ctx* ctx, const mtip_cdouble* Ilm, double* II, double* ccd, double* fqc);
/* the reciprocal metric l2_projection_diff (fxs_IO_methods.py:301-310, 96-127, 131-205): per FXS step int |F - F'|^2 / int |F|^2 with the
 * integrator weights radial_w (Nq), theta_w (n_theta) -- shell Nq - 2 zeroed by the caller, the reference's `square[~True] = 0`; NULL
 * weights switch the metric off.  fetch: the steps [first, first + n_steps) as (n, n_batch). */
int mtip_set_reciprocal_l2_metric(mtip_ctx* ctx, const double* radial_w, const double* theta_w);
int mtip_fetch_reciprocal_l2_metric(mtip_ctx* ctx, int64_t first_step, int64_t n_steps, double* out);
/* real-space constraints (fxs_Projections.py:72-130, pythonLibrary.py:1289-1320):
 * flags bit0 support, bit1 value lower bound, bit2 value upper bound, bit3 limit_imag;
 * hio_mask_flags: which of those feed the HIO mask gamma ('considered_projections',
 * fxs_IO_methods.py:40-64; ['all'] = same as flags). */
int mtip_set_real_constraints(mtip_ctx* ctx, uint32_t flags, double value_lo, double value_hi,
                              double imag_threshold, uint32_t hio_mask_flags);
/* initial support S0 (fxs_Projections.py:133-155), (Nq, n_theta, n_phi) bytes, shared by the batch;
 * resets every restart's support to S0 and enforce_initial_support to 1 (fxs_Projections.py:34). */
int mtip_set_initial_support(mtip_ctx* ctx, const uint8_t* support);
/* error metric weights (fxs_IO_methods.py:97-128 with mathLibrary.py:1223-1235):
 * E = sum m(x) wr[q] wt[theta] |w-P|^2 / sum m(x) wr[q] wt[theta] |w|^2 ;
 * use_initial_support_mask: m = S0, else m = 1. */
int mtip_set_error_weights(mtip_ctx* ctx, const double* radial_w, const double* theta_w,
                           int use_initial_support_mask);

/* ---- state ------------------------------------------------------------------------------------ */
/* inject a starting density for restart `batch` (reconstruct.py:957-966: rho0 -> F0=FT(rho0),
 * rho0'=IFT(F0) is done by mtip_init_state) */
/* The guesses are staged in device scratch that every mtip_op_* call may overwrite: set all restarts, then call
 * mtip_init_state, with no mtip_op_* call in between. */
int mtip_set_density(mtip_ctx* ctx, int batch, const mtip_cdouble* rho);
int mtip_init_state(mtip_ctx* ctx);
/* which = 0 latest pair, 1 best pair (reconstruct.py:934-938) */
int mtip_get_density(mtip_ctx* ctx, int batch, int which, mtip_cdouble* rho);
int mtip_get_reciprocal_density(mtip_ctx* ctx, int batch, int which, mtip_cdouble* F);
int mtip_get_support(mtip_ctx* ctx, int batch, int which, uint8_t* support);
int mtip_set_support(mtip_ctx* ctx, int batch, const uint8_t* support, int enforce_initial_support);
/* fxs_unknowns U_l of the last step (k_l x (2l+1), row-major), fxs_Projections.py:752-767 */
int mtip_get_unknowns(mtip_ctx* ctx, int batch, int l, mtip_cdouble* U);
/* best_error per restart (n_batch doubles) and the number of steps done */
int mtip_get_best_error(mtip_ctx* ctx, double* best_error, int64_t* n_steps_done);
/* take the best pair as the latest one (reconstruct.py:945-949) */
int mtip_select_best(mtip_ctx* ctx);
/* the same for the restarts with select[b] != 0 only (the reference decides per reconstruction process); NULL = all */
int mtip_select_best_where(mtip_ctx* ctx, const uint8_t* select);

/* ---- the loop (reconstruct.py:854-951) ------------------------------------------------------------
 * runs n_steps steps of `method` for every restart without host synchronisation; betas[n_steps] is the
 * HIO beta per step (ExponentialRamp, reconstruct.py:911).  real_err (n_steps x n_batch, may be NULL)
 * receives the l2_projection_diff error per step; deg2_err (n_steps x n_batch x (L+1), may be NULL)
 * the deg2_invariant_l2_diff metric when enabled. The call returns after the results are on the host. */
int mtip_run(mtip_ctx* ctx, int method, int ft_stab, int n_steps, const double* betas,
             double* real_err, double* deg2_err);
/* same, but only enqueues (no download, no sync): errors stay on the device until mtip_fetch_errors */
int mtip_run_async(mtip_ctx* ctx, int method, int ft_stab, int n_steps, const double* betas);
/* mtip_run_async for the n_ctx contexts of ONE device that a worker runs side by side (the restart groups of
 * reconstruct.py:104 `n_gpu_workers`, one stream each): the same steps, every context's results bit-identical to its own
 * mtip_run_async, enqueued so that the contexts take turns at the transforms of a step (which fill the chip) while the
 * projections of the others (a long chain on a few CUs) run beside them -- see mtip_api.hip.  All contexts must be in the
 * same loop state a mtip_run_async call would need; n_ctx = 1 is mtip_run_async.  On an error return the failing context holds
 * the message (mtip_last_error) and the contexts may have advanced by different numbers of steps (mtip_get_best_error reports each). */
int mtip_run_group_async(mtip_ctx* const* ctxs, int n_ctx, int method, int ft_stab, int n_steps, const double* betas);
/* ft_stab per restart for the runs that follow (ft_stab = 1 in mtip_run*): mask[n_batch] != 0 = this restart takes the
 * add-back.  The reference decides `ft_stab: link_to_enforce_initial_support` per reconstruction process
 * (reconstruct.py:836-850), so restarts of one batch can disagree.  NULL (default) or all set = every restart; needs the
 * fused step (cfg.fused = 1) when the values differ. */
int mtip_set_ft_stab_mask(mtip_ctx* ctx, const uint8_t* mask);
int mtip_fetch_errors(mtip_ctx* ctx, int64_t first_step, int64_t n_steps, double* real_err, double* deg2_err);
/* the main error per step (n_steps x n_batch): what best-pair tracking and the enforce_initial_support decision use */
int mtip_fetch_main_errors(mtip_ctx* ctx, int64_t first_step, int64_t n_steps, double* main_err);
/* one shrink-wrap update (reconstruct.py:598-605, 877-885; fxs_Projections.py:245-258):
 * enforce_initial_support_b = (last main error_b > error_limit); enforced[n_batch] (may be NULL)
 * returns the decision per restart.  The fixed amplitudes of the *_non_FXS variants survive this call and
 * mtip_refresh_reciprocal_density (reconstruct.py:898-904: only an FXS method or a new sub-loop resets them); they are
 * |F'| of the pair the reference's stale `hist` ends with (901): the input pair of the most recent step. */
int mtip_shrinkwrap(mtip_ctx* ctx, double sigma, double threshold, double error_limit, uint8_t* enforced);
/* Top of a sub-loop call (reconstruct.py:852-866): the reference re-reads its local `hist` from the state and forgets
 * the fixed amplitudes of the *_non_FXS variants.  Call it before the first method of every sub-loop. */
int mtip_begin_sub_loop(mtip_ctx* ctx);
/* 'SW_center' (reconstruct.py:606-613, 886-897), after mtip_shrinkwrap.  Reproduces the reference literally: its process
 * returns (support, copy(rho), FT(rho)), which the loop unpacks as (support, ft_density, density), so the latest pair
 * becomes (reciprocal, real) = (rho, FT(rho)); the history is rebuilt from the stale `hist` (893), i.e. the pair of the
 * most recent step is dropped.  The best pair is left untouched. */
int mtip_refresh_reciprocal_density(mtip_ctx* ctx);
/* B_l = I_l I_l^+ of FT(latest rho) (reconstruct.py:757-765, 992-993), (L+1, Nq, Nq) complex */
int mtip_last_deg2_invariant(mtip_ctx* ctx, int batch, mtip_cdouble* Bl);

/* ---- single operators (parity tests; the operator registry of reconstruct.py:370,391,445,485; the
 *      averaging of f-1).  All arrays carry the leading n_batch dimension.  The caller's buffers may be
 *      host memory or memory of the context's device (unified addressing decides the copy): the
 *      averaging keeps its batch in HBM between these calls. ---------------------------------------- */
int mtip_op_sht_forward(mtip_ctx* ctx, const mtip_cdouble* grid, mtip_cdouble* coeff, int prologue);
int mtip_op_sht_inverse(mtip_ctx* ctx, const mtip_cdouble* coeff, mtip_cdouble* grid);
/* inverse transform and, on the grid it produced, the forward transform of prologue(grid) -- inverse_harmonic_transform then
 * [square_grid, misk.py:159-168, then] harmonic_transform, the way three links of a phasing step chain them
 * (reconstruct.py:518-528, 576-593): grid = iSHT(coeff), coeff_out = SHT(grid) (prologue 0) or SHT(|grid|^2) (prologue 1).
 * One kernel per shell where the angular grid allows it (csrc/k_sht_chain.hip; profile family "sht_chain"), else the two
 * transforms one after the other. */
int mtip_op_sht_inverse_forward(mtip_ctx* ctx, const mtip_cdouble* coeff, mtip_cdouble* grid, mtip_cdouble* coeff_out, int prologue);
int mtip_op_hankel(mtip_ctx* ctx, const mtip_cdouble* coeff_in, mtip_cdouble* coeff_out, int inverse);
int mtip_op_fourier_transform(mtip_ctx* ctx, const mtip_cdouble* grid_in, mtip_cdouble* grid_out, int inverse);
/* approximate_unknowns + mtip_projection (fxs_Projections.py:752-767, 832-872) on 'direct' coefficients */
int mtip_op_project_coefficients(mtip_ctx* ctx, const mtip_cdouble* Ilm, mtip_cdouble* Ilm_projected);
/* the same for coefficients of a REAL intensity, I_{l,-m} = (-1)^m conj(I_{l,m}) -- what the phasing loop always passes
 * (reconstruct.py:866-870: SHT of |F|^2).  Only the m >= 0 half is read.  With projection matrices whose imaginary part
 * is exactly zero (eigenvectors of a real B_l, fxs_invariant_tools.py:1114-1141, 1207: what the reference's cross-correlation
 * route gives; env MTIP_PROJ_REAL_TOL=t additionally drops imaginary parts below t max|V_l|, the rounding residue of its
 * `density` route) the polar factors are computed in real arithmetic; any other case takes the general path of
 * mtip_op_project_coefficients.  Same result as that function to rounding. */
int mtip_op_project_real_intensity(mtip_ctx* ctx, const mtip_cdouble* Ilm, mtip_cdouble* Ilm_projected);
/* mtip_projection with caller-supplied unknowns (fxs_Projections.py:832-849, 866-871; registry operator
 * 'mtip_projection(Ilm, unknowns)', reconstruct.py:391): U = per restart the concatenation over l = 0..L of the
 * row-major (min(2l+1, Nq), 2l+1) blocks, i.e. the layout mtip_get_unknowns reads order by order */
int mtip_op_apply_unknowns(mtip_ctx* ctx, const mtip_cdouble* Ilm, const mtip_cdouble* U, mtip_cdouble* Ilm_projected);
/* project_to_modified_intensity (fxs_Projections.py:899-909): F' = F sqrt(Re I'/|F|^2) */
int mtip_op_modulus_replacement(mtip_ctx* ctx, const mtip_cdouble* F, const mtip_cdouble* I_new, mtip_cdouble* F_new);
/* real_projection + hybrid_input_output / error_reduction + l2 error (fxs_Projections.py:110-130,
 * fxs_IO_methods.py:40-68, 97-128) using the ctx's current support of each restart */
int mtip_op_real_space_update(mtip_ctx* ctx, const mtip_cdouble* w, const mtip_cdouble* rho_prev, int method,
                              double beta, mtip_cdouble* rho_new, double* error);
int mtip_op_deg2_invariants(mtip_ctx* ctx, const mtip_cdouble* Ilm, mtip_cdouble* Bl);
/* generic y = M x used by the GPU-process boundary test (tests/test_framework_integration.py:230-400
 * of the reference: gpu_func(vects) == matrix @ vects), float64 row-major */
int mtip_op_apply_matrix(mtip_ctx* ctx, const double* matrix, const double* vects, double* out,
                         int n_rows, int n_cols, int n_vec);

/* ---- rotational alignment of reconstructions (xframe/projects/fxs/average.py:920-960 -> pysofft through
 *      externalLibraries/soft_plugin.py:64-99; conventions: oracle/alignment.py, pysofft itself is not available) ---------
 * d_table: d^l_mn(beta_b), beta_b = pi (2b+1) / 4bw, b < 2bw, bw = L + 1: (2bw, sum_l (2l+1)^2) doubles, order l at offset
 * l(4l^2-1)/3, row m = -l..l, column n = -l..l */
int mtip_set_so3_tables(mtip_ctx* ctx, int bw, const double* d_table);
/* C(alpha_j, beta_b, gamma_k) = mean over the shells r_lo <= r < r_hi of Re <ref_r, R(alpha, beta, gamma) sig_r> for every
 * restart of the batch: ref (Nq, nlm), sig (n_batch, Nq, nlm) 'direct' coefficients, C (n_batch, 2bw, 2bw, 2bw) doubles
 * indexed [alpha][beta][gamma] (soft.calc_mean_C, soft_plugin.py:82-99) */
int mtip_op_so3_correlation(mtip_ctx* ctx, const mtip_cdouble* ref, const mtip_cdouble* sig, int r_lo, int r_hi, double* C);
/* f_lm -> sum_n D^l_mn f_ln on every shell (soft.rotate_coeff, soft_plugin.py:64-79): D (n_batch, sum_l (2l+1)^2) Wigner
 * matrices in the table layout above (one rotation per restart) */
int mtip_op_rotate_coefficients(mtip_ctx* ctx, const mtip_cdouble* coeff, const mtip_cdouble* D, mtip_cdouble* out);
/* find_rotation (average.py:920-947): the correlation and, per restart, its arg-max in the order the reference reads it in --
 * arg[b] = i_beta nb^2 + i_alpha nb + i_gamma of its np.argmax over mean_C (indexed [beta, alpha, gamma], tabulated at the angles
 * whose flip alpha -> 2 pi - alpha, gamma -> 2 pi - gamma is the aligning rotation; first maximum wins), vmax[b] the maximum;
 * C (B, nb, nb, nb) as mtip_op_so3_correlation returns it, or NULL */
int mtip_op_so3_find_rotation(mtip_ctx* ctx, const mtip_cdouble* ref, const mtip_cdouble* sig, int r_lo, int r_hi, int64_t* arg,
                              double* vmax, double* C);
/* rotate (average.py:948-960) by Euler angles (alpha[b], beta of grid sample beta_index[b], gamma[b]) -- what find_rotation hands
 * out; D^l_mn is built on the device from the Wigner table (host arrays of B entries) */
int mtip_op_rotate_coefficients_grid(mtip_ctx* ctx, const mtip_cdouble* coeff, const int32_t* beta_index, const double* alpha,
                                     const double* gamma, mtip_cdouble* out);

/* ---- grid arithmetic of the averaging worker (average.py:359-627, 721-727) between the transforms: csrc/k_average.hip.  Stacks of
 * n grids (n, Nq, n_theta, n_phi) complex128; every buffer may be host memory or memory of the context's device.
 * mtip_op_grid_stats: per grid 12 doubles -- [0] sum w Re, [1..3] sum w Re {x, y, z} (the centre of mass of misk.py:295-312 is
 *   [1..3] / [0]), [4] sum w Re^2, [5] sum w (Re ref - Re)^2 (0 without ref), [6] max Re, [7] min Re, [8], [9] the sum and [10] the
 *   count of the entries numpy calls > 0 (Re > 0, or Re == 0 and Im > 0; average.py:432-435), [11] 0; w = radial_w[q] theta_w[t]
 *   (the SphericalIntegrator's weights, mathLibrary.py:1223-1237, without its 1 / volume)
 * mtip_op_grid_phase_ramp: grid b *= exp(-i sign k . c_b), k the cartesian reciprocal grid vector, c_b = centers_cartesian[b]
 *   (generate_shift_by_operator, fxs_Projections.py:1419-1444; sign -1 = opposite direction), in place
 * mtip_op_grid_combine: op 0 dst[b] = conj(a[b]); 1 dst[b] = a[b] * scalars[b]; 2 dst = sum_b a[b] (one grid); 3 dst = sum_b |a[b]|^2;
 *   4 dst[b] = (a[b] - scalars[0]) * scalars[1] (average.py:721-727); dst may be a
 * mtip_op_prtf: resolution_metrics.py:62-78 -- nd = sqrt(a1 conj(a2) / (b1 b2)), b = sqrt(Re I), with the reference's rules for
 *   vanishing b; per shell the mean (complex, (Nq)) and the standard deviation ((Nq)) over the sphere */
int mtip_op_grid_stats(mtip_ctx* ctx, const mtip_cdouble* grids, int n, const mtip_cdouble* ref, const double* radial_w,
                       const double* theta_w, double* out);
int mtip_op_grid_phase_ramp(mtip_ctx* ctx, mtip_cdouble* grids, int n, const double* centers_cartesian, double sign);
int mtip_op_grid_combine(mtip_ctx* ctx, int op, mtip_cdouble* dst, const mtip_cdouble* a, int n, const mtip_cdouble* scalars);
int mtip_op_prtf(mtip_ctx* ctx, const mtip_cdouble* a1, const mtip_cdouble* a2, const mtip_cdouble* I1, const mtip_cdouble* I2,
                 mtip_cdouble* mean, double* std_dev);

/* ---- upstream step `extract`: B_l -> V_l (fxs_invariant_tools.py:1079-1131, 1171-1207) -------------------------------
 * eigen-decomposition of n_mat Hermitian n x n matrices A (row-major, only their Hermitian part matters): eigvals (n_mat, n)
 * unsorted, eigvecs (n_mat, n, n) with eigenvector i of matrix k in eigvecs[k][i][:] (one eigenvector per row).  Sorting,
 * the cut to min(2l+1, Nq) pairs, clipping of negative eigenvalues and V_l = eigvecs sqrt(eigvals) are host bookkeeping. */
int mtip_op_hermitian_eig(mtip_ctx* ctx, int n, int n_mat, const mtip_cdouble* A, double* eigvals, mtip_cdouble* eigvecs);
/* the same for REAL symmetric matrices up to 128 x 128 (B_l of a real intensity has no imaginary part; the rules of
 * fxs_invariant_tools.py:1114-1131 stay on the host): LDS-resident solver, eigenvalues accurate to eps |A|_F as LAPACK's */
int mtip_op_symmetric_eig(mtip_ctx* ctx, int n, int n_mat, const double* A, double* eigvals, double* eigvecs);

/* ---- the 2-D (polar) variant, operator level --------------------------------------
 * Grids (n_batch, Nq, n_phi) complex128 with n_phi = 2 M + 1 (harmonic_transforms.py:44-47); harmonic coefficients in numpy's
 * FFT order (orders 0..M, -M..-1); coefficients of the real transform (n_batch, Nq, M + 1).  Buffers: host or device memory. */
typedef struct mtip2d_ctx mtip2d_ctx;
mtip2d_ctx* mtip2d_create(int n_radial, int n_phi, int n_batch, int device);
void mtip2d_destroy(mtip2d_ctx* ctx);
const char* mtip2d_last_error(const mtip2d_ctx* ctx);
/* forward / inverse weights (Nq summed, Nq new, n_phi orders) as assemble_weights_mid returns them (hankel_transforms.py:300-362),
 * unused_orders (n_phi): 1 = zeroed by the transform pair (generate_polar_ht, 613-628) */
int mtip2d_set_hankel_weights(mtip2d_ctx* ctx, const mtip_cdouble* forward, const mtip_cdouble* inverse, const uint8_t* unused_orders);
/* the 2-D reciprocal projection (fxs_Projections.py:723-745, 803-826, 855-863): order_ids (n_used, ascending, 0 included),
 * projection vectors (n_used, Nq), their radial masks (n_used, Nq), reciprocal radial points (Nq), number of particles */
int mtip2d_set_projection(mtip2d_ctx* ctx, int n_used, const int32_t* order_ids, const mtip_cdouble* projection_vectors,
                          const uint8_t* radial_mask, const double* radial_points, double n_particles);
/* projections.reciprocal.SO_freedom for dimensions == 2 (fxs_Projections.py:744-750, 933-971): the unknown at `position` among the
 * used orders is set to 1 in every projection; -1 switches it off (mtip2d_set_projection resets it) */
int mtip2d_set_so_freedom(mtip2d_ctx* ctx, int position);
/* circularHarmonicTransform_complex_forward / _inverse (mathLibrary.py:469-483) */
int mtip2d_op_harmonic(mtip2d_ctx* ctx, const mtip_cdouble* in, mtip_cdouble* out, int inverse);
/* circularHarmonicTransform_real_forward (485-491: of the real part of a complex grid) / _real_inverse (493-496: real grid out) */
int mtip2d_op_real_harmonic_forward(mtip2d_ctx* ctx, const mtip_cdouble* grid, mtip_cdouble* coeff);
int mtip2d_op_real_harmonic_inverse(mtip2d_ctx* ctx, const mtip_cdouble* coeff, double* grid);
/* generate_polar_ht (hankel_transforms.py:629-640) and generate_ft for dimensions = 2 (fourier_transforms.py:57-88) */
int mtip2d_op_hankel(mtip2d_ctx* ctx, const mtip_cdouble* in, mtip_cdouble* out, int inverse);
int mtip2d_op_fourier_transform(mtip2d_ctx* ctx, const mtip_cdouble* in, mtip_cdouble* out, int inverse);
/* approximate_unknowns + mtip_projection + the number-of-particles rule on coefficients of the real transform; unknowns
 * (n_batch, n_used) or NULL */
int mtip2d_op_project(mtip2d_ctx* ctx, const mtip_cdouble* I, mtip_cdouble* I_projected, mtip_cdouble* unknowns);
/* ---- the operators of the 2-D phasing loop (the schedule, ramps, support bookkeeping and best tracking are host logic) ---- */
int mtip2d_set_real_constraints(mtip2d_ctx* ctx, uint32_t flags, double value_lo, double value_hi, double imag_threshold, uint32_t hio_flags);
/* (Nq, n_phi) weights of l2_projection_diff: PolarIntegrator weights (mathLibrary.py:1242-1265) times the metric's mask */
int mtip2d_set_error_weights(mtip2d_ctx* ctx, const double* weights);
/* one HIO (0) / ER (1) step, optionally with the ft_stab add-back (sketches reconstruct.py:518-528, 576-593): rho, support
 * (n_batch, Nq, n_phi) in; F' and the new density, the error per restart and the unknowns (n_batch, n_used; or NULL) out */
int mtip2d_op_step(mtip2d_ctx* ctx, int method, int ft_stab, double beta, const mtip_cdouble* rho, const uint8_t* support,
                   mtip_cdouble* F_new, mtip_cdouble* rho_new, double* err, mtip_cdouble* unknowns);
/* the same with the loop's sub-variants: method 2 / 3 = HIO_non_FXS / ER_non_FXS (sketch MTIP_start_non_FXS, reconstruct.py:530-535,
 * 899-904: F' = F sqrt(fixed / |F|^2) with fixed_intensity (n_batch, Nq, n_phi), no harmonic transform, no unknowns), and the inputs
 * of the reciprocal error metrics (fxs_IO_methods.py:301-310, 370-400) when asked for: F_out = FT(rho) (n_batch, Nq, n_phi), I_out =
 * the harmonic coefficients of |F|^2 (n_batch, Nq, M + 1; FXS methods only); any of fixed_intensity / F_out / I_out / unknowns
 * may be NULL */
int mtip2d_op_step_ex(mtip2d_ctx* ctx, int method, int ft_stab, double beta, const mtip_cdouble* rho, const uint8_t* support,
                      const double* fixed_intensity, mtip_cdouble* F_new, mtip_cdouble* rho_new, double* err, mtip_cdouble* unknowns,
                      mtip_cdouble* F_out, mtip_cdouble* I_out);
/* the SW sketch (reconstruct.py:598-605, fxs_Projections.py:245-258, 294-298): support mask (n_batch, Nq, n_phi) from rho */
int mtip2d_op_shrinkwrap(mtip2d_ctx* ctx, const mtip_cdouble* rho, double sigma, double threshold, uint8_t* mask);

/* ---- timing ----------------------------------------------------------------------------------- */
/* average duration (ms) and launch count of kernel family `name` ("sht_fwd", "sht_inv", "hankel",
 * "proj", "real_update", ...) measured with hipEvents on the ctx stream since the last reset;
 * enable with mtip_profile(ctx, 1). */
int mtip_profile(mtip_ctx* ctx, int enable);
int mtip_profile_get(mtip_ctx* ctx, const char* name, double* total_ms, int64_t* launches);
int mtip_profile_reset(mtip_ctx* ctx);
/* diagnostic of the last polar-factor solve, (n_batch, L+1) int32: bits 0-7 Jacobi sweeps used, bits 8+ the
 * number of columns of X_l that were still non-zero (not deflated) in the final sweep */
int mtip_debug_jacobi_sweeps(mtip_ctx* ctx, int32_t* out);
/* diagnostic: phase stamps of the chained inverse -> forward SHT kernel (csrc/k_sht_chain.hip), (3 kinds: store / modulus /
 * real-space epilogue, n_batch * n_radial shells, 18) int64 s_memtime ticks of the last launch of each kind: [0] start, [1]
 * tables staged, [2] / [3] Legendre synthesis done (wave 0 / all), then per FFT pass p (6 p + 4 ...): step 1 done, barrier,
 * epilogue + forward phase 1 done, barrier, forward phase 2 done, barrier; [16] Legendre sums done, [17] end.  The first call
 * switches the stamps on (out may be null). */
int mtip_debug_chain_timing(mtip_ctx* ctx, int64_t* out);
/* diagnostic: workgroups per restart of the real projection kernel (k_rproj: the host-packed slots of orders), 0 before the
 * first projection or when the general complex kernels are in use; negative = error code */
int mtip_debug_projection_slots(mtip_ctx* ctx);
/* diagnostic: in-kernel timers of the real projection kernel (k_rproj), (n_batch, L+1, MTIP_POLAR_TIMING_SLOTS = 40) int64
 * s_memtime ticks of the last projection per (restart, order): [0..4] phases X~ product, warm start, Jacobi, U, apply;
 * [5] rounds; [6], [7] start / end; [8] hardware id; [10..17] busy and [18..25] LDS drain + barrier per wave, [26..32] the
 * segments of a round of wave 0 and [33..39] of wave 5 (summed over the rounds; largest order of L = 32 only).  The first call
 * (out may be NULL) switches the timers on. */
#define MTIP_POLAR_TIMING_SLOTS 40
int mtip_debug_polar_timing(mtip_ctx* ctx, int64_t* out);
/* diagnostic: enqueue a one-workgroup kernel that spins for `microseconds` on the context's stream (asynchronous).  Used to
 * check that the streams of several engines of one process really execute side by side (HIP maps streams onto a limited
 * pool of hardware queues, GPU_MAX_HW_QUEUES; two streams on one queue serialise). */
int mtip_debug_spin(mtip_ctx* ctx, double microseconds);
/* diagnostic: build and verify the resident-column pairing schedule of the polar-factor kernel for every column
 * count 2..k_max <= 127 (every pair exactly once per sweep, no column twice in a round); MTIP_OK or MTIP_EINVAL */
int mtip_debug_check_jacobi_schedule(mtip_ctx* ctx, int k_max);

#ifdef __cplusplus
}
#endif
#endif /* MTIP_HIP_H */
