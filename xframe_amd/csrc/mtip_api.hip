// C ABI of libmtip_hip.so (include/mtip_hip.h): context, one-off uploads, the device-resident phasing loop
// (xframe/projects/fxs/reconstruct.py:854-951 -- one step = sketches 518-528 + 576-593) and the
// single-operator entry points used by the parity tests.
#include "mtip_internal.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

static thread_local std::string g_create_error;     // mtip_create may run on several host threads (one engine each)

#define CTX_CHECK(c)                       \
    do {                                   \
        if ((c) == nullptr) return MTIP_EINVAL; \
    } while (0)
#define FAIL(c, code, msg)      \
    do {                        \
        (c)->err = (msg);       \
        return (code);          \
    } while (0)

static int post_launch(mtip_ctx* c, const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        c->err = std::string(what) + ": " + hipGetErrorString(e);
        return MTIP_EHIP;
    }
    return MTIP_OK;
}

template <typename T>
static int dev_alloc(mtip_ctx* c, T** p, size_t n) {
    hipError_t e = hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(T));
    if (e != hipSuccess) {
        c->err = std::string("hipMalloc: ") + hipGetErrorString(e);
        *p = nullptr;
        return MTIP_ENOMEM;
    }
    return MTIP_OK;
}

extern "C" {

int mtip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* mtip_last_error(const mtip_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

void mtip_destroy(mtip_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_invariant_metrics(c);
    void* ptrs[] = {c->d_cost, c->d_gw, c->d_P, c->d_r, c->d_q, c->d_poff, c->d_PT, c->d_AB, c->d_lmtab, c->d_twN, c->d_tw, c->d_W, c->d_htiles32, c->d_kl, c->d_used, c->d_active, c->d_sweeps, c->d_jsched, c->d_jsched_off, c->d_jsched_rounds, c->d_jorder, c->d_pg_tiles[0], c->d_pg_tiles[1], c->d_pg_tiles[2], c->d_pg_tiles[3], c->d_pg_tiles[4], c->d_pg_tiles[5], c->d_voff,
                    c->d_xoff, c->d_uoff, c->d_V, c->d_rmask, c->d_Bref, c->d_Bnorm, c->d_deg2_part, c->d_S0, c->d_sup, c->d_err_wr,
                    c->d_err_wt, c->d_rho, c->d_Fp, c->d_slot, c->d_best_err, c->d_last_err, c->d_op_err, c->d_gq, c->d_polar_dbg, c->d_so3_d, c->d_so3_tw, c->d_so3_T, c->d_so3_S, c->d_so3_P, c->d_so3_D, c->d_so3_C, c->d_err_hist, c->d_main_hist,
                    c->d_deg2_hist, c->d_F, c->d_T1, c->d_T2, c->d_fixed, c->d_g, c->d_c[0], c->d_c[1], c->d_c[2],
                    c->d_c[3], c->d_c[4], c->d_c[5], c->d_X, c->d_Vr, c->d_U, c->d_partial, c->d_minmax, c->d_Bl,
                    c->d_rp_DV, c->d_rp_Vt, c->d_rp_slots, c->d_c0n, c->d_mk, c->d_chain_dbg, c->d_PTc, c->d_lmc, c->d_ftmask};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (hipEvent_t e : c->prof_events) (void)hipEventDestroy(e);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->turn_ev) (void)hipEventDestroy(c->turn_ev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

static bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

mtip_ctx* mtip_create(const mtip_cfg* cfg, int device) {
    if (!cfg) {
        g_create_error = "cfg is NULL";
        return nullptr;
    }
    const int ndev = mtip_device_count();
    if (ndev <= 0) {
        g_create_error = "no HIP device visible (libmtip_hip needs an MI355X; there is no CPU fallback)";
        return nullptr;
    }
    if (device < 0 || device >= ndev) {
        g_create_error = "device index out of range";
        return nullptr;
    }
    if (cfg->n_radial < 2 || cfg->l_max < 0 || cfg->l_max > 63 || cfg->n_batch < 1 || !is_pow2(cfg->n_phi) ||
        cfg->n_phi < 4 || cfg->n_phi > 512 || cfg->n_phi <= 2 * cfg->l_max || cfg->n_theta <= cfg->l_max) {
        g_create_error = "invalid cfg: need Nq>=2, 0<=L<=63, n_batch>=1, n_phi a power of two in [4,512] and > 2L, n_theta > L";
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        g_create_error = "hipSetDevice failed";
        return nullptr;
    }
    mtip_ctx* c = new mtip_ctx();
    c->cfg = *cfg;
    c->device = device;
    c->N = cfg->n_radial;
    c->L = cfg->l_max;
    c->nt = cfg->n_theta;
    c->np = cfg->n_phi;
    c->B = cfg->n_batch;
    c->nlm = (c->L + 1) * (c->L + 1);
    c->nm = 2 * c->L + 1;
    c->Np = cfg->hankel_trapz ? c->N - 1 : c->N;
    if (const char* e = std::getenv("MTIP_POLAR_ABS_TOL")) c->polar_abs_tol = std::atof(e);
    c->G = (size_t)c->N * c->nt * c->np;
    c->C = (size_t)c->N * c->nlm;
    const int L = c->L, N = c->N, B = c->B;
    int rc = MTIP_OK;
    auto A = [&](int r) { if (rc == MTIP_OK) rc = r; };
    // not synchronised with the null stream (see mtip_copy in mtip_internal.h)
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) A(MTIP_EHIP);
    {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) c->n_cu = ncu;
    }
    (void)hipEventCreate(&c->ev0);
    (void)hipEventCreate(&c->ev1);
    A(dev_alloc(c, &c->d_cost, c->nt));
    A(dev_alloc(c, &c->d_gw, c->nt));
    A(dev_alloc(c, &c->d_P, (size_t)(L + 1) * (L + 2) / 2 * c->nt));
    A(dev_alloc(c, &c->d_poff, L + 2));
    c->npairs = (L + 1) * (L + 2) / 2;
    A(dev_alloc(c, &c->d_PT, (size_t)(c->nt / 2 + 1) * c->npairs));
    A(dev_alloc(c, &c->d_AB, (size_t)c->npairs));
    A(dev_alloc(c, &c->d_lmtab, c->npairs));
    A(dev_alloc(c, &c->d_PTc, (size_t)(c->nt / 2 + 1) * 768));
    A(dev_alloc(c, &c->d_lmc, 768));
    if (const char* e = std::getenv("MTIP_PROJ_FUSE")) c->proj_fuse = std::atoi(e) != 0;
    if (const char* e = std::getenv("MTIP_DEG2_SIMPLE")) c->deg2_simple = std::atoi(e) != 0;
    if (const char* e = std::getenv("MTIP_SHT_FWD_PAIR")) c->sht_fwd_pair = std::atoi(e) != 0;
    if (const char* e = std::getenv("MTIP_FUSE_REAL")) c->fuse_real_update = std::atoi(e) != 0;
    if (const char* e = std::getenv("MTIP_SHT_WIDE")) c->sht_wide = std::atoi(e) != 0;
    if (const char* e = std::getenv("MTIP_SHT_CHAIN")) c->sht_chain = std::atoi(e) != 0;
    if (const char* e = std::getenv("MTIP_SHT_CHAIN_LC")) c->sht_chain_lc = std::atoi(e) != 0;
    if (const char* e = std::getenv("MTIP_JAC_RESIDENT")) c->jac_resident = std::atoi(e) != 0;
    if (const char* e = std::getenv("MTIP_JAC_TG")) c->jac_tg = std::atoi(e) == 8 ? 8 : 16;
    if (const char* e = std::getenv("MTIP_HANKEL_CT")) {
        const int v = std::atoi(e);
        c->htile_force = (v == 1 || v == 2 || v == 3 || v == 5) ? v : 0;
    }
    if (rc == MTIP_OK) rc = build_hankel_tiles(c);
    A(dev_alloc(c, &c->d_twN, c->np));
    if (const char* e = std::getenv("MTIP_SHT_MODE")) c->sht_mode = std::atoi(e);
    c->sht_unfused = c->sht_mode < 1;
    A(dev_alloc(c, &c->d_tw, c->np / 2));
    A(dev_alloc(c, &c->d_r, N));
    A(dev_alloc(c, &c->d_q, N));
    A(dev_alloc(c, &c->d_W, (size_t)(L + 1) * c->Np * N));
    // projection layout
    c->kl.assign(L + 1, 0);
    c->used.assign(L + 1, 0);
    c->active.assign(L + 1, 0);
    c->voff.assign(L + 2, 0);
    c->xoff.assign(L + 2, 0);
    c->uoff.assign(L + 2, 0);
    c->have_V.assign(L + 1, 0);
    c->v_real.assign(L + 1, 1);
    if (const char* e = std::getenv("MTIP_PROJ_REAL")) c->proj_real = std::atoi(e) != 0;
    if (const char* e = std::getenv("MTIP_RP_CORR")) c->rp_corr = std::atoi(e) != 0;
    // closing-step thresholds: A/B switches, held to the range the parity cases cover (test_projection_real_switches) -- a stray
    // shell variable must not loosen the 1e-10 operator
    if (const char* e = std::getenv("MTIP_RP_EARLY")) c->rp_early = std::min(std::max(std::atof(e), 1e-3), 0.2);
    if (const char* e = std::getenv("MTIP_RP_CORR2_MAX")) c->rp_corr2_max = std::min(std::max(std::atof(e), 1e-6), 5e-4);
    for (int l = 0; l <= L; ++l) {
        const int n = 2 * l + 1, k = std::min(n, N);
        c->kl[l] = k;                               // default; mtip_set_projection_matrix may give a smaller k_l
        c->voff[l + 1] = c->voff[l] + N * k;
        c->xoff[l + 1] = c->xoff[l] + k * n;
        c->uoff[l + 1] = c->uoff[l] + k * k;
    }
    c->vtot = c->voff[L + 1];
    c->xtot = c->xoff[L + 1];
    c->utot = c->uoff[L + 1];
    A(dev_alloc(c, &c->d_kl, L + 1));
    A(dev_alloc(c, &c->d_used, L + 1));
    A(dev_alloc(c, &c->d_active, L + 1));
    A(dev_alloc(c, &c->d_sweeps, (size_t)B * (L + 1)));
    A(dev_alloc(c, &c->d_voff, L + 2));
    A(dev_alloc(c, &c->d_xoff, L + 2));
    A(dev_alloc(c, &c->d_uoff, L + 2));
    A(dev_alloc(c, &c->d_V, (size_t)c->vtot));
    c->h_V.assign((size_t)c->vtot, make_double2(0.0, 0.0));
    A(dev_alloc(c, &c->d_rmask, (size_t)(L + 1) * N));
    A(dev_alloc(c, &c->d_Bref, (size_t)(L + 1) * N * N));
    A(dev_alloc(c, &c->d_Bnorm, L + 1));
    A(dev_alloc(c, &c->d_deg2_part, (size_t)B * (L + 1) * div_up(N, 16) * div_up(N, 16)));
    A(dev_alloc(c, &c->d_S0, c->G));
    A(dev_alloc(c, &c->d_sup, (size_t)3 * B * c->G));
    A(dev_alloc(c, &c->d_mk, (size_t)3 * B * c->G / 4));
    A(dev_alloc(c, &c->d_err_wr, N));
    A(dev_alloc(c, &c->d_err_wt, c->nt));
    A(dev_alloc(c, &c->d_rho, (size_t)3 * B * c->G));
    A(dev_alloc(c, &c->d_Fp, (size_t)3 * B * c->G));
    A(dev_alloc(c, &c->d_slot, (size_t)B * SL_N));
    A(dev_alloc(c, &c->d_best_err, B));
    A(dev_alloc(c, &c->d_last_err, B));
    A(dev_alloc(c, &c->d_op_err, B));
    A(dev_alloc(c, &c->d_gq, N));
    c->err_cap = 4096;
    A(dev_alloc(c, &c->d_err_hist, (size_t)c->err_cap * B));
    A(dev_alloc(c, &c->d_main_hist, (size_t)c->err_cap * B));
    A(dev_alloc(c, &c->d_deg2_hist, (size_t)c->err_cap * B * (L + 1)));
    A(dev_alloc(c, &c->d_F, (size_t)B * c->G));
    A(dev_alloc(c, &c->d_T1, (size_t)B * c->G));
    A(dev_alloc(c, &c->d_T2, (size_t)B * c->G));
    A(dev_alloc(c, &c->d_fixed, (size_t)B * c->G));
    A(dev_alloc(c, &c->d_g, (size_t)B * N * c->nt * c->nm));
    for (int i = 0; i < 6; ++i) A(dev_alloc(c, &c->d_c[i], (size_t)B * c->C));
    A(dev_alloc(c, &c->d_c0n, (size_t)B * c->C));
    A(dev_alloc(c, &c->d_X, (size_t)B * c->xtot));
    A(dev_alloc(c, &c->d_Vr, (size_t)B * c->utot));
    A(dev_alloc(c, &c->d_U, (size_t)B * c->xtot));
    c->n_partial_blocks = div_up((long long)c->G, 256 * 4);
    A(dev_alloc(c, &c->d_partial, (size_t)B * std::max(c->n_partial_blocks, 2 * N) * 2));
    A(dev_alloc(c, &c->d_minmax, (size_t)B * c->n_partial_blocks * 2));
    if (rc != MTIP_OK) {
        g_create_error = c->err.empty() ? "allocation failed" : c->err;
        mtip_destroy(c);
        return nullptr;
    }
    (void)mtip_copy(c, c->d_kl, c->kl.data(), (L + 1) * sizeof(int), hipMemcpyHostToDevice);
    (void)mtip_copy(c, c->d_voff, c->voff.data(), (L + 2) * sizeof(int), hipMemcpyHostToDevice);
    (void)mtip_copy(c, c->d_xoff, c->xoff.data(), (L + 2) * sizeof(int), hipMemcpyHostToDevice);
    (void)mtip_copy(c, c->d_uoff, c->uoff.data(), (L + 2) * sizeof(int), hipMemcpyHostToDevice);
    (void)hipMemsetAsync(c->d_used, 0, (L + 1) * sizeof(int), c->stream);
    (void)hipMemsetAsync(c->d_active, 0, (L + 1) * sizeof(int), c->stream);
    (void)hipMemsetAsync(c->d_sweeps, 0, (size_t)B * (L + 1) * sizeof(int), c->stream);
    (void)hipMemsetAsync(c->d_U, 0, (size_t)B * c->xtot * sizeof(double2), c->stream);
    (void)hipMemsetAsync(c->d_X, 0, (size_t)B * c->xtot * sizeof(double2), c->stream);
    (void)hipMemsetAsync(c->d_Vr, 0, (size_t)B * c->utot * sizeof(double2), c->stream);
    (void)hipMemsetAsync(c->d_V, 0, (size_t)c->vtot * sizeof(double2), c->stream);
    (void)hipMemsetAsync(c->d_rmask, 0, (size_t)(L + 1) * N, c->stream);
    (void)hipMemsetAsync(c->d_sup, 1, (size_t)3 * B * c->G, c->stream);
    (void)hipMemsetAsync(c->d_S0, 1, c->G, c->stream);
    (void)hipMemsetAsync(c->d_mk, 0xff, (size_t)3 * B * c->G / 4 * sizeof(uint16_t), c->stream);
    // default slots
    std::vector<int> slots((size_t)B * SL_N, 0);
    for (int b = 0; b < B; ++b) {
        slots[b * SL_N + SL_CUR] = 0;
        slots[b * SL_N + SL_OUT] = 1;
        slots[b * SL_N + SL_BEST] = 0;
        slots[b * SL_N + SL_HIST] = 0;
        slots[b * SL_N + SL_ENFORCE] = 1;
    }
    (void)mtip_copy(c, c->d_slot, slots.data(), slots.size() * sizeof(int), hipMemcpyHostToDevice);
    std::vector<double> inf(B, HUGE_VAL);
    (void)mtip_copy(c, c->d_best_err, inf.data(), B * sizeof(double), hipMemcpyHostToDevice);
    (void)mtip_copy(c, c->d_last_err, inf.data(), B * sizeof(double), hipMemcpyHostToDevice);
    return c;
}

int mtip_get_cfg(const mtip_ctx* c, mtip_cfg* out) {
    CTX_CHECK(c);
    if (!out) return MTIP_EINVAL;
    *out = c->cfg;
    return MTIP_OK;
}

int mtip_synchronize(mtip_ctx* c) {
    CTX_CHECK(c);
    MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return post_launch(c, "synchronize");
}

// ---- one-off setup ------------------------------------------------------------------------------------
int mtip_set_angular_grid(mtip_ctx* c, const double* cos_theta, const double* gauss_weights) {
    CTX_CHECK(c);
    if (!cos_theta || !gauss_weights) FAIL(c, MTIP_EINVAL, "null angular grid");
    (void)hipSetDevice(c->device);
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_cost, cos_theta, c->nt * sizeof(double), hipMemcpyHostToDevice));
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_gw, gauss_weights, c->nt * sizeof(double), hipMemcpyHostToDevice));
    build_legendre_tables(c, cos_theta);
    c->have_angular = true;
    return MTIP_OK;
}

int mtip_set_radial_grid(mtip_ctx* c, const double* r, const double* q) {
    CTX_CHECK(c);
    if (!r || !q) FAIL(c, MTIP_EINVAL, "null radial grid");
    (void)hipSetDevice(c->device);
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_r, r, c->N * sizeof(double), hipMemcpyHostToDevice));
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_q, q, c->N * sizeof(double), hipMemcpyHostToDevice));
    c->have_radial = true;
    return MTIP_OK;
}

int mtip_set_hankel_weights(mtip_ctx* c, const double* w_raw, double fwd_scale, double inv_scale) {
    CTX_CHECK(c);
    if (!w_raw) FAIL(c, MTIP_EINVAL, "null weights");
    (void)hipSetDevice(c->device);
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_W, w_raw, (size_t)(c->L + 1) * c->Np * c->N * sizeof(double), hipMemcpyHostToDevice));
    c->fwd_scale = fwd_scale;
    c->inv_scale = inv_scale;
    c->have_weights = true;
    return MTIP_OK;
}

int mtip_set_projection_matrix(mtip_ctx* c, int l, const mtip_cdouble* V, int k_l, const uint8_t* radial_mask, int used) {
    CTX_CHECK(c);
    if (l < 0 || l > c->L) FAIL(c, MTIP_EINVAL, "order out of range");
    const int kmax = std::min(2 * l + 1, c->N);
    if (k_l < 1 || k_l > kmax) FAIL(c, MTIP_EINVAL, "k_l must be in [1, min(2l+1, Nq)]");
    if (used && (!V || !radial_mask)) FAIL(c, MTIP_EINVAL, "null projection matrix");
    (void)hipSetDevice(c->device);
    if (c->d_jorder != nullptr) {                        // ... and so does the order list of the polar-factor kernel
        (void)hipStreamSynchronize(c->stream);
        (void)hipFree(c->d_jorder);
        c->d_jorder = nullptr;
    }
    if (c->d_pg_tiles[0] != nullptr) {                   // the tile lists of the projection GEMMs depend on k_l / used
        (void)hipStreamSynchronize(c->stream);
        for (int op = 0; op < 6; ++op) {
            (void)hipFree(c->d_pg_tiles[op]);
            c->d_pg_tiles[op] = nullptr;
        }
    }
    // the storage slot has room for kmax columns; a narrower matrix is zero padded (zero columns of V
    // do not contribute to V_l U_l)
    std::vector<double2> tmp((size_t)c->N * kmax, make_double2(0.0, 0.0));
    if (V)
        for (int q = 0; q < c->N; ++q)
            for (int i = 0; i < k_l; ++i) tmp[(size_t)q * kmax + i] = make_double2(V[(size_t)q * k_l + i].re, V[(size_t)q * k_l + i].im);
    // MTIP_PROJ_REAL_TOL = t > 0 (opt-in, default 0): imaginary parts of V_l below t * max|V_l| are rounding residue of an
    // eigen-decomposition of a complex-typed but real-valued B_l (the reference's `density` route of extract) and are dropped,
    // so that such matrices take the real-arithmetic projection too.  With the default only Im V_l == 0 exactly does.
    double real_tol = 0.0;
    if (const char* e = std::getenv("MTIP_PROJ_REAL_TOL")) real_tol = std::atof(e);
    if (real_tol > 0.0) {
        double vmax = 0.0, imax = 0.0;
        for (const double2& v : tmp) {
            vmax = std::max(vmax, std::max(std::fabs(v.x), std::fabs(v.y)));
            imax = std::max(imax, std::fabs(v.y));
        }
        if (imax <= real_tol * vmax)
            for (double2& v : tmp) v.y = 0.0;
    }
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_V + c->voff[l], tmp.data(), tmp.size() * sizeof(double2), hipMemcpyHostToDevice));
    std::copy(tmp.begin(), tmp.end(), c->h_V.begin() + c->voff[l]);
    free_rproj_tables(c);                                // tables and slot lists of the real projection depend on V_l / used
    bool is_real = true;
    for (const double2& v : tmp) is_real = is_real && v.y == 0.0;
    c->v_real[l] = is_real ? 1 : 0;
    if (radial_mask) MTIP_HIP_CHECK(c, mtip_copy(c, c->d_rmask + (size_t)l * c->N, radial_mask, c->N, hipMemcpyHostToDevice));
    c->used[l] = used ? 1 : 0;
    bool nonzero = false;
    for (const double2& v : tmp) nonzero = nonzero || v.x != 0.0 || v.y != 0.0;
    c->active[l] = (used && nonzero) ? 1 : 0;          // V_l == 0 (odd_orders_to_0): U_l stays 0, nothing to solve
    c->vr_valid = false;
    c->vr_kind = 0;
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_used, c->used.data(), (c->L + 1) * sizeof(int), hipMemcpyHostToDevice));
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_active, c->active.data(), (c->L + 1) * sizeof(int), hipMemcpyHostToDevice));
    MTIP_HIP_CHECK(c, hipMemsetAsync(c->d_U, 0, (size_t)c->B * c->xtot * sizeof(double2), c->stream));
    c->have_V[l] = 1;
    c->bref_dirty = true;
    return MTIP_OK;
}

int mtip_set_number_of_particles(mtip_ctx* c, double n) {
    CTX_CHECK(c);
    if (!(n > 0)) FAIL(c, MTIP_EINVAL, "number of particles must be > 0");
    c->n_particles = n;
    return MTIP_OK;
}

int mtip_set_so_freedom(mtip_ctx* c, int order) {
    CTX_CHECK(c);
    if (order > c->L || (order >= 0 && (order < 2 || std::min(2 * order + 1, c->N) < 5)))
        FAIL(c, MTIP_EINVAL, "SO_freedom: order must be -1 (off) or an order >= 2 with at least 5 unknown rows");
    c->so_order = order < 0 ? -1 : order;
    return MTIP_OK;
}

// reference B_l = V_l V_l^+ masked (fxs_Projections.py:631-637, fxs_IO_methods.py:408-425), host side one-off
static int build_bref(mtip_ctx* c) {
    const int N = c->N, L = c->L;
    std::vector<double2> V((size_t)c->vtot);
    std::vector<uint8_t> rm((size_t)(L + 1) * N);
    MTIP_HIP_CHECK(c, mtip_copy(c, V.data(), c->d_V, V.size() * sizeof(double2), hipMemcpyDeviceToHost));
    MTIP_HIP_CHECK(c, mtip_copy(c, rm.data(), c->d_rmask, rm.size(), hipMemcpyDeviceToHost));
    std::vector<double2> Bref((size_t)(L + 1) * N * N, make_double2(0.0, 0.0));
    std::vector<double> norm(L + 1, 0.0);
    for (int l = 0; l <= L; ++l) {
        const int k = c->kl[l];
        const double2* Vl = V.data() + c->voff[l];
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) {
                if (!(rm[(size_t)l * N + i] && rm[(size_t)l * N + j])) continue;
                double re = 0, im = 0;
                for (int cidx = 0; cidx < k; ++cidx) {
                    const double2 a = Vl[(size_t)i * k + cidx], b = Vl[(size_t)j * k + cidx];
                    re += a.x * b.x + a.y * b.y;
                    im += a.y * b.x - a.x * b.y;
                }
                Bref[((size_t)l * N + i) * N + j] = make_double2(re, im);
                norm[l] += re * re + im * im;
            }
    }
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_Bref, Bref.data(), Bref.size() * sizeof(double2), hipMemcpyHostToDevice));
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_Bnorm, norm.data(), norm.size() * sizeof(double), hipMemcpyHostToDevice));
    c->bref_dirty = false;
    return MTIP_OK;
}

int mtip_set_deg2_metric(mtip_ctx* c, int enable) {
    CTX_CHECK(c);
    c->deg2_enable = enable ? 1 : 0;
    return MTIP_OK;
}

int mtip_set_main_error(mtip_ctx* c, int use_reciprocal_deg2, int type) {
    CTX_CHECK(c);
    if (type < 0 || type > 3) FAIL(c, MTIP_EINVAL, "main error type: 0 mean, 1 min, 2 max, 3 prod");
    c->main_mode = use_reciprocal_deg2 ? 1 : 0;
    c->main_type = type;
    return MTIP_OK;
}

int mtip_set_real_constraints(mtip_ctx* c, uint32_t flags, double lo, double hi, double imag_thr, uint32_t hio_flags) {
    CTX_CHECK(c);
    c->rp.flags = flags;
    c->rp.hio_flags = hio_flags;
    c->rp.lo = lo;
    c->rp.hi = hi;
    c->rp.imag_thr = imag_thr;
    return MTIP_OK;
}

int mtip_set_initial_support(mtip_ctx* c, const uint8_t* support) {
    CTX_CHECK(c);
    if (!support) FAIL(c, MTIP_EINVAL, "null support");
    (void)hipSetDevice(c->device);
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_S0, support, c->G, hipMemcpyHostToDevice));
    std::vector<int> slots((size_t)c->B * SL_N);
    MTIP_HIP_CHECK(c, mtip_copy(c, slots.data(), c->d_slot, slots.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (int b = 0; b < c->B; ++b) {
        slots[b * SL_N + SL_SUP] = 0;
        slots[b * SL_N + SL_SUP_BEST] = 0;
        slots[b * SL_N + SL_ENFORCE] = 1;
        MTIP_HIP_CHECK(c, mtip_copy(c, c->d_sup + (size_t)b * c->G, support, c->G, hipMemcpyHostToDevice));
    }
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_slot, slots.data(), slots.size() * sizeof(int), hipMemcpyHostToDevice));
    launch_pack_masks(c);
    c->have_support = true;
    return post_launch(c, "mtip_set_initial_support");
}

int mtip_set_error_weights(mtip_ctx* c, const double* radial_w, const double* theta_w, int use_mask) {
    CTX_CHECK(c);
    if (!radial_w || !theta_w) FAIL(c, MTIP_EINVAL, "null error weights");
    (void)hipSetDevice(c->device);
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_err_wr, radial_w, c->N * sizeof(double), hipMemcpyHostToDevice));
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_err_wt, theta_w, c->nt * sizeof(double), hipMemcpyHostToDevice));
    c->err_use_mask = use_mask ? 1 : 0;
    c->have_errw = true;
    return MTIP_OK;
}

// ---- pipelines ------------------------------------------------------------------------------------------
static int require_transforms(mtip_ctx* c) {
    if (!c->have_angular) FAIL(c, MTIP_ESTATE, "mtip_set_angular_grid has not been called");
    if (!c->have_weights) FAIL(c, MTIP_ESTATE, "mtip_set_hankel_weights has not been called");
    return MTIP_OK;
}
static int require_loop(mtip_ctx* c) {
    int r = require_transforms(c);
    if (r) return r;
    if (!c->have_radial) FAIL(c, MTIP_ESTATE, "mtip_set_radial_grid has not been called");
    if (!c->have_support) FAIL(c, MTIP_ESTATE, "mtip_set_initial_support has not been called");
    if (!c->have_errw) FAIL(c, MTIP_ESTATE, "mtip_set_error_weights has not been called");
    for (int l = 0; l <= c->L; ++l)
        if (!c->have_V[l]) FAIL(c, MTIP_ESTATE, "mtip_set_projection_matrix missing for some order");
    if (!c->state_ready) FAIL(c, MTIP_ESTATE, "mtip_init_state has not been called");
    return MTIP_OK;
}

// grid -> grid Fourier transform, fourier_transforms.py:57-85 (in_slot/out via epilogue possible)
static void ft_pipeline(mtip_ctx* c, const double2* in, int in_slot, double2* out, int inverse, int prologue,
                        const InvEpilogue& epi, double2* ca, double2* cb) {
    launch_sht_forward(c, in, ca, prologue, in_slot);
    launch_hankel(c, ca, cb, inverse);
    launch_sht_inverse(c, cb, out, epi);
}

static int ensure_hist(mtip_ctx* c, long long need) {
    if (need <= c->err_cap) return MTIP_OK;
    long long cap = c->err_cap;
    while (cap < need) cap *= 2;
    double *nh = nullptr, *nd = nullptr, *nm = nullptr;
    int r = dev_alloc(c, &nh, (size_t)cap * c->B);
    if (r) return r;
    r = dev_alloc(c, &nd, (size_t)cap * c->B * (c->L + 1));
    if (!r) r = dev_alloc(c, &nm, (size_t)cap * c->B);
    if (r) {
        (void)hipFree(nh);
        if (nd) (void)hipFree(nd);
        return r;
    }
    MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    MTIP_HIP_CHECK(c, hipMemcpyAsync(nh, c->d_err_hist, (size_t)c->n_steps_done * c->B * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    MTIP_HIP_CHECK(c, hipMemcpyAsync(nm, c->d_main_hist, (size_t)c->n_steps_done * c->B * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    MTIP_HIP_CHECK(c, hipMemcpyAsync(nd, c->d_deg2_hist, (size_t)c->n_steps_done * c->B * (c->L + 1) * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(c->d_main_hist);
    c->d_main_hist = nm;
    (void)hipFree(c->d_err_hist);
    (void)hipFree(c->d_deg2_hist);
    c->d_err_hist = nh;
    c->d_deg2_hist = nd;
    if (c->d_rl2_hist) {
        double* ni = nullptr;
        r = dev_alloc(c, &ni, (size_t)cap * c->B);
        if (r) return r;
        MTIP_HIP_CHECK(c, hipMemcpyAsync(ni, c->d_rl2_hist, (size_t)c->n_steps_done * c->B * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
        (void)hipFree(c->d_rl2_hist);
        c->d_rl2_hist = ni;
    }
    if (c->d_im_hist) {
        double* ni = nullptr;
        const size_t rowlen = (size_t)c->B * (2 + c->N);
        r = dev_alloc(c, &ni, (size_t)cap * rowlen);
        if (r) return r;
        MTIP_HIP_CHECK(c, hipMemcpyAsync(ni, c->d_im_hist, (size_t)c->n_steps_done * rowlen * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
        (void)hipFree(c->d_im_hist);
        c->d_im_hist = ni;
    }
    c->err_cap = cap;
    return MTIP_OK;
}

// one phasing step for the whole batch (reference operator order), see file header.  `phases` selects the part to enqueue, so
// that mtip_run_group_async can put other contexts' launches between them: HEAD = F = FT(rho) and I_lm = SHT(|F|^2) (+ metrics),
// PROJ = the reciprocal-space projection of the coefficients (the long latency chain on few CUs), TAIL = everything after it
enum { STEP_HEAD = 1, STEP_PROJ = 2, STEP_TAIL = 4, STEP_ALL = 7 };
static int enqueue_step(mtip_ctx* c, int method, int ft_stab, double beta, int phases = STEP_ALL) {
    const bool fxs = (method == MTIP_HIO || method == MTIP_ER);
    double2 **cc = c->d_c;
    InvEpilogue store;
    // Chained kernels (k_sht_chain.hip): in the fused step every inverse SHT hands its shell to the forward SHT that reads the
    // same grid next -- F -> SHT(|F|^2), F' -> SHT(F'), rho_new -> SHT(rho_new) of the NEXT step (kept in d_c0n, valid while
    // nothing but k_finish_step's rotation touches the current density)
    const bool one_pass_diff = c->cfg.fused && ft_stab && hankel_has_difference(c) && sht_inverse_fuses_real_update(c);
    const bool chain = c->cfg.fused && sht_chain_supported(c);
    if (phases & STEP_HEAD) {
        // 1  F = FT(rho_cur)
        const double2* c0 = cc[0];
        if (chain && c->c0n_valid) c0 = c->d_c0n;
        else launch_sht_forward(c, c->d_rho, cc[0], MTIP_PRE_NONE, SL_CUR);
        c->c0n_valid = false;
        launch_hankel(c, c0, cc[1], 0);
        if (chain && fxs) launch_sht_chain(c, cc[1], c->d_F, store, MTIP_PRE_SQUARE, cc[2]);
        else launch_sht_inverse(c, cc[1], c->d_F, store);
        if (fxs) {
            // 2-3 I_lm = SHT(|F|^2)
            if (!chain) launch_sht_forward(c, c->d_F, cc[2], MTIP_PRE_SQUARE);
            if (c->deg2_enable) launch_deg2_metric(c, cc[2], c->d_deg2_hist + (size_t)c->n_steps_done * c->B * (c->L + 1));
            if (c->im_which) {
                const int rm = launch_invariant_metrics(c, cc[2], c->n_steps_done);
                if (rm != MTIP_OK) return rm;
            }
        }
    }
    if (phases & STEP_PROJ) {
        // 4-5 projection
        if (fxs) {
            const int rp = launch_project_coefficients(c, cc[2], cc[2], true);    // in place: I_lm is not needed afterwards; SHT of the real |F|^2
            if (rp != MTIP_OK) return rp;
        } else {
            launch_modulus_fixed_slots(c, c->d_F);
        }
    }
    if (!(phases & STEP_TAIL)) return MTIP_OK;
    if (fxs) {
        // 6-7 I' = iSHT, F' = F sqrt(I'/I) -> Fp[out]
        InvEpilogue mod;
        mod.mode = EPI_MODULUS;
        mod.F = c->d_F;
        mod.out_slot = SL_OUT;
        if (chain) launch_sht_chain(c, cc[2], c->d_Fp, mod, MTIP_PRE_NONE, cc[4]);
        else launch_sht_inverse(c, cc[2], c->d_Fp, mod);
        launch_reciprocal_l2_metric(c, c->d_F, c->d_Fp, c->n_steps_done);     // (non-default metric; no launch unless enabled)
    }
    // 9  rho' = IFT(F')
    if (!(chain && fxs)) launch_sht_forward(c, c->d_Fp, cc[4], MTIP_PRE_NONE, SL_OUT);
    // rho'' = rho + IFT(F' - F) on shells > 0, IFT(F') on shell 0, using SHT(F) == Hankel(SHT(rho)) = cc[1]: with the
    // workgroup-tiled Hankel kernel the difference is taken on load and shell 0 corrected in the same pass
    // per-restart ft_stab (mtip_set_ft_stab_mask: the reference decides the link to enforce_initial_support per reconstruction
    // process, reconstruct.py:836-850): the restarts without it neither subtract SHT(F) here nor add rho_prev back in the epilogue
    const uint8_t* ftm = (ft_stab && c->ftmask_mixed) ? c->d_ftmask : nullptr;
    if (ftm && !(one_pass_diff && sht_inverse_fuses_real_update(c)))
        FAIL(c, MTIP_ESTATE, "a per-restart ft_stab mask needs the fused one-pass step (cfg.fused, tiled Hankel kernel, real update in the SHT epilogue)");
    if (one_pass_diff) {
        ProfScope ps(c, "hankel");
        launch_hankel_mfma_sub(c, cc[4], cc[1], cc[5], 1, ftm);
    } else {
        launch_hankel(c, cc[4], cc[5], 1);
    }
    if (c->cfg.fused && ft_stab) {
        if (!one_pass_diff) launch_hankel(c, cc[1], cc[0], 1);
        if (sht_inverse_fuses_real_update(c)) {
            // coefficient difference on load, constraints + HIO/ER + error sums in the epilogue: the density of
            // this step is written once and nothing else of grid size moves
            InvEpilogue ru;
            ru.mode = EPI_REAL_UPDATE;
            ru.coeff_sub = one_pass_diff ? nullptr : cc[0];
            ru.real.prev = c->d_rho;
            ru.real.out = c->d_rho;
            ru.real.sup = c->d_sup;
            ru.real.S0 = c->d_S0;
            ru.real.wr = c->d_err_wr;
            ru.real.wt = c->d_err_wt;
            ru.real.partial = c->d_partial;
            ru.real.rp = c->rp;
            ru.real.method = method;
            ru.real.err_use_mask = c->err_use_mask;
            ru.real.add_prev = 1;
            ru.real.add_mask = ftm;
            ru.real.beta = beta;
            if (chain && one_pass_diff) {
                // the new density goes to slot SL_OUT, which k_finish_step makes the current one: its SHT is the next step's
                launch_sht_chain(c, cc[5], nullptr, ru, MTIP_PRE_NONE, c->d_c0n);
                launch_finish_step(c, c->n_steps_done, c->N);
                c->c0n_valid = true;
            } else {
                launch_sht_inverse(c, cc[5], nullptr, ru);
                launch_finish_step(c, c->n_steps_done, sht_inverse_real_update_blocks(c));
            }
            c->n_steps_done += 1;
            return MTIP_OK;
        }
        launch_coeff_diff(c, cc[5], cc[0], cc[4]);
        launch_sht_inverse(c, cc[4], c->d_T1, store);
        // prev enters twice: as the add-back (rho_rt = 0 path) -- handled by passing a zero round trip
        launch_real_update(c, c->d_T1, c->d_rho, c->d_T2 /*zeros*/, c->d_rho, method, beta, 1);
    } else {
        launch_sht_inverse(c, cc[5], c->d_T1, store);
        const double2* rt = nullptr;
        if (ft_stab) {
            // rho_rt = IFT(F)
            launch_sht_forward(c, c->d_F, cc[0], MTIP_PRE_NONE);
            launch_hankel(c, cc[0], cc[4], 1);
            launch_sht_inverse(c, cc[4], c->d_T2, store);
            rt = c->d_T2;
        }
        launch_real_update(c, c->d_T1, c->d_rho, rt, c->d_rho, method, beta, 1);
    }
    launch_finish_step(c, c->n_steps_done);
    c->n_steps_done += 1;
    return MTIP_OK;
}

// checks and one-time work in front of the steps of a run (shared by mtip_run_async and mtip_run_group_async)
static int run_prelude(mtip_ctx* c, int method, int ft_stab, int n_steps, const double* betas) {
    int r = require_loop(c);
    if (r) return r;
    if (method < 0 || method > 3) FAIL(c, MTIP_EINVAL, "unknown method");
    if (n_steps < 0 || (n_steps > 0 && !betas)) FAIL(c, MTIP_EINVAL, "bad n_steps / betas");
    (void)hipSetDevice(c->device);
    r = ensure_hist(c, c->n_steps_done + n_steps);
    if (r) return r;
    if (c->deg2_enable && c->bref_dirty) {
        r = build_bref(c);
        if (r) return r;
    }
    const bool fxs = (method == MTIP_HIO || method == MTIP_ER);
    if (c->main_mode == 1 && !(c->deg2_enable && fxs))
        FAIL(c, MTIP_ESTATE, "main error over deg2_invariant_l2_diff needs that metric enabled (mtip_set_deg2_metric) and an FXS method");
    if (!fxs) {
        if (!c->fixed_valid) {
            launch_abs_to_fixed(c);                  // reconstruct.py:899-902
            c->fixed_valid = true;
        }
    } else {
        c->fixed_valid = false;
    }
    // the separate real-space kernel of the fused ft_stab step reads a zero round trip (the add-back is already in the
    // coefficients); the epilogue path does not use T2 at all
    if (c->cfg.fused && ft_stab && !sht_inverse_fuses_real_update(c))
        MTIP_HIP_CHECK(c, hipMemsetAsync(c->d_T2, 0, (size_t)c->B * c->G * sizeof(double2), c->stream));
    return MTIP_OK;
}

int mtip_run_async(mtip_ctx* c, int method, int ft_stab, int n_steps, const double* betas) {
    CTX_CHECK(c);
    int r = run_prelude(c, method, ft_stab, n_steps, betas);
    if (r) return r;
    for (int s = 0; s < n_steps; ++s) {
        r = enqueue_step(c, method, ft_stab, betas[s]);
        if (r) return r;
    }
    return post_launch(c, "mtip_run");
}

// The same steps for SEVERAL contexts of one device (restart groups of one worker, each on its own stream), enqueued so that the
// contexts take TURNS at the chip-filling part of a step.  A step of a context is a long projection on a few CUs (k_rproj:
// ~0.3 ms, <= 27 workgroups) followed by transforms that want every CU (TAIL of the step + HEAD of the next: ~0.15 ms); contexts
// that run side by side overlap the projection of one with the transforms of the others.  Left to themselves their transform
// blocks collide, the chip serves colliding kernels by sharing CUs, both blocks finish late and both projections start late
// (measured: one 3-restart context alone 474 us per step; {3, 3, 2} side by side 588 us, the sum of their transform blocks being
// 476 us).  Here block (i, s) = TAIL(i, s) + HEAD(i, s + 1) waits for the block before it in the ring (i - 1, s) / (last, s - 1)
// (or, by default, for the one before that: MTIP_TURN_LAG below) through an event recorded on that context's stream: first come,
// first served.  Events only order work that has been enqueued
// already (this function enqueues in ring order), so nothing can wait for a record that never comes.
int mtip_run_group_async(mtip_ctx* const* ctxs, int n_ctx, int method, int ft_stab, int n_steps, const double* betas) {
    if (!ctxs || n_ctx < 1) return MTIP_EINVAL;
    for (int i = 0; i < n_ctx; ++i) {
        if (!ctxs[i]) return MTIP_EINVAL;
        if (ctxs[i]->device != ctxs[0]->device) FAIL(ctxs[i], MTIP_EINVAL, "contexts of a group must live on one device");
        for (int j = 0; j < i; ++j)
            if (ctxs[j] == ctxs[i]) FAIL(ctxs[i], MTIP_EINVAL, "a context appears twice in the group");
    }
    if (n_ctx == 1) return mtip_run_async(ctxs[0], method, ft_stab, n_steps, betas);
    for (int i = 0; i < n_ctx; ++i) {
        mtip_ctx* c = ctxs[i];
        const int r = run_prelude(c, method, ft_stab, n_steps, betas);
        if (r) return r;
        if (!c->turn_ev) MTIP_HIP_CHECK(c, hipEventCreateWithFlags(&c->turn_ev, hipEventDisableTiming | hipEventDisableSystemFence));
    }
    hipEvent_t prev = nullptr, prev2 = nullptr;     // end of the block before this one in the ring, and of the one before that
    // MTIP_TURN_LAG = 1: a block waits for the block right before it (strict turns: three stream-to-stream hand-overs of ~10 us per
    // period on the critical path, and the half-empty second round of every 384-workgroup kernel is idle chip time); 2 (default): for
    // the one before that -- at most two contexts are in their transforms at a time, one fills the other's tails.  Measured, {3, 3, 2}
    // restarts at 128 x L32: schedule 0.543 ms per step without turns, 0.592 with lag 1, 0.528 with lag 2; the HIO window (the chain of
    // one context is as long as the three transform blocks together) 0.59-0.60 in all three.
    const int lag = getenv("MTIP_TURN_LAG") && atoi(getenv("MTIP_TURN_LAG")) == 1 ? 1 : 2;
    auto turn = [&](mtip_ctx* c, bool tail, bool head, int s) -> int {
        hipEvent_t w = lag >= 2 ? prev2 : prev;
        if (w && w != c->turn_ev) MTIP_HIP_CHECK(c, hipStreamWaitEvent(c->stream, w, 0));
        if (tail) {
            const int r = enqueue_step(c, method, ft_stab, betas[s], STEP_TAIL);
            if (r) return r;
        }
        if (head) {
            const int r = enqueue_step(c, method, ft_stab, betas[s + (tail ? 1 : 0)], STEP_HEAD);
            if (r) return r;
        }
        MTIP_HIP_CHECK(c, hipEventRecord(c->turn_ev, c->stream));
        prev2 = prev;
        prev = c->turn_ev;
        return MTIP_OK;
    };
    if (n_steps > 0)
        for (int i = 0; i < n_ctx; ++i) {
            const int r = turn(ctxs[i], false, true, 0);
            if (r) return r;
        }
    for (int s = 0; s < n_steps; ++s)
        for (int i = 0; i < n_ctx; ++i) {
            mtip_ctx* c = ctxs[i];
            int r = enqueue_step(c, method, ft_stab, betas[s], STEP_PROJ);
            if (r) return r;
            r = turn(c, true, s + 1 < n_steps, s);
            if (r) return r;
        }
    for (int i = 0; i < n_ctx; ++i) {
        const int r = post_launch(ctxs[i], "mtip_run_group");
        if (r) return r;
    }
    return MTIP_OK;
}

int mtip_set_ft_stab_mask(mtip_ctx* c, const uint8_t* mask) {
    CTX_CHECK(c);
    (void)hipSetDevice(c->device);
    bool any = false, all = true;
    if (mask)
        for (int b = 0; b < c->B; ++b) {
            any = any || mask[b] != 0;
            all = all && mask[b] != 0;
        }
    c->ftmask_mixed = mask != nullptr && any && !all;
    if (mask != nullptr && !any) FAIL(c, MTIP_EINVAL, "ft_stab mask without any restart set: run the steps with ft_stab = 0 instead");
    if (c->ftmask_mixed) {
        if (!c->d_ftmask) {
            const int r = dev_alloc(c, &c->d_ftmask, (size_t)c->B);
            if (r) return r;
        }
        MTIP_HIP_CHECK(c, mtip_copy(c, c->d_ftmask, mask, (size_t)c->B, hipMemcpyHostToDevice));
    }
    return MTIP_OK;
}

int mtip_fetch_errors(mtip_ctx* c, int64_t first, int64_t n, double* real_err, double* deg2_err) {
    CTX_CHECK(c);
    if (first < 0 || n < 0 || first + n > c->n_steps_done) FAIL(c, MTIP_EINVAL, "step range out of bounds");
    (void)hipSetDevice(c->device);
    MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    if (real_err && n)
        MTIP_HIP_CHECK(c, mtip_copy(c, real_err, c->d_err_hist + (size_t)first * c->B, (size_t)n * c->B * sizeof(double), hipMemcpyDeviceToHost));
    if (deg2_err && n)
        MTIP_HIP_CHECK(c, mtip_copy(c, deg2_err, c->d_deg2_hist + (size_t)first * c->B * (c->L + 1),
                                    (size_t)n * c->B * (c->L + 1) * sizeof(double), hipMemcpyDeviceToHost));
    return post_launch(c, "mtip_fetch_errors");
}

int mtip_fetch_main_errors(mtip_ctx* c, int64_t first, int64_t n, double* main_err) {
    CTX_CHECK(c);
    if (first < 0 || n < 0 || first + n > c->n_steps_done || !main_err) FAIL(c, MTIP_EINVAL, "step range out of bounds / null output");
    (void)hipSetDevice(c->device);
    MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    const double* src = c->main_mode == 1 ? c->d_main_hist : c->d_err_hist;
    if (n) MTIP_HIP_CHECK(c, mtip_copy(c, main_err, src + (size_t)first * c->B, (size_t)n * c->B * sizeof(double), hipMemcpyDeviceToHost));
    return post_launch(c, "mtip_fetch_main_errors");
}

int mtip_run(mtip_ctx* c, int method, int ft_stab, int n_steps, const double* betas, double* real_err, double* deg2_err) {
    CTX_CHECK(c);
    const long long first = c->n_steps_done;
    int r = mtip_run_async(c, method, ft_stab, n_steps, betas);
    if (r) return r;
    return mtip_fetch_errors(c, first, n_steps, real_err, (c->deg2_enable ? deg2_err : nullptr));
}

// ---- state ------------------------------------------------------------------------------------------------
static int slot_of(mtip_ctx* c, int batch, int which, int* out) {
    std::vector<int> s(SL_N);
    MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    MTIP_HIP_CHECK(c, mtip_copy(c, s.data(), c->d_slot + (size_t)batch * SL_N, SL_N * sizeof(int), hipMemcpyDeviceToHost));
    *out = s[which];
    return MTIP_OK;
}

int mtip_set_density(mtip_ctx* c, int batch, const mtip_cdouble* rho) {
    CTX_CHECK(c);
    if (batch < 0 || batch >= c->B || !rho) FAIL(c, MTIP_EINVAL, "bad batch / null density");
    (void)hipSetDevice(c->device);
    MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_T1 + (size_t)batch * c->G, rho, c->G * sizeof(double2), hipMemcpyHostToDevice));
    return MTIP_OK;
}

int mtip_init_state(mtip_ctx* c) {
    CTX_CHECK(c);
    int r = require_transforms(c);
    if (r) return r;
    (void)hipSetDevice(c->device);
    // reconstruct.py:957-979: F0 = FT(rho0); rho0 <- IFT(F0); history = copies of that pair; best_error = inf
    std::vector<int> slots((size_t)c->B * SL_N);
    MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    MTIP_HIP_CHECK(c, mtip_copy(c, slots.data(), c->d_slot, slots.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (int b = 0; b < c->B; ++b) {
        slots[b * SL_N + SL_CUR] = 0;
        slots[b * SL_N + SL_OUT] = 0;      // temporarily: writes below go to slot 0
        slots[b * SL_N + SL_BEST] = 0;
        slots[b * SL_N + SL_HIST] = 0;
        slots[b * SL_N + SL_HAS_ERR] = 0;
    }
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_slot, slots.data(), slots.size() * sizeof(int), hipMemcpyHostToDevice));
    InvEpilogue to_slot;
    to_slot.out_slot = SL_OUT;
    ft_pipeline(c, c->d_T1, -1, c->d_Fp, 0, MTIP_PRE_NONE, to_slot, c->d_c[0], c->d_c[1]);
    ft_pipeline(c, c->d_Fp, SL_CUR, c->d_rho, 1, MTIP_PRE_NONE, to_slot, c->d_c[0], c->d_c[1]);
    MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    for (int b = 0; b < c->B; ++b) slots[b * SL_N + SL_OUT] = 1;
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_slot, slots.data(), slots.size() * sizeof(int), hipMemcpyHostToDevice));
    std::vector<double> inf(c->B, HUGE_VAL);
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_best_err, inf.data(), c->B * sizeof(double), hipMemcpyHostToDevice));
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_last_err, inf.data(), c->B * sizeof(double), hipMemcpyHostToDevice));
    c->n_steps_done = 0;
    c->fixed_valid = false;
    c->c0n_valid = false;
    c->vr_valid = false;                 // a fresh reconstruction does not warm-start its polar factors
    c->vr_kind = 0;
    c->proj_calls = 0;
    c->state_ready = true;
    return post_launch(c, "mtip_init_state");
}

static int get_grid_slot(mtip_ctx* c, const double2* base, int batch, int which, mtip_cdouble* out) {
    if (batch < 0 || batch >= c->B || !out || which < 0 || which > 1) FAIL(c, MTIP_EINVAL, "bad batch / which / null output");
    (void)hipSetDevice(c->device);
    int s = 0;
    int r = slot_of(c, batch, which == 0 ? SL_CUR : SL_BEST, &s);
    if (r) return r;
    MTIP_HIP_CHECK(c, mtip_copy(c, out, base + ((size_t)s * c->B + batch) * c->G, c->G * sizeof(double2), hipMemcpyDeviceToHost));
    return MTIP_OK;
}

int mtip_get_density(mtip_ctx* c, int batch, int which, mtip_cdouble* rho) {
    CTX_CHECK(c);
    return get_grid_slot(c, c->d_rho, batch, which, rho);
}

int mtip_get_reciprocal_density(mtip_ctx* c, int batch, int which, mtip_cdouble* F) {
    CTX_CHECK(c);
    return get_grid_slot(c, c->d_Fp, batch, which, F);
}

int mtip_get_support(mtip_ctx* c, int batch, int which, uint8_t* support) {
    CTX_CHECK(c);
    if (batch < 0 || batch >= c->B || !support || which < 0 || which > 1) FAIL(c, MTIP_EINVAL, "bad batch / which / null output");
    (void)hipSetDevice(c->device);
    int s = 0;
    int r = slot_of(c, batch, which == 0 ? SL_SUP : SL_SUP_BEST, &s);
    if (r) return r;
    MTIP_HIP_CHECK(c, mtip_copy(c, support, c->d_sup + ((size_t)s * c->B + batch) * c->G, c->G, hipMemcpyDeviceToHost));
    return MTIP_OK;
}

int mtip_set_support(mtip_ctx* c, int batch, const uint8_t* support, int enforce) {
    CTX_CHECK(c);
    if (batch < 0 || batch >= c->B || !support) FAIL(c, MTIP_EINVAL, "bad batch / null support");
    (void)hipSetDevice(c->device);
    std::vector<int> s(SL_N);
    MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    MTIP_HIP_CHECK(c, mtip_copy(c, s.data(), c->d_slot + (size_t)batch * SL_N, SL_N * sizeof(int), hipMemcpyDeviceToHost));
    int fs = 0;
    while (fs == s[SL_SUP] || fs == s[SL_SUP_BEST]) ++fs;
    // fxs_Projections.py:53-58: effective support = S0 & support when the initial support is enforced
    std::vector<uint8_t> eff(support, support + c->G);
    if (enforce) {
        std::vector<uint8_t> s0(c->G);
        MTIP_HIP_CHECK(c, mtip_copy(c, s0.data(), c->d_S0, c->G, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < c->G; ++i) eff[i] = (eff[i] && s0[i]) ? 1 : 0;
    }
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_sup + ((size_t)fs * c->B + batch) * c->G, eff.data(), c->G, hipMemcpyHostToDevice));
    s[SL_SUP] = fs;
    s[SL_ENFORCE] = enforce ? 1 : 0;
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_slot + (size_t)batch * SL_N, s.data(), SL_N * sizeof(int), hipMemcpyHostToDevice));
    launch_pack_masks(c);
    return post_launch(c, "mtip_set_support");
}

int mtip_get_unknowns(mtip_ctx* c, int batch, int l, mtip_cdouble* U) {
    CTX_CHECK(c);
    if (batch < 0 || batch >= c->B || l < 0 || l > c->L || !U) FAIL(c, MTIP_EINVAL, "bad batch / order / null output");
    (void)hipSetDevice(c->device);
    MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    MTIP_HIP_CHECK(c, mtip_copy(c, U, c->d_U + (size_t)batch * c->xtot + c->xoff[l], (size_t)c->kl[l] * (2 * l + 1) * sizeof(double2), hipMemcpyDeviceToHost));
    return MTIP_OK;
}

int mtip_get_best_error(mtip_ctx* c, double* best, int64_t* n_steps_done) {
    CTX_CHECK(c);
    (void)hipSetDevice(c->device);
    MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    if (best) MTIP_HIP_CHECK(c, mtip_copy(c, best, c->d_best_err, c->B * sizeof(double), hipMemcpyDeviceToHost));
    if (n_steps_done) *n_steps_done = c->n_steps_done;
    return MTIP_OK;
}

int mtip_select_best(mtip_ctx* c) { return mtip_select_best_where(c, nullptr); }

int mtip_select_best_where(mtip_ctx* c, const uint8_t* select) {
    CTX_CHECK(c);
    (void)hipSetDevice(c->device);
    std::vector<int> slots((size_t)c->B * SL_N);
    MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    MTIP_HIP_CHECK(c, mtip_copy(c, slots.data(), c->d_slot, slots.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (int b = 0; b < c->B; ++b) {
        if (select != nullptr && !select[b]) continue;
        int* s = &slots[(size_t)b * SL_N];
        s[SL_CUR] = s[SL_BEST];
        s[SL_HIST] = s[SL_BEST];
        s[SL_SUP] = s[SL_SUP_BEST];
        int nxt = 0;
        while (nxt == s[SL_CUR] || nxt == s[SL_BEST]) ++nxt;
        s[SL_OUT] = nxt;
    }
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_slot, slots.data(), slots.size() * sizeof(int), hipMemcpyHostToDevice));
    c->fixed_valid = false;
    c->c0n_valid = false;                // the current density is another one now
    return MTIP_OK;
}

// ---- shrink wrap (sketch SW, reconstruct.py:598-605) ----------------------------------------------------------
int mtip_shrinkwrap(mtip_ctx* c, double sigma, double threshold, double error_limit, uint8_t* enforced) {
    CTX_CHECK(c);
    int r = require_loop(c);
    if (r) return r;
    (void)hipSetDevice(c->device);
    // G_sigma(q) = sigma sqrt(2 pi) exp(-2 pi^2 sigma^2 q^4)   (mathLibrary.py:616-624, sic: q^4)
    std::vector<double> q(c->N), gq(c->N);
    MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    MTIP_HIP_CHECK(c, mtip_copy(c, q.data(), c->d_q, c->N * sizeof(double), hipMemcpyDeviceToHost));
    const double pi = 3.14159265358979323846;
    const double a = 1.0 / (2.0 * sigma * sigma);
    for (int i = 0; i < c->N; ++i) {
        const double q2 = q[i] * q[i];
        gq[i] = std::sqrt(pi / a) * std::exp(-pi * pi * (q2 * q2) / a);
    }
    // own buffer: d_fixed may hold the *_non_FXS amplitudes, which survive support updates (reconstruct.py:898-904)
    double* d_gq = c->d_gq;
    MTIP_HIP_CHECK(c, mtip_copy(c, d_gq, gq.data(), c->N * sizeof(double), hipMemcpyHostToDevice));
    InvEpilogue scale, store;
    scale.mode = EPI_SCALE_SHELL;
    scale.shell_scale = d_gq;
    // |rho| -> FT -> * G_sigma
    ft_pipeline(c, c->d_rho, SL_CUR, c->d_T1, 0, MTIP_PRE_ABS, scale, c->d_c[0], c->d_c[1]);
    // IFT
    ft_pipeline(c, c->d_T1, -1, c->d_T2, 1, MTIP_PRE_NONE, store, c->d_c[0], c->d_c[1]);
    double* tmp = reinterpret_cast<double*>(c->d_T1);
    launch_sw_clamp(c, c->d_T2, tmp);
    launch_sw_threshold(c, tmp, threshold, error_limit);
    r = post_launch(c, "mtip_shrinkwrap");
    if (r) return r;
    if (enforced) {
        std::vector<int> slots((size_t)c->B * SL_N);
        MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
        MTIP_HIP_CHECK(c, mtip_copy(c, slots.data(), c->d_slot, slots.size() * sizeof(int), hipMemcpyDeviceToHost));
        for (int b = 0; b < c->B; ++b) enforced[b] = (uint8_t)slots[(size_t)b * SL_N + SL_ENFORCE];
    }
    return MTIP_OK;
}

int mtip_begin_sub_loop(mtip_ctx* c) {
    CTX_CHECK(c);
    (void)hipSetDevice(c->device);
    // reconstruct.py:859, 866: `hist` is read from the state and latest_intensity reset at the top of every sub-loop call
    std::vector<int> slots((size_t)c->B * SL_N);
    MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    MTIP_HIP_CHECK(c, mtip_copy(c, slots.data(), c->d_slot, slots.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (int b = 0; b < c->B; ++b) slots[(size_t)b * SL_N + SL_HIST] = slots[(size_t)b * SL_N + SL_CUR];
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_slot, slots.data(), slots.size() * sizeof(int), hipMemcpyHostToDevice));
    c->fixed_valid = false;
    return MTIP_OK;
}

int mtip_refresh_reciprocal_density(mtip_ctx* c) {
    CTX_CHECK(c);
    int r = require_loop(c);
    if (r) return r;
    (void)hipSetDevice(c->device);
    // The reference's SW_center process returns (support, copy(rho), FT(rho)) and the loop unpacks it as
    // (support, ft_density, density) (reconstruct.py:606-613, 891): the pair appended to the history is (rho, FT(rho)) --
    // its "real" half is FT(rho).  The new history is hist[1:] + (pair,) with the stale `hist` (893): everything after the
    // pair `hist` ends with (SL_HIST) is dropped.  So the new pair overwrites the latest pair in place when that one was
    // produced after `hist` was read and is not the best pair, else it takes the slot that holds neither SL_HIST nor the
    // best pair.
    std::vector<int> slots((size_t)c->B * SL_N);
    MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    MTIP_HIP_CHECK(c, mtip_copy(c, slots.data(), c->d_slot, slots.size() * sizeof(int), hipMemcpyDeviceToHost));
    InvEpilogue store;
    c->c0n_valid = false;                // the real half of the latest pair becomes FT(rho)
    ft_pipeline(c, c->d_rho, SL_CUR, c->d_T1, 0, MTIP_PRE_NONE, store, c->d_c[0], c->d_c[1]);
    for (int b = 0; b < c->B; ++b) {
        int* s = slots.data() + (size_t)b * SL_N;
        const int cur = s[SL_CUR], hist = s[SL_HIST], best = s[SL_BEST];
        int t = cur;
        if (cur == hist || cur == best) {
            t = 0;
            while (t == hist || t == best) ++t;
        }
        MTIP_HIP_CHECK(c, hipMemcpyAsync(c->d_Fp + ((size_t)t * c->B + b) * c->G, c->d_rho + ((size_t)cur * c->B + b) * c->G,
                                         c->G * sizeof(double2), hipMemcpyDeviceToDevice, c->stream));
        MTIP_HIP_CHECK(c, hipMemcpyAsync(c->d_rho + ((size_t)t * c->B + b) * c->G, c->d_T1 + (size_t)b * c->G,
                                         c->G * sizeof(double2), hipMemcpyDeviceToDevice, c->stream));
        s[SL_CUR] = t;
        int f = 0;
        while (f == s[SL_CUR] || f == s[SL_BEST]) ++f;
        s[SL_OUT] = f;
    }
    MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_slot, slots.data(), slots.size() * sizeof(int), hipMemcpyHostToDevice));
    return post_launch(c, "mtip_refresh_reciprocal_density");
}

int mtip_last_deg2_invariant(mtip_ctx* c, int batch, mtip_cdouble* Bl) {
    CTX_CHECK(c);
    int r = require_transforms(c);
    if (r) return r;
    if (batch < 0 || batch >= c->B || !Bl) FAIL(c, MTIP_EINVAL, "bad batch / null output");
    (void)hipSetDevice(c->device);
    const size_t per = (size_t)(c->L + 1) * c->N * c->N;
    if (!c->d_Bl) {
        r = dev_alloc(c, &c->d_Bl, (size_t)c->B * per);
        if (r) return r;
    }
    InvEpilogue store;
    ft_pipeline(c, c->d_rho, SL_CUR, c->d_T1, 0, MTIP_PRE_NONE, store, c->d_c[0], c->d_c[1]);
    launch_sht_forward(c, c->d_T1, c->d_c[2], MTIP_PRE_SQUARE);
    launch_deg2(c, c->d_c[2], c->d_Bl);
    MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    MTIP_HIP_CHECK(c, mtip_copy(c, Bl, c->d_Bl + (size_t)batch * per, per * sizeof(double2), hipMemcpyDeviceToHost));
    return post_launch(c, "mtip_last_deg2_invariant");
}

// ---- single operators on host arrays ---------------------------------------------------------------------
#define H2D(dst, src, n) MTIP_HIP_CHECK(c, mtip_copy(c, (dst), (src), (n), hipMemcpyHostToDevice))
#define D2H(dst, src, n) MTIP_HIP_CHECK(c, mtip_copy(c, (dst), (src), (n), hipMemcpyDeviceToHost))
#define SYNC() MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream))

int mtip_op_sht_forward(mtip_ctx* c, const mtip_cdouble* grid, mtip_cdouble* coeff, int prologue) {
    CTX_CHECK(c);
    if (!c->have_angular) FAIL(c, MTIP_ESTATE, "mtip_set_angular_grid has not been called");
    if (!grid || !coeff || prologue < 0 || prologue > 2) FAIL(c, MTIP_EINVAL, "null buffer / bad prologue");
    (void)hipSetDevice(c->device);
    SYNC();
    H2D(c->d_T1, grid, (size_t)c->B * c->G * sizeof(double2));
    launch_sht_forward(c, c->d_T1, c->d_c[0], prologue);
    SYNC();
    D2H(coeff, c->d_c[0], (size_t)c->B * c->C * sizeof(double2));
    return post_launch(c, "mtip_op_sht_forward");
}

int mtip_op_sht_inverse(mtip_ctx* c, const mtip_cdouble* coeff, mtip_cdouble* grid) {
    CTX_CHECK(c);
    if (!c->have_angular) FAIL(c, MTIP_ESTATE, "mtip_set_angular_grid has not been called");
    if (!grid || !coeff) FAIL(c, MTIP_EINVAL, "null buffer");
    (void)hipSetDevice(c->device);
    SYNC();
    H2D(c->d_c[0], coeff, (size_t)c->B * c->C * sizeof(double2));
    InvEpilogue store;
    launch_sht_inverse(c, c->d_c[0], c->d_T1, store);
    SYNC();
    D2H(grid, c->d_T1, (size_t)c->B * c->G * sizeof(double2));
    return post_launch(c, "mtip_op_sht_inverse");
}

int mtip_op_sht_inverse_forward(mtip_ctx* c, const mtip_cdouble* coeff, mtip_cdouble* grid, mtip_cdouble* coeff_out, int prologue) {
    CTX_CHECK(c);
    if (!c->have_angular) FAIL(c, MTIP_ESTATE, "mtip_set_angular_grid has not been called");
    if (!grid || !coeff || !coeff_out || prologue < 0 || prologue > 1) FAIL(c, MTIP_EINVAL, "null buffer / bad prologue");
    (void)hipSetDevice(c->device);
    SYNC();
    H2D(c->d_c[0], coeff, (size_t)c->B * c->C * sizeof(double2));
    InvEpilogue store;
    if (sht_chain_supported(c)) {
        launch_sht_chain(c, c->d_c[0], c->d_T1, store, prologue, c->d_c[1]);
    } else {
        launch_sht_inverse(c, c->d_c[0], c->d_T1, store);
        launch_sht_forward(c, c->d_T1, c->d_c[1], prologue);
    }
    SYNC();
    D2H(grid, c->d_T1, (size_t)c->B * c->G * sizeof(double2));
    D2H(coeff_out, c->d_c[1], (size_t)c->B * c->C * sizeof(double2));
    return post_launch(c, "mtip_op_sht_inverse_forward");
}

int mtip_op_hankel(mtip_ctx* c, const mtip_cdouble* in, mtip_cdouble* out, int inverse) {
    CTX_CHECK(c);
    if (!c->have_weights) FAIL(c, MTIP_ESTATE, "mtip_set_hankel_weights has not been called");
    if (!in || !out) FAIL(c, MTIP_EINVAL, "null buffer");
    (void)hipSetDevice(c->device);
    SYNC();
    H2D(c->d_c[0], in, (size_t)c->B * c->C * sizeof(double2));
    launch_hankel(c, c->d_c[0], c->d_c[1], inverse ? 1 : 0);
    SYNC();
    D2H(out, c->d_c[1], (size_t)c->B * c->C * sizeof(double2));
    return post_launch(c, "mtip_op_hankel");
}

int mtip_op_fourier_transform(mtip_ctx* c, const mtip_cdouble* in, mtip_cdouble* out, int inverse) {
    CTX_CHECK(c);
    int r = require_transforms(c);
    if (r) return r;
    if (!in || !out) FAIL(c, MTIP_EINVAL, "null buffer");
    (void)hipSetDevice(c->device);
    SYNC();
    H2D(c->d_T1, in, (size_t)c->B * c->G * sizeof(double2));
    InvEpilogue store;
    ft_pipeline(c, c->d_T1, -1, c->d_T2, inverse ? 1 : 0, MTIP_PRE_NONE, store, c->d_c[0], c->d_c[1]);
    SYNC();
    D2H(out, c->d_T2, (size_t)c->B * c->G * sizeof(double2));
    return post_launch(c, "mtip_op_fourier_transform");
}

static int op_project(mtip_ctx* c, const mtip_cdouble* Ilm, mtip_cdouble* out, bool real_intensity) {
    CTX_CHECK(c);
    if (!c->have_radial) FAIL(c, MTIP_ESTATE, "mtip_set_radial_grid has not been called");
    for (int l = 0; l <= c->L; ++l)
        if (!c->have_V[l]) FAIL(c, MTIP_ESTATE, "mtip_set_projection_matrix missing for some order");
    if (!Ilm || !out) FAIL(c, MTIP_EINVAL, "null buffer");
    (void)hipSetDevice(c->device);
    SYNC();
    H2D(c->d_c[2], Ilm, (size_t)c->B * c->C * sizeof(double2));
    const int rp = launch_project_coefficients(c, c->d_c[2], c->d_c[3], real_intensity);
    if (rp != MTIP_OK) return rp;
    SYNC();
    D2H(out, c->d_c[3], (size_t)c->B * c->C * sizeof(double2));
    return post_launch(c, "mtip_op_project_coefficients");
}

int mtip_op_project_coefficients(mtip_ctx* c, const mtip_cdouble* Ilm, mtip_cdouble* out) { return op_project(c, Ilm, out, false); }

int mtip_op_project_real_intensity(mtip_ctx* c, const mtip_cdouble* Ilm, mtip_cdouble* out) { return op_project(c, Ilm, out, true); }

int mtip_op_apply_unknowns(mtip_ctx* c, const mtip_cdouble* Ilm, const mtip_cdouble* U, mtip_cdouble* out) {
    CTX_CHECK(c);
    for (int l = 0; l <= c->L; ++l)
        if (!c->have_V[l]) FAIL(c, MTIP_ESTATE, "mtip_set_projection_matrix missing for some order");
    if (!Ilm || !U || !out) FAIL(c, MTIP_EINVAL, "null buffer");
    (void)hipSetDevice(c->device);
    SYNC();
    H2D(c->d_c[2], Ilm, (size_t)c->B * c->C * sizeof(double2));
    H2D(c->d_U, U, (size_t)c->B * c->xtot * sizeof(double2));
    c->vr_valid = false;                     // d_U no longer belongs to the carried right singular vectors
    c->vr_kind = 0;
    const int rp = launch_apply_unknowns(c, c->d_c[2], c->d_c[3]);
    if (rp != MTIP_OK) return rp;
    SYNC();
    D2H(out, c->d_c[3], (size_t)c->B * c->C * sizeof(double2));
    return post_launch(c, "mtip_op_apply_unknowns");
}

int mtip_op_modulus_replacement(mtip_ctx* c, const mtip_cdouble* F, const mtip_cdouble* I_new, mtip_cdouble* F_new) {
    CTX_CHECK(c);
    if (!F || !I_new || !F_new) FAIL(c, MTIP_EINVAL, "null buffer");
    (void)hipSetDevice(c->device);
    SYNC();
    H2D(c->d_F, F, (size_t)c->B * c->G * sizeof(double2));
    H2D(c->d_T1, I_new, (size_t)c->B * c->G * sizeof(double2));
    launch_modulus_plain(c, c->d_F, c->d_T1, c->d_T2);
    SYNC();
    D2H(F_new, c->d_T2, (size_t)c->B * c->G * sizeof(double2));
    return post_launch(c, "mtip_op_modulus_replacement");
}

int mtip_op_real_space_update(mtip_ctx* c, const mtip_cdouble* w, const mtip_cdouble* rho_prev, int method, double beta,
                              mtip_cdouble* rho_new, double* error) {
    CTX_CHECK(c);
    if (!c->have_support) FAIL(c, MTIP_ESTATE, "mtip_set_initial_support has not been called");
    if (!c->have_errw) FAIL(c, MTIP_ESTATE, "mtip_set_error_weights has not been called");
    if (!w || !rho_prev || !rho_new) FAIL(c, MTIP_EINVAL, "null buffer");
    if (method < 0 || method > 3) FAIL(c, MTIP_EINVAL, "unknown method");
    (void)hipSetDevice(c->device);
    SYNC();
    H2D(c->d_T1, w, (size_t)c->B * c->G * sizeof(double2));
    H2D(c->d_T2, rho_prev, (size_t)c->B * c->G * sizeof(double2));
    launch_real_update(c, c->d_T1, c->d_T2, nullptr, c->d_F, method, beta, 0);
    launch_finish_step(c, -1);
    SYNC();
    D2H(rho_new, c->d_F, (size_t)c->B * c->G * sizeof(double2));
    if (error) D2H(error, c->d_op_err, c->B * sizeof(double));
    return post_launch(c, "mtip_op_real_space_update");
}

int mtip_op_deg2_invariants(mtip_ctx* c, const mtip_cdouble* Ilm, mtip_cdouble* Bl) {
    CTX_CHECK(c);
    if (!Ilm || !Bl) FAIL(c, MTIP_EINVAL, "null buffer");
    (void)hipSetDevice(c->device);
    const size_t per = (size_t)(c->L + 1) * c->N * c->N;
    if (!c->d_Bl) {
        int r = dev_alloc(c, &c->d_Bl, (size_t)c->B * per);
        if (r) return r;
    }
    SYNC();
    H2D(c->d_c[2], Ilm, (size_t)c->B * c->C * sizeof(double2));
    launch_deg2(c, c->d_c[2], c->d_Bl);
    SYNC();
    D2H(Bl, c->d_Bl, (size_t)c->B * per * sizeof(double2));
    return post_launch(c, "mtip_op_deg2_invariants");
}

int mtip_op_apply_matrix(mtip_ctx* c, const double* matrix, const double* vects, double* out, int nr, int nc, int nv) {
    CTX_CHECK(c);
    if (!matrix || !vects || !out || nr < 1 || nc < 1 || nv < 1) FAIL(c, MTIP_EINVAL, "bad apply_matrix arguments");
    (void)hipSetDevice(c->device);
    double *dM = nullptr, *dx = nullptr, *dy = nullptr;
    int r = dev_alloc(c, &dM, (size_t)nr * nc);
    if (!r) r = dev_alloc(c, &dx, (size_t)nc * nv);
    if (!r) r = dev_alloc(c, &dy, (size_t)nr * nv);
    if (!r) {
        hipError_t e = mtip_copy(c, dM, matrix, (size_t)nr * nc * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = mtip_copy(c, dx, vects, (size_t)nc * nv * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            launch_apply_matrix(c, dM, dx, dy, nr, nc, nv);
            e = hipStreamSynchronize(c->stream);
        }
        if (e == hipSuccess) e = mtip_copy(c, out, dy, (size_t)nr * nv * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            c->err = std::string("apply_matrix: ") + hipGetErrorString(e);
            r = MTIP_EHIP;
        }
    }
    if (dM) (void)hipFree(dM);
    if (dx) (void)hipFree(dx);
    if (dy) (void)hipFree(dy);
    return r ? r : post_launch(c, "mtip_op_apply_matrix");
}

// ---- timing ---------------------------------------------------------------------------------------------
int mtip_profile(mtip_ctx* c, int enable) {
    CTX_CHECK(c);
    c->prof = enable ? 1 : 0;                // pending brackets are resolved by mtip_profile_get / _reset
    return MTIP_OK;
}

int mtip_profile_get(mtip_ctx* c, const char* name, double* total_ms, int64_t* launches) {
    CTX_CHECK(c);
    if (!name) return MTIP_EINVAL;
    prof_flush(c);
    auto it = c->prof_data.find(name);
    if (total_ms) *total_ms = it == c->prof_data.end() ? 0.0 : it->second.ms;
    if (launches) *launches = it == c->prof_data.end() ? 0 : it->second.n;
    return MTIP_OK;
}

int mtip_debug_jacobi_sweeps(mtip_ctx* c, int32_t* out) {
    CTX_CHECK(c);
    if (!out) return MTIP_EINVAL;
    (void)hipSetDevice(c->device);
    MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    MTIP_HIP_CHECK(c, mtip_copy(c, out, c->d_sweeps, (size_t)c->B * (c->L + 1) * sizeof(int), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < (size_t)c->B * (c->L + 1); ++i)
        if (out[i] == 0x7fffffff) FAIL(c, MTIP_ESTATE, "k_rproj: an order's LDS layout exceeded the launch's allocation and was skipped");
    return MTIP_OK;
}

int mtip_debug_projection_slots(mtip_ctx* c) {
    CTX_CHECK(c);
    return c->vr_kind == 2 ? c->rp_n_slots : 0;
}

__global__ void k_debug_spin(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
}

int mtip_debug_spin(mtip_ctx* c, double microseconds) {
    CTX_CHECK(c);
    if (!(microseconds >= 0.0) || microseconds > 1e6) FAIL(c, MTIP_EINVAL, "spin time out of range");
    (void)hipSetDevice(c->device);
    int khz = 0;
    MTIP_HIP_CHECK(c, hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device));
    hipLaunchKernelGGL(k_debug_spin, dim3(1), dim3(64), 0, c->stream, (long long)(microseconds * 1e-3 * khz));
    return post_launch(c, "mtip_debug_spin");
}

int mtip_debug_polar_timing(mtip_ctx* c, int64_t* out) {
    CTX_CHECK(c);
    (void)hipSetDevice(c->device);
    const size_t n = (size_t)c->B * (c->L + 1) * MTIP_POLAR_DBG_SLOTS;
    MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    if (!c->d_polar_dbg) {                       // first call: switch the timers on (the next projections fill them)
        int r = dev_alloc(c, &c->d_polar_dbg, n);
        if (r) return r;
        MTIP_HIP_CHECK(c, hipMemsetAsync(c->d_polar_dbg, 0, n * sizeof(long long), c->stream));
    }
    if (out) MTIP_HIP_CHECK(c, mtip_copy(c, out, c->d_polar_dbg, n * sizeof(long long), hipMemcpyDeviceToHost));
    return MTIP_OK;
}

int mtip_debug_chain_timing(mtip_ctx* c, int64_t* out) {
    CTX_CHECK(c);
    (void)hipSetDevice(c->device);
    const size_t n = (size_t)3 * c->B * c->N * MTIP_CHAIN_DBG_SLOTS;
    MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    if (!c->d_chain_dbg) {                       // first call: switch the stamps on (the next chained launches fill them)
        int r = dev_alloc(c, &c->d_chain_dbg, n);
        if (r) return r;
        MTIP_HIP_CHECK(c, hipMemsetAsync(c->d_chain_dbg, 0, n * sizeof(long long), c->stream));
    }
    if (out) MTIP_HIP_CHECK(c, mtip_copy(c, out, c->d_chain_dbg, n * sizeof(long long), hipMemcpyDeviceToHost));
    return MTIP_OK;
}

int mtip_debug_check_jacobi_schedule(mtip_ctx* c, int k_max) {
    CTX_CHECK(c);
    if (k_max < 2 || k_max > 127) FAIL(c, MTIP_EINVAL, "k_max must be in [2, 127]");
    (void)hipSetDevice(c->device);
    MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    const int r = build_jacobi_schedule(c, k_max);
    if (r != MTIP_OK) FAIL(c, r, "pairing schedule failed its verification");
    return MTIP_OK;
}

int mtip_profile_reset(mtip_ctx* c) {
    CTX_CHECK(c);
    prof_flush(c);
    c->prof_data.clear();
    return MTIP_OK;
}

}  // extern "C"

void prof_flush(mtip_ctx* c) {
    if (c->prof_pending.empty()) {
        c->prof_next = 0;
        return;
    }
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& p : c->prof_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->prof_events[p.second], c->prof_events[p.second + 1]) == hipSuccess) {
            auto& e = c->prof_data[p.first];
            e.ms += ms;
            e.n += 1;
        }
    }
    c->prof_pending.clear();
    c->prof_next = 0;
}
