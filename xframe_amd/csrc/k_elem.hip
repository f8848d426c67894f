// Real-space constraints, HIO/ER update, error metric, shrink-wrap support update and state bookkeeping
// (rows a6, a10-a15 of SURVEY section 8).
//   real_projection       xframe/projects/fxs/projectLibrary/fxs_Projections.py:72-130,
//                         xframe/library/pythonLibrary.py:1289-1320 (value threshold)
//   hybrid_input_output   xframe/projects/fxs/projectLibrary/fxs_IO_methods.py:40-64;  error_reduction 67-68
//   l2_projection_diff    fxs_IO_methods.py:97-128 with SphericalIntegrator (xframe/library/mathLibrary.py:1223-1235)
//   add_above_zero_index  xframe/projects/fxs/projectLibrary/misk.py:326-329 (ft_stab add-back)
//   shrink wrap           fxs_Projections.py:245-258, reconstruct.py:877-885
//   best tracking         reconstruct.py:924-939
#include "mtip_internal.h"
#include <cmath>

#define ELEM_ITEMS 4       // grid points per thread in the streaming kernels

struct RealUpdateArgs {
    const double2* rho_p;      // (B,G)   IFT(F')
    const double2* rho_rt;     // (B,G)   IFT(F) or nullptr (no ft_stab)
    const double2* prev;       // previous density: slot base (3,B,G) if use_slots else (B,G)
    double2* out;              // slot base (3,B,G) if use_slots else (B,G)
    const uint8_t* sup;        // support slot base (3,B,G)
    const uint8_t* S0;         // (G)
    const int* slot;           // (B, SL_N)
    const double* wr;          // (Nq) radial error weights
    const double* wt;          // (nt) polar error weights
    double* partial;           // (B, nblk, 2)
    RealParams rp;
    int use_slots, method, err_use_mask;
    double beta;
    int B, nt, np;
    long long G;
};

__global__ void __launch_bounds__(256) k_real_update(RealUpdateArgs a) {
    __shared__ double red0[256];
    __shared__ double red1[256];
    const int b = blockIdx.y;
    const int* sl = a.slot + b * SL_N;
    const size_t gb = (size_t)b * a.G;
    const double2* prev = a.use_slots ? a.prev + ((size_t)sl[SL_CUR] * a.B) * a.G + gb : a.prev + gb;
    double2* out = a.use_slots ? a.out + ((size_t)sl[SL_OUT] * a.B) * a.G + gb : a.out + gb;
    const uint8_t* sup = a.sup + ((size_t)sl[SL_SUP] * a.B) * a.G + gb;
    const double2* rho_p = a.rho_p + gb;
    const double2* rho_rt = a.rho_rt ? a.rho_rt + gb : nullptr;
    const long long shell = (long long)a.nt * a.np;
    double num = 0.0, den = 0.0;
    const long long base = ((long long)blockIdx.x * blockDim.x) * ELEM_ITEMS + threadIdx.x;
    for (int it = 0; it < ELEM_ITEMS; ++it) {
        const long long i = base + (long long)it * blockDim.x;
        if (i < a.G) {
            const int q = (int)(i / shell);
            const int th = (int)((i / a.np) % a.nt);
            const double2 pv = prev[i];
            double2 w = rho_p[i];
            if (rho_rt != nullptr && q > 0) {
                const double2 d = csub(pv, rho_rt[i]);
                w = cadd(w, d);
            }
            const bool S = sup[i] != 0;
            double2 P;
            const double2 nw = real_update_point(a.rp, a.method, a.beta, w, pv, S, P);
            out[i] = nw;
            const bool em = a.err_use_mask ? (a.S0[i] != 0) : true;
            if (em) {
                const double wg = a.wr[q] * a.wt[th];
                const double dx = w.x - P.x, dy = w.y - P.y;
                num = fma(wg, dx * dx + dy * dy, num);
                den = fma(wg, w.x * w.x + w.y * w.y, den);
            }
        }
    }
    red0[threadIdx.x] = num;
    red1[threadIdx.x] = den;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            red0[threadIdx.x] += red0[threadIdx.x + s];
            red1[threadIdx.x] += red1[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double* p = a.partial + ((size_t)b * gridDim.x + blockIdx.x) * 2;
        p[0] = red0[0];
        p[1] = red1[0];
    }
}

// one block per restart: finish the error reduction (fixed order -> bitwise reproducible), record it,
// track the best pair (reconstruct.py:934-938) and rotate the pair slots.
// main error (generate_main_error_routine, fxs_IO_methods.py:746-765): main_mode 0 = the real l2 metric itself (any
// reduction of one value), 1 = mean / min / max / prod (main_type 0..3) over the per-order values of the reciprocal
// deg2_invariant_l2_diff metric of this step (one entry per used order, -1 where the reference invariant vanishes).
__global__ void __launch_bounds__(256) k_finish_step(const double* __restrict__ partial, int nblk, int* __restrict__ slot,
                                                     double* __restrict__ best_err, double* __restrict__ last_err,
                                                     double* __restrict__ err_hist, int B, int update_slots, int main_mode,
                                                     int main_type, const double* __restrict__ deg2_step,
                                                     const int* __restrict__ used, int L, double* __restrict__ main_hist) {
    __shared__ double red0[256];
    __shared__ double red1[256];
    const int b = blockIdx.x;
    double num = 0.0, den = 0.0;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) {
        num += partial[((size_t)b * nblk + i) * 2];
        den += partial[((size_t)b * nblk + i) * 2 + 1];
    }
    red0[threadIdx.x] = num;
    red1[threadIdx.x] = den;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            red0[threadIdx.x] += red0[threadIdx.x + s];
            red1[threadIdx.x] += red1[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double inf = __builtin_huge_val();
        const double real_err = (red1[0] != 0.0) ? red0[0] / red1[0] : inf;    // fxs_IO_methods.py:123-126
        err_hist[b] = real_err;
        double err = real_err;
        if (main_mode == 1 && update_slots) {
            double acc = (main_type == 3) ? 1.0 : (main_type == 1 ? inf : (main_type == 2 ? -inf : 0.0));
            int cnt = 0;
            for (int l = 0; l <= L; ++l) {
                if (!used[l]) continue;
                const double v = deg2_step[(size_t)b * (L + 1) + l];
                if (main_type == 0) acc += v;
                else if (main_type == 1) acc = v < acc ? v : acc;
                else if (main_type == 2) acc = v > acc ? v : acc;
                else acc *= v;
                ++cnt;
            }
            err = (main_type == 0) ? acc / (double)cnt : acc;
            main_hist[b] = err;
        }
        if (update_slots) {
            int* sl = slot + b * SL_N;
            last_err[b] = err;
            sl[SL_HAS_ERR] = 1;
            const int produced = sl[SL_OUT];
            sl[SL_HIST] = sl[SL_CUR];                            // `hist` as re-read at the top of this step (reconstruct.py:913)
            sl[SL_CUR] = produced;
            if (best_err[b] > err) {
                best_err[b] = err;
                sl[SL_BEST] = produced;
                sl[SL_SUP_BEST] = sl[SL_SUP];
            }
            int nxt = 0;
            while (nxt == sl[SL_CUR] || nxt == sl[SL_BEST]) ++nxt;
            sl[SL_OUT] = nxt;
        }
    }
}

void launch_real_update_impl(mtip_ctx* c, const RealUpdateArgs& a0) {
    ProfScope ps(c, "real_update");
    RealUpdateArgs a = a0;
    hipLaunchKernelGGL(k_real_update, dim3((unsigned)c->n_partial_blocks, (unsigned)c->B), dim3(256), 0, c->stream, a);
}

void launch_real_update(mtip_ctx* c, const double2* rho_p, const double2* prev, const double2* rho_rt, double2* out,
                        int method, double beta, int use_slots) {
    RealUpdateArgs a;
    a.rho_p = rho_p;
    a.rho_rt = rho_rt;
    a.prev = prev;
    a.out = out;
    a.sup = c->d_sup;
    a.S0 = c->d_S0;
    a.slot = c->d_slot;
    a.wr = c->d_err_wr;
    a.wt = c->d_err_wt;
    a.partial = c->d_partial;
    a.rp = c->rp;
    a.use_slots = use_slots;
    a.method = method;
    a.err_use_mask = c->err_use_mask;
    a.beta = beta;
    a.B = c->B;
    a.nt = c->nt;
    a.np = c->np;
    a.G = (long long)c->G;
    launch_real_update_impl(c, a);
}

void launch_finish_step(mtip_ctx* c, long long step_index, int nblk) {
    double* hist = step_index >= 0 ? c->d_err_hist + (size_t)step_index * c->B : c->d_op_err;
    const bool loop_step = step_index >= 0;
    const int mode = (loop_step && c->main_mode == 1) ? 1 : 0;
    hipLaunchKernelGGL(k_finish_step, dim3((unsigned)c->B), dim3(256), 0, c->stream, (const double*)c->d_partial,
                       nblk > 0 ? nblk : c->n_partial_blocks, c->d_slot, c->d_best_err, c->d_last_err, hist, c->B, loop_step ? 1 : 0,
                       mode, c->main_type,
                       mode ? (const double*)(c->d_deg2_hist + (size_t)step_index * c->B * (c->L + 1)) : (const double*)nullptr,
                       (const int*)c->d_used, c->L, mode ? c->d_main_hist + (size_t)step_index * c->B : (double*)nullptr);
}

// ---- non-FXS variants: fixed = |F'| of the pair the stale `hist` ends with (reconstruct.py:899-902), F' = F sqrt(fixed/|F|^2) ----
__global__ void __launch_bounds__(256) k_abs_to_fixed(const double2* __restrict__ Fp, const int* __restrict__ slot,
                                                      double* __restrict__ fixed, int B, long long G) {
    const int b = blockIdx.y;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G) return;
    const double2 v = Fp[((size_t)slot[b * SL_N + SL_HIST] * B + b) * G + i];    // np.abs(hist[-1][0]), reconstruct.py:901
    fixed[(size_t)b * G + i] = sqrt(cabs2(v));
}

void launch_abs_to_fixed(mtip_ctx* c) {
    hipLaunchKernelGGL(k_abs_to_fixed, dim3((unsigned)div_up((long long)c->G, 256), (unsigned)c->B), dim3(256), 0, c->stream,
                       (const double2*)c->d_Fp, (const int*)c->d_slot, c->d_fixed, c->B, (long long)c->G);
}

// mode 0: F' = F sqrt(Re Inew / |F|^2)  (fxs_Projections.py:899-909);  mode 1: F' = F sqrt(fixed/|F|^2) (911-923)
__global__ void __launch_bounds__(256) k_modulus(const double2* __restrict__ F, const double2* __restrict__ Inew,
                                                 const double* __restrict__ fixed, double2* __restrict__ out,
                                                 const int* __restrict__ slot, int use_slots, int B, long long G) {
    const int b = blockIdx.y;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G) return;
    const size_t gi = (size_t)b * G + i;
    const double2 Fv = F[gi];
    const double I = cabs2(Fv);
    const double target = Inew ? Inew[gi].x : fixed[gi];
    const bool ok = (I >= 0.0) && (target >= 0.0);
    const double mult = ok ? sqrt(target / I) : 0.0;
    double2* o = use_slots ? out + ((size_t)slot[b * SL_N + SL_OUT] * B + b) * G + i : out + gi;
    *o = cscale(Fv, mult);
}

void launch_modulus_plain(mtip_ctx* c, const double2* F, const double2* Inew, double2* out) {
    hipLaunchKernelGGL(k_modulus, dim3((unsigned)div_up((long long)c->G, 256), (unsigned)c->B), dim3(256), 0, c->stream, F,
                       Inew, (const double*)nullptr, out, (const int*)c->d_slot, 0, c->B, (long long)c->G);
}

void launch_modulus_fixed_slots(mtip_ctx* c, const double2* F) {
    hipLaunchKernelGGL(k_modulus, dim3((unsigned)div_up((long long)c->G, 256), (unsigned)c->B), dim3(256), 0, c->stream, F,
                       (const double2*)nullptr, (const double*)c->d_fixed, c->d_Fp, (const int*)c->d_slot, 1, c->B,
                       (long long)c->G);
}

// copy a plain (B,G) grid into the OUT slot of a (3,B,G) slot array
__global__ void __launch_bounds__(256) k_copy_to_slot(const double2* __restrict__ src, double2* __restrict__ dst,
                                                      const int* __restrict__ slot, int which, int B, long long G) {
    const int b = blockIdx.y;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G) return;
    dst[((size_t)slot[b * SL_N + which] * B + b) * G + i] = src[(size_t)b * G + i];
}

void launch_copy_to_slot(mtip_ctx* c, const double2* src, double2* dst, int which) {
    hipLaunchKernelGGL(k_copy_to_slot, dim3((unsigned)div_up((long long)c->G, 256), (unsigned)c->B), dim3(256), 0, c->stream,
                       src, dst, (const int*)c->d_slot, which, c->B, (long long)c->G);
}

// ---- shrink wrap -------------------------------------------------------------------------------------
// tmp = max(Re conv, 0) and per-block min / max  (fxs_Projections.py:249-254)
__global__ void __launch_bounds__(256) k_sw_clamp(const double2* __restrict__ conv, double* __restrict__ tmp,
                                                  double* __restrict__ minmax, long long G) {
    __shared__ double rmin[256];
    __shared__ double rmax[256];
    const int b = blockIdx.y;
    const double inf = __builtin_huge_val();
    double mn = inf, mx = -inf;
    const long long base = ((long long)blockIdx.x * blockDim.x) * ELEM_ITEMS + threadIdx.x;
    for (int it = 0; it < ELEM_ITEMS; ++it) {
        const long long i = base + (long long)it * blockDim.x;
        if (i < G) {
            double v = conv[(size_t)b * G + i].x;
            if (v < 0.0) v = 0.0;
            tmp[(size_t)b * G + i] = v;
            mn = fmin(mn, v);
            mx = fmax(mx, v);
        }
    }
    rmin[threadIdx.x] = mn;
    rmax[threadIdx.x] = mx;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            rmin[threadIdx.x] = fmin(rmin[threadIdx.x], rmin[threadIdx.x + s]);
            rmax[threadIdx.x] = fmax(rmax[threadIdx.x], rmax[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        minmax[((size_t)b * gridDim.x + blockIdx.x) * 2] = rmin[0];
        minmax[((size_t)b * gridDim.x + blockIdx.x) * 2 + 1] = rmax[0];
    }
}

// new support = tmp >= min + thr (max - min)  (fxs_Projections.py:255-257), combined with the initial
// support when enforced (fxs_Projections.py:53-58; decision reconstruct.py:879-882); written to the free
// support slot -- the slot table itself is switched by k_sw_commit afterwards.
__global__ void __launch_bounds__(256) k_sw_threshold(const double* __restrict__ tmp, const double* __restrict__ minmax,
                                                      int nblk, const uint8_t* __restrict__ S0, uint8_t* __restrict__ sup,
                                                      const int* __restrict__ slot, const double* __restrict__ last_err,
                                                      double threshold, double error_limit, int B, long long G) {
    __shared__ double rmin[256];
    __shared__ double rmax[256];
    const int b = blockIdx.y;
    const double inf = __builtin_huge_val();
    double mn = inf, mx = -inf;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) {
        mn = fmin(mn, minmax[((size_t)b * nblk + i) * 2]);
        mx = fmax(mx, minmax[((size_t)b * nblk + i) * 2 + 1]);
    }
    rmin[threadIdx.x] = mn;
    rmax[threadIdx.x] = mx;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            rmin[threadIdx.x] = fmin(rmin[threadIdx.x], rmin[threadIdx.x + s]);
            rmax[threadIdx.x] = fmax(rmax[threadIdx.x], rmax[threadIdx.x + s]);
        }
        __syncthreads();
    }
    const double vmin = rmin[0], vmax = rmax[0];
    const double cut = vmin + threshold * (vmax - vmin);
    const int* sl = slot + b * SL_N;
    const bool enforce = sl[SL_HAS_ERR] && (last_err[b] > error_limit);
    int free_slot = 0;
    while (free_slot == sl[SL_SUP] || free_slot == sl[SL_SUP_BEST]) ++free_slot;
    uint8_t* dst = sup + ((size_t)free_slot * B + b) * G;
    const long long base = ((long long)blockIdx.x * blockDim.x) * ELEM_ITEMS + threadIdx.x;
    for (int it = 0; it < ELEM_ITEMS; ++it) {
        const long long i = base + (long long)it * blockDim.x;
        if (i < G) {
            bool m = tmp[(size_t)b * G + i] >= cut;
            if (enforce) m = m && (S0[i] != 0);
            dst[i] = m ? 1 : 0;
        }
    }
}

__global__ void k_sw_commit(int* __restrict__ slot, const double* __restrict__ last_err, double error_limit, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int* sl = slot + b * SL_N;
    int free_slot = 0;
    while (free_slot == sl[SL_SUP] || free_slot == sl[SL_SUP_BEST]) ++free_slot;
    sl[SL_ENFORCE] = (sl[SL_HAS_ERR] && (last_err[b] > error_limit)) ? 1 : 0;
    sl[SL_SUP] = free_slot;
}

void launch_sw_clamp(mtip_ctx* c, const double2* conv, double* tmp_real) {
    hipLaunchKernelGGL(k_sw_clamp, dim3((unsigned)c->n_partial_blocks, (unsigned)c->B), dim3(256), 0, c->stream, conv,
                       tmp_real, c->d_minmax, (long long)c->G);
}

void launch_sw_threshold(mtip_ctx* c, const double* tmp_real, double threshold, double error_limit) {
    hipLaunchKernelGGL(k_sw_threshold, dim3((unsigned)c->n_partial_blocks, (unsigned)c->B), dim3(256), 0, c->stream,
                       tmp_real, (const double*)c->d_minmax, c->n_partial_blocks, (const uint8_t*)c->d_S0, c->d_sup,
                       (const int*)c->d_slot, (const double*)c->d_last_err, threshold, error_limit, c->B, (long long)c->G);
    hipLaunchKernelGGL(k_sw_commit, dim3((unsigned)div_up(c->B, 64)), dim3(64), 0, c->stream, c->d_slot,
                       (const double*)c->d_last_err, error_limit, c->B);
    launch_pack_masks(c);
}

// ---- support masks packed for the chained SHT kernel: its step-2 thread (row, n2) owns the points R2 n1 + n2 of its row and
//      reads their support / initial-support bits with ONE 16-bit load (16 byte loads and registers otherwise)
__global__ void __launch_bounds__(256) k_pack_masks(const uint8_t* __restrict__ sup, const uint8_t* __restrict__ S0,
                                                    uint16_t* __restrict__ mk, long long G, long long n_words, int R1, int R2) {
    const long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // word (slot * B + b, row of the grid, n2)
    if (w >= n_words) return;
    const int n2 = (int)(w % R2);
    const long long row = w / R2;                                              // global row over (slot, b, shell, theta)
    const long long Nrow = (long long)R1 * R2;
    const uint8_t* s = sup + row * Nrow + n2;
    const uint8_t* s0 = S0 + (row * Nrow) % G + n2;
    unsigned v = 0;
    for (int n1 = 0; n1 < R1; ++n1) {
        if (s[R2 * n1]) v |= 1u << n1;
        if (s0[R2 * n1]) v |= 1u << (8 + n1);
    }
    mk[w] = (uint16_t)v;
}

void launch_pack_masks(mtip_ctx* c) {
    if (!c->d_mk) return;
    int r1 = 8, r2 = c->np / 8;
    switch (c->np) {                                                           // the register-FFT radix pairs (k_sht_common.h)
        case 16: r1 = 4; r2 = 4; break;
        case 32: r1 = 4; r2 = 8; break;
        case 64: r1 = 8; r2 = 8; break;
        case 128: r1 = 8; r2 = 16; break;
        default: return;                                                       // no chained kernel for this grid
    }
    const long long n_words = 3LL * c->B * (long long)c->G / r1;
    hipLaunchKernelGGL(k_pack_masks, dim3((unsigned)div_up(n_words, 256)), dim3(256), 0, c->stream, (const uint8_t*)c->d_sup,
                       (const uint8_t*)c->d_S0, c->d_mk, (long long)c->G, n_words, r1, r2);
}

// ---- generic y = M x (GPU-process boundary example of the reference docs/tests) -------------------------
__global__ void __launch_bounds__(256) k_apply_matrix(const double* __restrict__ M, const double* __restrict__ x,
                                                      double* __restrict__ y, int nr, int nc, int nv) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)nr * nv) return;
    const int r = (int)(idx / nv), v = (int)(idx % nv);
    double acc = 0.0;
    for (int cidx = 0; cidx < nc; ++cidx) acc += M[(size_t)r * nc + cidx] * x[(size_t)cidx * nv + v];
    y[idx] = acc;
}

void launch_apply_matrix(mtip_ctx* c, const double* M, const double* x, double* y, int nr, int nc, int nv) {
    hipLaunchKernelGGL(k_apply_matrix, dim3((unsigned)div_up((long long)nr * nv, 256)), dim3(256), 0, c->stream, M, x, y, nr,
                       nc, nv);
}
