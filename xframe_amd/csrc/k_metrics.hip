// The reference's non-default reciprocal error metrics, per step on the device from B_l = I_l I_l^+ of the
// current intensity coefficients:
//   II_error   xframe/projects/fxs/projectLibrary/fxs_IO_methods.py:587-627
//   ccd_diff   fxs_IO_methods.py:651-683
//   fqc_error  fxs_IO_methods.py:507-550
// All three are contractions of the masked B_l (L+1, Nq, Nq) with constant tensors the host prepares once (the normalised
// associated Legendre products of fxs_invariant_tools.py:23-33, 48-58, the masked reference invariants): a few MB per restart and
// step, diagnostics that are off by default -- plain kernels, one workgroup per restart (II, ccd) or per (restart, shell) (fqc).
#include "mtip_internal.h"

__device__ __forceinline__ double2 c_sqrt(double2 z) {           // principal branch, as numpy's
    const double r = sqrt(sqrt(z.x * z.x + z.y * z.y));
    const double t = 0.5 * atan2(z.y, z.x);
    double s, c;
    sincos(t, &s, &c);
    return make_double2(r * c, r * s);
}

__device__ __forceinline__ double2 c_div(double2 a, double2 b) {
    const double d = b.x * b.x + b.y * b.y;
    return make_double2((a.x * b.x + a.y * b.y) / d, (a.y * b.x - a.x * b.y) / d);
}

__device__ __forceinline__ double2 block_sum2(double2 v, double2* red) {
    for (int o = 32; o > 0; o >>= 1) {
        v.x += __shfl_xor(v.x, o, 64);
        v.y += __shfl_xor(v.y, o, 64);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double2 s = make_double2(0.0, 0.0);
    for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) s = cadd(s, red[wv]);
    return s;
}

#define IM_BLOCKS 64            // workgroups per restart of the II / ccd sums (partials folded by k_metric_fold)

// II_error = 1 - sum(cur ref qq) / sqrt(sum(cur^2 qq) sum(ref^2 qq)), cur = sum_{l >= 1} masked B_l (complex, not conjugated: 624).
// Workgroup (chunk, restart): partial sums of its (q, q') elements -> part[(b * IM_BLOCKS + chunk) * 4 + 0..2]
__global__ void __launch_bounds__(256) k_metric_II(const double2* __restrict__ Bl, const uint8_t* __restrict__ zmask,
                                                   const double2* __restrict__ ref, const double* __restrict__ qq, double2* __restrict__ part,
                                                   int N, int L) {
    __shared__ double2 red[4];
    const int b = blockIdx.y;
    const size_t NN = (size_t)N * N;
    double2 sa = make_double2(0.0, 0.0), sb = sa, sc = sa;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < NN; e += (size_t)gridDim.x * blockDim.x) {
        double2 cur = make_double2(0.0, 0.0);
        for (int l = 1; l <= L; ++l)
            if (!zmask[(size_t)l * NN + e]) cur = cadd(cur, Bl[((size_t)b * (L + 1) + l) * NN + e]);
        const double w = qq[e];
        const double2 r = ref[e];
        sa = cadd(sa, cscale(cmul(cur, r), w));
        sb = cadd(sb, cscale(cmul(cur, cur), w));
        sc = cadd(sc, cscale(cmul(r, r), w));
    }
    sa = block_sum2(sa, red);
    sb = block_sum2(sb, red);
    sc = block_sum2(sc, red);
    if (threadIdx.x == 0) {
        double2* o = part + ((size_t)b * gridDim.x + blockIdx.x) * 4;
        o[0] = sa; o[1] = sb; o[2] = sc;
    }
}

// ccd_diff = sum |sum_l masked B_l T_l - ref|^2 / norm  (T_l = 0 for order 0 and the orders below C_order): partial sums -> part[.. * 4 + 3]
__global__ void __launch_bounds__(256) k_metric_ccd(const double2* __restrict__ Bl, const uint8_t* __restrict__ zmask,
                                                    const double* __restrict__ T, const double2* __restrict__ ref, double2* __restrict__ part,
                                                    int N, int L) {
    __shared__ double2 red[4];
    const int b = blockIdx.y;
    const size_t NN = (size_t)N * N;
    double2 acc = make_double2(0.0, 0.0);
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < NN; e += (size_t)gridDim.x * blockDim.x) {
        double2 d = make_double2(-ref[e].x, -ref[e].y);
        for (int l = 0; l <= L; ++l)
            if (!zmask[(size_t)l * NN + e]) d = cadd(d, cscale(Bl[((size_t)b * (L + 1) + l) * NN + e], T[(size_t)l * NN + e]));
        acc.x += d.x * d.x + d.y * d.y;
    }
    acc = block_sum2(acc, red);
    if (threadIdx.x == 0) part[((size_t)b * gridDim.x + blockIdx.x) * 4 + 3] = acc;
}

// the partials of a restart in a fixed order -> II_error (which & 1) and ccd_diff (which & 2); one thread per restart
__global__ void k_metric_fold(const double2* __restrict__ part, int nblk, int which, double ccd_inv_norm, double* __restrict__ out_II,
                              double* __restrict__ out_ccd, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double2 sa = make_double2(0.0, 0.0), sb = sa, sc = sa;
    double acc = 0.0;
    for (int k = 0; k < nblk; ++k) {
        const double2* o = part + ((size_t)b * nblk + k) * 4;
        if (which & 1) { sa = cadd(sa, o[0]); sb = cadd(sb, o[1]); sc = cadd(sc, o[2]); }
        if (which & 2) acc += o[3].x;
    }
    if (which & 1) out_II[b] = 1.0 - c_div(sa, c_sqrt(cmul(sb, sc))).x;
    if (which & 2) out_ccd[b] = acc * ccd_inv_norm;
}

// fqc_error[q] = 1 - mean_{q' <= q} fqc[q, q'];  workgroup = shell q, for up to FQC_B restarts at a time (the table P -- 143 MB at
// 128 x L32 -- is read once per chunk of restarts, not once per restart).  A wave half (32 lanes) per q': the lanes run over the index j of
// P[l][q][q'][j] (contiguous: coalesced runs of 8 (L + 1) bytes), two j per lane and pass (j, j + 32: L + 1 = 33 columns are one pass),
// B_l(q, q') is a broadcast load, masked orders enter with weight 0 (independent loads in the l loop), the j-sum is a 32-lane butterfly.
#define FQC_B 4
#define FQC_QP 16               // q' per workgroup: grid (Nq, ceil(Nq / FQC_QP)); fq (B, Nq, Nq) in global memory, folded by k_metric_fqc_fold
__global__ void __launch_bounds__(256) k_metric_fqc(const double2* __restrict__ Bl, const uint8_t* __restrict__ zmask,
                                                    const double* __restrict__ P, const double* __restrict__ ref_avg,
                                                    const double* __restrict__ ref_w, double* __restrict__ fq, int N, int L, int B) {
    const int q = blockIdx.x;
    const size_t NN = (size_t)N * N;
    const int M1 = L + 1;
    const int jl = threadIdx.x & 31, sub = threadIdx.x >> 5;
    for (int b0 = 0; b0 < B; b0 += FQC_B) {
        const int nb = min(FQC_B, B - b0);
        for (int qp0 = blockIdx.y * FQC_QP; qp0 < min((int)(blockIdx.y + 1) * FQC_QP, N); qp0 += 8) {
            const int qp = qp0 + sub;
            const bool live = qp < N;
            const size_t e = (size_t)q * N + (live ? qp : 0);
            double avg[FQC_B];
            double2 ctrl[FQC_B];
#pragma unroll
            for (int bb = 0; bb < FQC_B; ++bb) {
                avg[bb] = 0.0;
                ctrl[bb] = make_double2(0.0, 0.0);
            }
            for (int j0 = 0; j0 < M1; j0 += 64) {
                const int ja = j0 + jl, jb = ja + 32;
                const bool va = ja < M1, vb = jb < M1;
                double2 ca[FQC_B], cb[FQC_B];
#pragma unroll
                for (int bb = 0; bb < FQC_B; ++bb) ca[bb] = cb[bb] = make_double2(0.0, 0.0);
#pragma unroll 2
                for (int l = 1; l <= L; ++l) {
                    const double m = zmask[(size_t)l * NN + e] ? 0.0 : 1.0;
                    const double* pr = P + ((size_t)l * NN + e) * M1;
                    const double pa = va ? m * pr[ja] : 0.0, pb = vb ? m * pr[jb] : 0.0;
                    const double rw = m * ref_w[(size_t)l * NN + e];
#pragma unroll
                    for (int bb = 0; bb < FQC_B; ++bb) {
                        if (bb < nb) {
                            const double2 bv = Bl[((size_t)(b0 + bb) * M1 + l) * NN + e];
                            ca[bb] = cadd(ca[bb], cscale(bv, pa));
                            cb[bb] = cadd(cb[bb], cscale(bv, pb));
                            if (j0 == 0) ctrl[bb] = cadd(ctrl[bb], cscale(bv, rw));
                        }
                    }
                }
#pragma unroll
                for (int bb = 0; bb < FQC_B; ++bb) {
                    // Re(c_0 c_0) + 2 sum |c_j|^2 (521-522)
                    if (va) avg[bb] += ja == 0 ? (ca[bb].x * ca[bb].x - ca[bb].y * ca[bb].y) : 2.0 * (ca[bb].x * ca[bb].x + ca[bb].y * ca[bb].y);
                    if (vb) avg[bb] += 2.0 * (cb[bb].x * cb[bb].x + cb[bb].y * cb[bb].y);
                }
            }
#pragma unroll
            for (int bb = 0; bb < FQC_B; ++bb) {
                double a = avg[bb];
                for (int o = 16; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
                if (live && jl == 0 && bb < nb) {
                    const double prod = a * ref_avg[e];
                    // norm = sqrt(prod); `norm >= 0` is false for NaN (negative prod): fqc = 1 there (536-543); a zero norm divides as numpy does
                    fq[(size_t)(b0 + bb) * NN + e] = prod >= 0.0 ? ctrl[bb].x / sqrt(prod) : 1.0;
                }
            }
        }
    }
}

// out[b][q] = 1 - mean_{q' <= q} fq[b][q][q'] (in the order of q', as the reference's cumulative mean)
__global__ void k_metric_fqc_fold(const double* __restrict__ fq, double* __restrict__ out, int N, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * N) return;
    const int q = i % N;
    const double* row = fq + (size_t)i * N;
    double s = 0.0;
    for (int qp = 0; qp <= q; ++qp) s += row[qp];
    out[i] = 1.0 - s / (double)(q + 1);
}

// reciprocal l2_projection_diff (fxs_IO_methods.py:301-310 -> 131-205 / 96-127): E = int |F - F'|^2 / int |F|^2 over the grid, F the
// amplitude before and F' after the modulus projection.  The cache-aware branch asks for the type 'reziprocal' (303) and so takes the
// REAL grid's integrator, the plain branch the reciprocal grid's: the two radial grids are proportional, the ratio is the same; the
// mask is `True` in both, i.e. shell N - 2 drops out (`square[~True] = 0`) -- the weights the host hands over carry both facts.
// One workgroup per (shell, restart): polar-weighted sums of the shell; a second launch folds the shells with the radial weights.
__global__ void __launch_bounds__(256) k_metric_rl2_shell(const double2* __restrict__ F, const double2* __restrict__ Fp, const int* __restrict__ slot,
                                                          const double* __restrict__ wt, double* __restrict__ part, int B, int N, int nt, int np) {
    const int q = blockIdx.x, b = blockIdx.y;
    const size_t G = (size_t)N * nt * np, sh = (size_t)q * nt * np;
    const double2* f = F + (size_t)b * G + sh;
    const double2* fp = Fp + ((size_t)slot[b * SL_N + SL_OUT] * B + b) * G + sh;
    double sd = 0.0, sv = 0.0;
    for (int i = threadIdx.x; i < nt * np; i += 256) {
        const double w = wt[i / np];
        const double2 a = f[i], p = fp[i];
        const double dx = a.x - p.x, dy = a.y - p.y;
        sd += w * (dx * dx + dy * dy);
        sv += w * (a.x * a.x + a.y * a.y);
    }
    __shared__ double r0[256], r1[256];
    r0[threadIdx.x] = sd;
    r1[threadIdx.x] = sv;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            r0[threadIdx.x] += r0[threadIdx.x + s];
            r1[threadIdx.x] += r1[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[((size_t)b * N + q) * 2] = r0[0];
        part[((size_t)b * N + q) * 2 + 1] = r1[0];
    }
}

__global__ void k_metric_rl2_finish(const double* __restrict__ part, const double* __restrict__ wr, double* __restrict__ out, int B, int N) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double d = 0.0, v = 0.0;
    for (int q = 0; q < N; ++q) {
        d += wr[q] * part[((size_t)b * N + q) * 2];
        v += wr[q] * part[((size_t)b * N + q) * 2 + 1];
    }
    out[b] = v != 0.0 ? d / v : INFINITY;
}

int launch_reciprocal_l2_metric(mtip_ctx* c, const double2* F, const double2* Fp, long long step) {
    if (!c->d_rl2_hist) return MTIP_OK;
    hipLaunchKernelGGL(k_metric_rl2_shell, dim3((unsigned)c->N, (unsigned)c->B), dim3(256), 0, c->stream, F, Fp, (const int*)c->d_slot,
                       (const double*)c->d_rl2_wt, c->d_rl2_part, c->B, c->N, c->nt, c->np);
    hipLaunchKernelGGL(k_metric_rl2_finish, dim3((unsigned)((c->B + 63) / 64)), dim3(64), 0, c->stream, (const double*)c->d_rl2_part,
                       (const double*)c->d_rl2_wr, c->d_rl2_hist + (size_t)step * c->B, c->B, c->N);
    return MTIP_OK;
}

void free_invariant_metrics(mtip_ctx* c) {
    for (void* p : {(void*)c->d_rl2_wr, (void*)c->d_rl2_wt, (void*)c->d_rl2_part, (void*)c->d_rl2_hist})
        if (p) (void)hipFree(p);
    c->d_rl2_wr = nullptr; c->d_rl2_wt = nullptr; c->d_rl2_part = nullptr; c->d_rl2_hist = nullptr;
    if (c->d_im_part) (void)hipFree(c->d_im_part);
    if (c->d_im_fq) (void)hipFree(c->d_im_fq);
    c->d_im_part = nullptr;
    c->d_im_fq = nullptr;
    for (void* p : {(void*)c->d_im_zmask, (void*)c->d_im_IIref, (void*)c->d_im_qq, (void*)c->d_im_ccdT, (void*)c->d_im_ccdref, (void*)c->d_im_P,
                    (void*)c->d_im_refavg, (void*)c->d_im_refw, (void*)c->d_im_hist})
        if (p) (void)hipFree(p);
    c->d_im_zmask = nullptr; c->d_im_IIref = nullptr; c->d_im_qq = nullptr; c->d_im_ccdT = nullptr; c->d_im_ccdref = nullptr;
    c->d_im_P = nullptr; c->d_im_refavg = nullptr; c->d_im_refw = nullptr; c->d_im_hist = nullptr;
    c->im_which = 0;
}

// per step: B_l of the current coefficients, then the enabled metrics into the step's row of the history
// (row layout: II (B) | ccd (B) | fqc (B, Nq))
static int launch_invariant_metrics_row(mtip_ctx* c, const double2* Ilm, double* row);

int launch_invariant_metrics(mtip_ctx* c, const double2* Ilm, long long step) {
    if (!c->im_which) return MTIP_OK;
    return launch_invariant_metrics_row(c, Ilm, c->d_im_hist + (size_t)step * c->B * (2 + c->N));
}

static int launch_invariant_metrics_row(mtip_ctx* c, const double2* Ilm, double* row) {
    const size_t per = (size_t)(c->L + 1) * c->N * c->N;
    if (!c->d_Bl && hipMalloc((void**)&c->d_Bl, (size_t)c->B * per * sizeof(double2)) != hipSuccess) {
        c->err = "invariant metrics: out of device memory for B_l";
        return MTIP_ENOMEM;
    }
    launch_deg2(c, Ilm, c->d_Bl);
    if ((c->im_which & 3) && !c->d_im_part && hipMalloc((void**)&c->d_im_part, (size_t)c->B * IM_BLOCKS * 4 * sizeof(double2)) != hipSuccess) {
        c->err = "invariant metrics: out of device memory for the partial sums";
        return MTIP_ENOMEM;
    }
    if (c->im_which & 1)
        hipLaunchKernelGGL(k_metric_II, dim3(IM_BLOCKS, (unsigned)c->B), dim3(256), 0, c->stream, (const double2*)c->d_Bl,
                           (const uint8_t*)c->d_im_zmask, (const double2*)c->d_im_IIref, (const double*)c->d_im_qq, c->d_im_part, c->N, c->L);
    if (c->im_which & 2)
        hipLaunchKernelGGL(k_metric_ccd, dim3(IM_BLOCKS, (unsigned)c->B), dim3(256), 0, c->stream, (const double2*)c->d_Bl,
                           (const uint8_t*)c->d_im_zmask, (const double*)c->d_im_ccdT, (const double2*)c->d_im_ccdref, c->d_im_part, c->N, c->L);
    if (c->im_which & 3)
        hipLaunchKernelGGL(k_metric_fold, dim3((unsigned)((c->B + 63) / 64)), dim3(64), 0, c->stream, (const double2*)c->d_im_part, IM_BLOCKS,
                           (int)(c->im_which & 3), c->im_ccd_inv_norm, row, row + c->B, c->B);
    if (c->im_which & 4)
    {
        if (!c->d_im_fq && hipMalloc((void**)&c->d_im_fq, (size_t)c->B * c->N * c->N * sizeof(double)) != hipSuccess) {
            c->err = "invariant metrics: out of device memory for the fqc matrix";
            return MTIP_ENOMEM;
        }
        hipLaunchKernelGGL(k_metric_fqc, dim3((unsigned)c->N, (unsigned)((c->N + FQC_QP - 1) / FQC_QP)), dim3(256), 0, c->stream,
                           (const double2*)c->d_Bl, (const uint8_t*)c->d_im_zmask, (const double*)c->d_im_P, (const double*)c->d_im_refavg,
                           (const double*)c->d_im_refw, c->d_im_fq, c->N, c->L, c->B);
        hipLaunchKernelGGL(k_metric_fqc_fold, dim3((unsigned)((c->B * c->N + 63) / 64)), dim3(64), 0, c->stream, (const double*)c->d_im_fq,
                           row + 2 * (size_t)c->B, c->N, c->B);
    }
    return MTIP_OK;
}

extern "C" {

int mtip_set_invariant_metrics(mtip_ctx* c, uint32_t which, const uint8_t* zero_mask, const mtip_cdouble* II_reference, const double* qq,
                               const double* ccd_weights, const mtip_cdouble* ccd_reference, double ccd_norm, const double* fqc_P,
                               const double* fqc_reference_average, const double* fqc_reference_weights) {
    if (!c) return MTIP_EINVAL;
    (void)hipSetDevice(c->device);
    MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    free_invariant_metrics(c);
    if (which == 0) return MTIP_OK;
    if (which > 7 || !zero_mask || ((which & 1) && (!II_reference || !qq)) || ((which & 2) && (!ccd_weights || !ccd_reference || !(ccd_norm != 0.0))) ||
        ((which & 4) && (!fqc_P || !fqc_reference_average || !fqc_reference_weights))) {
        c->err = "invariant metrics: flags 1 (II_error) | 2 (ccd_diff) | 4 (fqc_error) with their tables, zero_mask not null";
        return MTIP_EINVAL;
    }
    const size_t NN = (size_t)c->N * c->N, LNN = (size_t)(c->L + 1) * NN;
#define IM_UP(dst, src, bytes)                                                        \
    do {                                                                              \
        MTIP_HIP_CHECK(c, hipMalloc((void**)&(dst), (bytes)));                        \
        MTIP_HIP_CHECK(c, mtip_copy(c, (dst), (src), (bytes), hipMemcpyHostToDevice)); \
    } while (0)
    IM_UP(c->d_im_zmask, zero_mask, LNN);
    if (which & 1) {
        IM_UP(c->d_im_IIref, II_reference, NN * sizeof(double2));
        IM_UP(c->d_im_qq, qq, NN * sizeof(double));
    }
    if (which & 2) {
        IM_UP(c->d_im_ccdT, ccd_weights, LNN * sizeof(double));
        IM_UP(c->d_im_ccdref, ccd_reference, NN * sizeof(double2));
        c->im_ccd_inv_norm = 1.0 / ccd_norm;
    }
    if (which & 4) {
        IM_UP(c->d_im_P, fqc_P, LNN * (c->L + 1) * sizeof(double));
        IM_UP(c->d_im_refavg, fqc_reference_average, NN * sizeof(double));
        IM_UP(c->d_im_refw, fqc_reference_weights, LNN * sizeof(double));
    }
#undef IM_UP
    MTIP_HIP_CHECK(c, hipMalloc((void**)&c->d_im_hist, (size_t)c->err_cap * c->B * (2 + c->N) * sizeof(double)));
    MTIP_HIP_CHECK(c, hipMemsetAsync(c->d_im_hist, 0, (size_t)c->err_cap * c->B * (2 + c->N) * sizeof(double), c->stream));
    c->im_which = which;
    return MTIP_OK;
}

/* per step of steps [first, first + n): II (n, B), ccd (n, B), fqc (n, B, Nq); a null pointer skips that metric */
int mtip_fetch_invariant_metrics(mtip_ctx* c, int64_t first, int64_t n, double* II, double* ccd, double* fqc) {
    if (!c) return MTIP_EINVAL;
    if (!c->im_which || first < 0 || n < 0 || first + n > c->n_steps_done) {
        c->err = "fetch_invariant_metrics: metrics not enabled or steps out of range";
        return MTIP_EINVAL;
    }
    (void)hipSetDevice(c->device);
    const size_t rowlen = (size_t)c->B * (2 + c->N);
    std::vector<double> rows((size_t)n * rowlen);
    if (n) MTIP_HIP_CHECK(c, mtip_copy(c, rows.data(), c->d_im_hist + (size_t)first * rowlen, rows.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int64_t s = 0; s < n; ++s) {
        const double* r = rows.data() + (size_t)s * rowlen;
        if (II) std::copy(r, r + c->B, II + (size_t)s * c->B);
        if (ccd) std::copy(r + c->B, r + 2 * c->B, ccd + (size_t)s * c->B);
        if (fqc) std::copy(r + 2 * c->B, r + rowlen, fqc + (size_t)s * c->B * c->N);
    }
    return MTIP_OK;
}

/* the reciprocal metric l2_projection_diff per FXS step: radial_w (Nq), theta_w (n_theta) of the integrator (shell N - 2 zeroed by the
   caller, see k_metric_rl2_shell); NULL weights switch it off */
int mtip_set_reciprocal_l2_metric(mtip_ctx* c, const double* radial_w, const double* theta_w) {
    if (!c) return MTIP_EINVAL;
    (void)hipSetDevice(c->device);
    for (void* p : {(void*)c->d_rl2_wr, (void*)c->d_rl2_wt, (void*)c->d_rl2_part, (void*)c->d_rl2_hist})
        if (p) (void)hipFree(p);
    c->d_rl2_wr = nullptr; c->d_rl2_wt = nullptr; c->d_rl2_part = nullptr; c->d_rl2_hist = nullptr;
    if (!radial_w || !theta_w) return MTIP_OK;
    MTIP_HIP_CHECK(c, hipMalloc((void**)&c->d_rl2_wr, c->N * sizeof(double)));
    MTIP_HIP_CHECK(c, hipMalloc((void**)&c->d_rl2_wt, c->nt * sizeof(double)));
    MTIP_HIP_CHECK(c, hipMalloc((void**)&c->d_rl2_part, (size_t)c->B * c->N * 2 * sizeof(double)));
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_rl2_wr, radial_w, c->N * sizeof(double), hipMemcpyHostToDevice));
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_rl2_wt, theta_w, c->nt * sizeof(double), hipMemcpyHostToDevice));
    MTIP_HIP_CHECK(c, hipMalloc((void**)&c->d_rl2_hist, (size_t)c->err_cap * c->B * sizeof(double)));
    MTIP_HIP_CHECK(c, hipMemsetAsync(c->d_rl2_hist, 0, (size_t)c->err_cap * c->B * sizeof(double), c->stream));
    return MTIP_OK;
}

/* steps [first, first + n): out (n, B) */
int mtip_fetch_reciprocal_l2_metric(mtip_ctx* c, int64_t first, int64_t n, double* out) {
    if (!c) return MTIP_EINVAL;
    if (!c->d_rl2_hist || !out || first < 0 || n < 0 || first + n > c->n_steps_done) {
        c->err = "fetch_reciprocal_l2_metric: metric not enabled or steps out of range";
        return MTIP_EINVAL;
    }
    (void)hipSetDevice(c->device);
    if (n) MTIP_HIP_CHECK(c, mtip_copy(c, out, c->d_rl2_hist + (size_t)first * c->B, (size_t)n * c->B * sizeof(double), hipMemcpyDeviceToHost));
    return MTIP_OK;
}

/* the enabled metrics of given intensity coefficients (n_batch, Nq, (L+1)^2): II (n_batch), ccd (n_batch), fqc (n_batch, Nq); NULL skips */
int mtip_op_invariant_metrics(mtip_ctx* c, const mtip_cdouble* Ilm, double* II, double* ccd, double* fqc) {
    if (!c) return MTIP_EINVAL;
    if (!c->im_which || !Ilm) {
        c->err = "op_invariant_metrics: metrics not enabled (mtip_set_invariant_metrics) or null coefficients";
        return MTIP_EINVAL;
    }
    (void)hipSetDevice(c->device);
    const size_t rowlen = (size_t)c->B * (2 + c->N);
    double* d_row = nullptr;
    MTIP_HIP_CHECK(c, hipMalloc((void**)&d_row, rowlen * sizeof(double)));
    hipError_t he = hipMemsetAsync(d_row, 0, rowlen * sizeof(double), c->stream);
    if (he == hipSuccess) he = mtip_copy(c, c->d_c[2], Ilm, (size_t)c->B * c->C * sizeof(double2), hipMemcpyHostToDevice);
    if (he != hipSuccess) {                                  // (the row buffer must not outlive a failed call)
        (void)hipFree(d_row);
        c->err = std::string("op_invariant_metrics: ") + hipGetErrorString(he);
        return MTIP_EHIP;
    }
    int rc = launch_invariant_metrics_row(c, c->d_c[2], d_row);
    std::vector<double> row(rowlen);
    if (rc == MTIP_OK && mtip_copy(c, row.data(), d_row, rowlen * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = MTIP_EHIP;
    (void)hipFree(d_row);
    if (rc != MTIP_OK) return rc;
    if (II) std::copy(row.begin(), row.begin() + c->B, II);
    if (ccd) std::copy(row.begin() + c->B, row.begin() + 2 * c->B, ccd);
    if (fqc) std::copy(row.begin() + 2 * c->B, row.end(), fqc);
    return MTIP_OK;
}

}  // extern "C"
