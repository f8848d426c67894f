// Rotational alignment of reconstructions: the two operators of
// xframe/projects/fxs/average.py that the reference delegates to the third-party pysofft
//   find_rotation  average.py:920-947 -> soft.calc_mean_C (soft_plugin.py:82-99):  C(R) = <f, g o R> on the (2 bw)^3 Euler grid
//   rotate         average.py:948-960 -> soft.rotate_coeff (soft_plugin.py:64-79):  f_lm -> sum_n D^l_mn f_ln
// on 'direct' harmonic coefficients (Nq, (L+1)^2), index l(l+1)+m.  Conventions (ZYZ, D^l_mn = e^{-i m alpha} d^l_mn(beta)
// e^{-i n gamma}, alpha_j = gamma_j = 2 pi j / 2bw, beta_k = pi (2k+1) / 4bw, bw = L + 1) are those of oracle/alignment.py;
// pysofft itself is not available, see the parity note there.
//
//   T^l_mn   = mean over shells r_lo <= r < r_hi of conj(ref_lm(r)) sig_ln(r)                        (k_so3_T)
//   S_b(m,n) = sum_l T^l_mn d^l_mn(beta_b)                                                          (k_so3_S)
//   P_b(m,k) = sum_n S_b(m,n) e^{-i n gamma_k},   C(j,b,k) = Re sum_m e^{-i m alpha_j} P_b(m,k)      (k_so3_P, k_so3_C)
// The Wigner table d^l_mn(beta_b) is computed once on the host (eigen-decomposition of J_y) and uploaded.
#include "mtip_internal.h"

// one workgroup per (order, restart): T^l (2l+1 x 2l+1), row m, column n
__global__ void __launch_bounds__(256) k_so3_T(const double2* __restrict__ ref, const double2* __restrict__ sig,
                                               double2* __restrict__ T, int N, int nlm, int ntab, int r_lo, int r_hi) {
    const int l = blockIdx.x, b = blockIdx.y, n = 2 * l + 1;
    const int off = l * (4 * l * l - 1) / 3;                       // sum_{l' < l} (2l'+1)^2
    const double inv = 1.0 / (double)(r_hi - r_lo);
    for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
        const int m = e / n, nn = e - m * n;
        double2 acc = make_double2(0.0, 0.0);
        for (int r = r_lo; r < r_hi; ++r) {
            const double2 a = ref[(size_t)r * nlm + l * l + m];
            const double2 c = sig[((size_t)b * N + r) * nlm + l * l + nn];
            acc.x += a.x * c.x + a.y * c.y;                         // conj(a) * c
            acc.y += a.x * c.y - a.y * c.x;
        }
        T[(size_t)b * ntab + off + e] = cscale(acc, inv);
    }
}

// S_b(m, n), m, n = -L..L stored at (m + L) * M + (n + L), M = 2L + 1
__global__ void __launch_bounds__(256) k_so3_S(const double2* __restrict__ T, const double* __restrict__ dtab,
                                               double2* __restrict__ S, int L, int ntab) {
    const int bi = blockIdx.x, b = blockIdx.y, M = 2 * L + 1, nb = gridDim.x;
    for (int e = threadIdx.x; e < M * M; e += blockDim.x) {
        const int m = e / M - L, n = e % M - L;
        const int l0 = max(abs(m), abs(n));
        double2 acc = make_double2(0.0, 0.0);
        for (int l = l0; l <= L; ++l) {
            const int off = l * (4 * l * l - 1) / 3 + (m + l) * (2 * l + 1) + (n + l);
            const double d = dtab[(size_t)bi * ntab + off];
            const double2 t = T[(size_t)b * ntab + off];
            acc.x = fma(d, t.x, acc.x);
            acc.y = fma(d, t.y, acc.y);
        }
        S[((size_t)b * nb + bi) * M * M + e] = acc;
    }
}

// P_b(m, k) = sum_n S_b(m, n) e^{-2 pi i n k / nb}
__global__ void __launch_bounds__(256) k_so3_P(const double2* __restrict__ S, const double2* __restrict__ tw,
                                               double2* __restrict__ P, int L) {
    const int bi = blockIdx.x, b = blockIdx.y, M = 2 * L + 1, nb = gridDim.x;
    const double2* Sb = S + ((size_t)b * nb + bi) * M * M;
    for (int e = threadIdx.x; e < M * nb; e += blockDim.x) {
        const int mi = e / nb, k = e - mi * nb;
        double2 acc = make_double2(0.0, 0.0);
        for (int ni = 0; ni < M; ++ni) {
            int idx = ((ni - L) * k) % nb;
            if (idx < 0) idx += nb;
            acc = cadd(acc, cmul(Sb[mi * M + ni], tw[idx]));
        }
        P[((size_t)b * nb + bi) * M * nb + e] = acc;
    }
}

// C(j, b, k) = Re sum_m e^{-2 pi i m j / nb} P_b(m, k), output indexed [restart][alpha j][beta b][gamma k]
__global__ void __launch_bounds__(256) k_so3_C(const double2* __restrict__ P, const double2* __restrict__ tw,
                                               double* __restrict__ C, int L) {
    const int bi = blockIdx.x, b = blockIdx.y, M = 2 * L + 1, nb = gridDim.x;
    const double2* Pb = P + ((size_t)b * nb + bi) * M * nb;
    for (int e = threadIdx.x; e < nb * nb; e += blockDim.x) {
        const int j = e / nb, k = e - j * nb;
        double acc = 0.0;
        for (int mi = 0; mi < M; ++mi) {
            int idx = ((mi - L) * j) % nb;
            if (idx < 0) idx += nb;
            const double2 p = Pb[mi * nb + k], w = tw[idx];
            acc += p.x * w.x - p.y * w.y;
        }
        C[(((size_t)b * nb + j) * nb + bi) * nb + k] = acc;
    }
}

// out_lm(r) = sum_n D^l_mn c_ln(r); D per restart in the table layout
__global__ void __launch_bounds__(256) k_rotate_coeff(const double2* __restrict__ coeff, const double2* __restrict__ D,
                                                      double2* __restrict__ out, int N, int L, int nlm, int ntab) {
    const int r = blockIdx.x, b = blockIdx.y;
    const double2* c = coeff + ((size_t)b * N + r) * nlm;
    for (int e = threadIdx.x; e < nlm; e += blockDim.x) {
        const int l = isqrt_lm(e), mi = e - l * l, n = 2 * l + 1;
        const double2* Dl = D + (size_t)b * ntab + l * (4 * l * l - 1) / 3 + mi * n;
        double2 acc = make_double2(0.0, 0.0);
        for (int ni = 0; ni < n; ++ni) acc = cadd(acc, cmul(Dl[ni], c[l * l + ni]));
        out[((size_t)b * N + r) * nlm + e] = acc;
    }
}

// D^l_mn = e^{-i m alpha} d^l_mn(beta_bi) e^{-i n gamma} from the device's Wigner table, for rotations whose beta is a grid sample
// (what find_rotation hands out): one workgroup per (order, restart)
__global__ void __launch_bounds__(256) k_so3_build_D(const double* __restrict__ dtab, const int* __restrict__ beta_idx,
                                                     const double* __restrict__ alpha, const double* __restrict__ gamma,
                                                     double2* __restrict__ D, int ntab) {
    const int l = blockIdx.x, b = blockIdx.y, n = 2 * l + 1;
    const int off = l * (4 * l * l - 1) / 3;
    const double a = alpha[b], g = gamma[b];
    const double* d = dtab + (size_t)beta_idx[b] * ntab + off;
    for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
        const int m = e / n - l, nn = e % n - l;
        double sn, cs;
        sincos(-(m * a + nn * g), &sn, &cs);
        D[(size_t)b * ntab + off + e] = make_double2(d[e] * cs, d[e] * sn);
    }
}

// arg-max of the correlation in the order the reference reads it in (average.py:936-940: [beta, alpha, gamma], tabulated at
// the angles whose flip alpha -> 2 pi - alpha, gamma -> 2 pi - gamma is the aligning rotation -- oracle/alignment.py
// mean_C_layout): the key of element C[j][bi][k] is bi nb^2 + ((-j) mod nb) nb + ((-k) mod nb); the first maximum in that order
// wins, as numpy's argmax does -- and like numpy's, a NaN counts as the maximum (the first NaN in reading order is returned).
// One workgroup per restart.
__device__ __forceinline__ bool so3_better(double v2, long long k2, double v, long long k) {
    if (v2 != v2) return (v == v) || k2 < k;
    return (v == v) && (v2 > v || (v2 == v && k2 < k));
}
__global__ void __launch_bounds__(1024) k_so3_argmax(const double* __restrict__ C, int nb, long long* __restrict__ arg, double* __restrict__ vmax) {
    __shared__ double s_v[16];
    __shared__ long long s_k[16];
    const int b = blockIdx.x;
    const double* Cb = C + (size_t)b * nb * nb * nb;
    double best = -HUGE_VAL;
    long long key = 0x7fffffffffffffffLL;
    for (int e = threadIdx.x; e < nb * nb * nb; e += blockDim.x) {
        const int j = e / (nb * nb), bi = (e / nb) % nb, k = e % nb;
        const long long kk = (long long)bi * nb * nb + (long long)((nb - j) % nb) * nb + (nb - k) % nb;
        const double v = Cb[e];
        if (so3_better(v, kk, best, key)) {
            best = v;
            key = kk;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double v2 = __shfl_xor(best, o, 64);
        const long long k2 = __shfl_xor(key, o, 64);
        if (so3_better(v2, k2, best, key)) {
            best = v2;
            key = k2;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        s_v[threadIdx.x >> 6] = best;
        s_k[threadIdx.x >> 6] = key;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int wv = 1; wv < (int)(blockDim.x >> 6); ++wv)
            if (so3_better(s_v[wv], s_k[wv], best, key)) {
                best = s_v[wv];
                key = s_k[wv];
            }
        arg[b] = key;
        vmax[b] = best;
    }
}

static int so3_ntab(int L) { return (L + 1) * (2 * L + 1) * (2 * L + 3) / 3; }

// enqueue + wait: C of (ref, sig) into d_so3_C
static int so3_correlate(mtip_ctx* c, const mtip_cdouble* ref, const mtip_cdouble* sig, int r_lo, int r_hi) {
    const int nb = 2 * c->so3_bw, ntab = so3_ntab(c->L);
    MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_c[0], ref, c->C * sizeof(double2), hipMemcpyHostToDevice));
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_c[1], sig, (size_t)c->B * c->C * sizeof(double2), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_so3_T, dim3((unsigned)(c->L + 1), (unsigned)c->B), dim3(256), 0, c->stream, (const double2*)c->d_c[0],
                       (const double2*)c->d_c[1], c->d_so3_T, c->N, c->nlm, ntab, r_lo, r_hi);
    hipLaunchKernelGGL(k_so3_S, dim3((unsigned)nb, (unsigned)c->B), dim3(256), 0, c->stream, (const double2*)c->d_so3_T,
                       (const double*)c->d_so3_d, c->d_so3_S, c->L, ntab);
    hipLaunchKernelGGL(k_so3_P, dim3((unsigned)nb, (unsigned)c->B), dim3(256), 0, c->stream, (const double2*)c->d_so3_S,
                       (const double2*)c->d_so3_tw, c->d_so3_P, c->L);
    hipLaunchKernelGGL(k_so3_C, dim3((unsigned)nb, (unsigned)c->B), dim3(256), 0, c->stream, (const double2*)c->d_so3_P,
                       (const double2*)c->d_so3_tw, c->d_so3_C, c->L);
    MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        c->err = std::string("so3 correlation: ") + hipGetErrorString(e);
        return MTIP_EHIP;
    }
    return MTIP_OK;
}

extern "C" {

int mtip_set_so3_tables(mtip_ctx* c, int bw, const double* d_table) {
    if (!c) return MTIP_EINVAL;
    if (bw != c->L + 1 || !d_table) {
        c->err = "SO(3) tables: bandwidth must be L + 1 and the table non-null";
        return MTIP_EINVAL;
    }
    (void)hipSetDevice(c->device);
    const int nb = 2 * bw, ntab = so3_ntab(c->L), M = 2 * c->L + 1;
    MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    for (void* p : {(void*)c->d_so3_d, (void*)c->d_so3_tw, (void*)c->d_so3_T, (void*)c->d_so3_S, (void*)c->d_so3_P, (void*)c->d_so3_C, (void*)c->d_so3_D})
        if (p) (void)hipFree(p);
    c->d_so3_d = nullptr; c->d_so3_tw = nullptr; c->d_so3_T = nullptr; c->d_so3_S = nullptr; c->d_so3_P = nullptr; c->d_so3_C = nullptr; c->d_so3_D = nullptr;
    MTIP_HIP_CHECK(c, hipMalloc((void**)&c->d_so3_d, (size_t)nb * ntab * sizeof(double)));
    MTIP_HIP_CHECK(c, hipMalloc((void**)&c->d_so3_tw, (size_t)nb * sizeof(double2)));
    MTIP_HIP_CHECK(c, hipMalloc((void**)&c->d_so3_T, (size_t)c->B * ntab * sizeof(double2)));
    MTIP_HIP_CHECK(c, hipMalloc((void**)&c->d_so3_D, (size_t)c->B * ntab * sizeof(double2)));
    MTIP_HIP_CHECK(c, hipMalloc((void**)&c->d_so3_S, (size_t)c->B * nb * M * M * sizeof(double2)));
    MTIP_HIP_CHECK(c, hipMalloc((void**)&c->d_so3_P, (size_t)c->B * nb * M * nb * sizeof(double2)));
    MTIP_HIP_CHECK(c, hipMalloc((void**)&c->d_so3_C, (size_t)c->B * nb * nb * nb * sizeof(double)));
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_so3_d, d_table, (size_t)nb * ntab * sizeof(double), hipMemcpyHostToDevice));
    std::vector<double2> tw(nb);
    const double pi = 3.14159265358979323846;
    for (int k = 0; k < nb; ++k) tw[k] = make_double2(std::cos(2 * pi * k / nb), -std::sin(2 * pi * k / nb));
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_so3_tw, tw.data(), nb * sizeof(double2), hipMemcpyHostToDevice));
    c->so3_bw = bw;
    return MTIP_OK;
}

int mtip_op_so3_correlation(mtip_ctx* c, const mtip_cdouble* ref, const mtip_cdouble* sig, int r_lo, int r_hi, double* C) {
    if (!c) return MTIP_EINVAL;
    if (c->so3_bw == 0) {
        c->err = "mtip_set_so3_tables has not been called";
        return MTIP_ESTATE;
    }
    if (!ref || !sig || !C || r_lo < 0 || r_hi > c->N || r_lo >= r_hi) {
        c->err = "so3_correlation: null buffer or bad shell range";
        return MTIP_EINVAL;
    }
    (void)hipSetDevice(c->device);
    const int nb = 2 * c->so3_bw;
    const int rc = so3_correlate(c, ref, sig, r_lo, r_hi);
    if (rc != MTIP_OK) return rc;
    MTIP_HIP_CHECK(c, mtip_copy(c, C, c->d_so3_C, (size_t)c->B * nb * nb * nb * sizeof(double), hipMemcpyDeviceToHost));
    return MTIP_OK;
}

/* the correlation and, per restart, its arg-max in the reference's reading order (see k_so3_argmax): arg (B) int64 =
 * i_beta nb^2 + i_alpha nb + i_gamma of average.py:936-938's argmax, vmax (B) the maximum; C (B, nb, nb, nb) is copied out too
 * unless null.  All buffers host or device memory. */
int mtip_op_so3_find_rotation(mtip_ctx* c, const mtip_cdouble* ref, const mtip_cdouble* sig, int r_lo, int r_hi, int64_t* arg,
                              double* vmax, double* C) {
    if (!c) return MTIP_EINVAL;
    if (c->so3_bw == 0) {
        c->err = "mtip_set_so3_tables has not been called";
        return MTIP_ESTATE;
    }
    if (!ref || !sig || !arg || !vmax || r_lo < 0 || r_hi > c->N || r_lo >= r_hi) {
        c->err = "so3_find_rotation: null buffer or bad shell range";
        return MTIP_EINVAL;
    }
    (void)hipSetDevice(c->device);
    const int nb = 2 * c->so3_bw;
    const int rc = so3_correlate(c, ref, sig, r_lo, r_hi);
    if (rc != MTIP_OK) return rc;
    // (d_so3_P is free again: the first words of it take the B results)
    long long* d_arg = reinterpret_cast<long long*>(c->d_so3_P);
    double* d_max = reinterpret_cast<double*>(c->d_so3_P) + c->B;
    hipLaunchKernelGGL(k_so3_argmax, dim3((unsigned)c->B), dim3(1024), 0, c->stream, (const double*)c->d_so3_C, nb, d_arg, d_max);
    MTIP_HIP_CHECK(c, mtip_copy(c, arg, d_arg, (size_t)c->B * sizeof(long long), hipMemcpyDeviceToHost));
    MTIP_HIP_CHECK(c, mtip_copy(c, vmax, d_max, (size_t)c->B * sizeof(double), hipMemcpyDeviceToHost));
    if (C) MTIP_HIP_CHECK(c, mtip_copy(c, C, c->d_so3_C, (size_t)c->B * nb * nb * nb * sizeof(double), hipMemcpyDeviceToHost));
    return MTIP_OK;
}

/* rotate_coefficients for Euler angles whose beta is sample beta_index[b] of the SO(3) grid (what find_rotation hands out):
 * D^l_mn = e^{-i m alpha} d^l_mn(beta) e^{-i n gamma} is built on the device from the Wigner table */
int mtip_op_rotate_coefficients_grid(mtip_ctx* c, const mtip_cdouble* coeff, const int32_t* beta_index, const double* alpha,
                                     const double* gamma, mtip_cdouble* out) {
    if (!c) return MTIP_EINVAL;
    if (c->so3_bw == 0) {
        c->err = "mtip_set_so3_tables has not been called";
        return MTIP_ESTATE;
    }
    if (!coeff || !beta_index || !alpha || !gamma || !out) {
        c->err = "rotate_coefficients_grid: null buffer";
        return MTIP_EINVAL;
    }
    for (int b = 0; b < c->B; ++b)
        if (beta_index[b] < 0 || beta_index[b] >= 2 * c->so3_bw) {
            c->err = "rotate_coefficients_grid: beta index outside the SO(3) grid";
            return MTIP_EINVAL;
        }
    (void)hipSetDevice(c->device);
    const int ntab = so3_ntab(c->L);
    MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_c[0], coeff, (size_t)c->B * c->C * sizeof(double2), hipMemcpyHostToDevice));
    // (angles and indices are host arrays of B entries: staged in the head of d_so3_S)
    double* d_al = reinterpret_cast<double*>(c->d_so3_S);
    double* d_ga = d_al + c->B;
    int* d_bi = reinterpret_cast<int*>(d_ga + c->B);
    MTIP_HIP_CHECK(c, mtip_copy(c, d_al, alpha, (size_t)c->B * sizeof(double), hipMemcpyHostToDevice));
    MTIP_HIP_CHECK(c, mtip_copy(c, d_ga, gamma, (size_t)c->B * sizeof(double), hipMemcpyHostToDevice));
    MTIP_HIP_CHECK(c, mtip_copy(c, d_bi, beta_index, (size_t)c->B * sizeof(int), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_so3_build_D, dim3((unsigned)(c->L + 1), (unsigned)c->B), dim3(256), 0, c->stream, (const double*)c->d_so3_d,
                       (const int*)d_bi, (const double*)d_al, (const double*)d_ga, c->d_so3_D, ntab);
    hipLaunchKernelGGL(k_rotate_coeff, dim3((unsigned)c->N, (unsigned)c->B), dim3(256), 0, c->stream, (const double2*)c->d_c[0],
                       (const double2*)c->d_so3_D, c->d_c[1], c->N, c->L, c->nlm, ntab);
    MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    MTIP_HIP_CHECK(c, mtip_copy(c, out, c->d_c[1], (size_t)c->B * c->C * sizeof(double2), hipMemcpyDeviceToHost));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        c->err = std::string("mtip_op_rotate_coefficients_grid: ") + hipGetErrorString(e);
        return MTIP_EHIP;
    }
    return MTIP_OK;
}

int mtip_op_rotate_coefficients(mtip_ctx* c, const mtip_cdouble* coeff, const mtip_cdouble* D, mtip_cdouble* out) {
    if (!c) return MTIP_EINVAL;
    if (c->so3_bw == 0) {
        c->err = "mtip_set_so3_tables has not been called";
        return MTIP_ESTATE;
    }
    if (!coeff || !D || !out) {
        c->err = "rotate_coefficients: null buffer";
        return MTIP_EINVAL;
    }
    (void)hipSetDevice(c->device);
    const int ntab = so3_ntab(c->L);
    MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_c[0], coeff, (size_t)c->B * c->C * sizeof(double2), hipMemcpyHostToDevice));
    MTIP_HIP_CHECK(c, mtip_copy(c, c->d_so3_D, D, (size_t)c->B * ntab * sizeof(double2), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_rotate_coeff, dim3((unsigned)c->N, (unsigned)c->B), dim3(256), 0, c->stream, (const double2*)c->d_c[0],
                       (const double2*)c->d_so3_D, c->d_c[1], c->N, c->L, c->nlm, ntab);
    MTIP_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    MTIP_HIP_CHECK(c, mtip_copy(c, out, c->d_c[1], (size_t)c->B * c->C * sizeof(double2), hipMemcpyDeviceToHost));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        c->err = std::string("mtip_op_rotate_coefficients: ") + hipGetErrorString(e);
        return MTIP_EHIP;
    }
    return MTIP_OK;
}

}  // extern "C"
