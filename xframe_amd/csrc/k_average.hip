// Grid arithmetic of the averaging worker (row f-1 of SURVEY section 8: xframe/projects/fxs/average.py:359-627, 721-727) between the
// transforms / SO(3) kernels: weighted moments and integrals, phase ramps, scaling, sums over aligned reconstructions, the PRTF
// shell statistics, normalisation.  The batch of reconstructions lives in device memory owned by the caller (the Python side
// keeps it in torch tensors); every entry point takes device OR host pointers (host arrays are staged through a temporary).
//   centre of mass            misk.py:295-312 (generate_calc_center)           -> mtip_op_grid_stats [0..3]
//   normed integrals          mathLibrary.py:1223-1237 (SphericalIntegrator)   -> mtip_op_grid_stats [4], [5]
//   normalisation factors     average.py:424-435, 721-727                      -> mtip_op_grid_stats [6..10], mtip_op_grid_combine
//   shift operator            fxs_Projections.py:1419-1444                     -> mtip_op_grid_phase_ramp
//   sums / scaling / conj     average.py:432-437, 526-536                      -> mtip_op_grid_combine
//   PRTF                      resolution_metrics.py:62-110                     -> mtip_op_prtf
#include "mtip_internal.h"

#define AV_THREADS 256
#define AV_NSTAT 12

// a caller's array as device memory: itself when it is device memory, else a temporary that is filled / copied back
struct DevView {
    mtip_ctx* c = nullptr;
    void* dev = nullptr;
    void* host = nullptr;
    size_t bytes = 0;
    bool temp = false, writeback = false;
    hipError_t err = hipSuccess;
    DevView(mtip_ctx* c_, const void* p, size_t n, bool read, bool write) : c(c_), bytes(n), writeback(write) {
        if (p == nullptr || n == 0) return;
        hipPointerAttribute_t at;
        const bool is_dev = hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeDevice;
        (void)hipGetLastError();                    // (an unregistered host pointer sets the sticky error on some runtimes)
        if (is_dev) {
            dev = const_cast<void*>(p);
            return;
        }
        temp = true;
        host = const_cast<void*>(p);
        err = hipMalloc(&dev, n);
        if (err == hipSuccess && read) err = mtip_copy(c, dev, p, n, hipMemcpyHostToDevice);
    }
    hipError_t finish() {                           // after the stream has been synchronised
        if (temp && writeback && err == hipSuccess) err = mtip_copy(c, host, dev, bytes, hipMemcpyDeviceToHost);
        return err;
    }
    ~DevView() {
        if (temp && dev) (void)hipFree(dev);
    }
};

// ---- moments, integrals and extrema of n grids (partials per workgroup, then a fixed-order reduction: reproducible) ----------
// per grid: [0] sum w Re, [1..3] sum w Re {x, y, z}, [4] sum w Re^2, [5] sum w (Re - Re ref)^2, [6] max Re, [7] min Re,
// [8], [9] sum of the entries "> 0" in numpy's lexicographic order on complex numbers (Re > 0, or Re == 0 and Im > 0), [10] their count
__global__ void __launch_bounds__(AV_THREADS) k_av_stats(const double2* __restrict__ g, const double2* __restrict__ ref,
                                                         const double* __restrict__ wr, const double* __restrict__ wt,
                                                         const double* __restrict__ rs, const double* __restrict__ cost,
                                                         int nt, int np, long long G, double* __restrict__ part) {
    __shared__ double red[AV_THREADS];
    const int b = blockIdx.y;
    const double2* gb = g + (size_t)b * G;
    double s[AV_NSTAT];
#pragma unroll
    for (int i = 0; i < AV_NSTAT; ++i) s[i] = 0.0;
    s[6] = -__builtin_huge_val();
    s[7] = __builtin_huge_val();
    const double dphi = 2.0 * 3.14159265358979323846 / np;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < G; i += (long long)gridDim.x * blockDim.x) {
        const int p = (int)(i % np), th = (int)((i / np) % nt), q = (int)(i / ((long long)np * nt));
        const double2 v = gb[i];
        const double w = wr[q] * wt[th];
        const double ct = cost[th], st = sqrt(fmax(0.0, 1.0 - ct * ct));
        double sp, cp;
        sincos(dphi * p, &sp, &cp);
        const double r = rs[q];
        const double wv = w * v.x;
        s[0] += wv;
        s[1] = fma(wv, r * st * cp, s[1]);
        s[2] = fma(wv, r * st * sp, s[2]);
        s[3] = fma(wv, r * ct, s[3]);
        s[4] = fma(wv, v.x, s[4]);
        if (ref != nullptr) {
            const double d = ref[i].x - v.x;
            s[5] = fma(w * d, d, s[5]);
        }
        s[6] = fmax(s[6], v.x);
        s[7] = fmin(s[7], v.x);
        if (v.x > 0.0 || (v.x == 0.0 && v.y > 0.0)) {
            s[8] += v.x;
            s[9] += v.y;
            s[10] += 1.0;
        }
    }
    double* out = part + ((size_t)b * gridDim.x + blockIdx.x) * AV_NSTAT;
    for (int k = 0; k < AV_NSTAT; ++k) {
        red[threadIdx.x] = s[k];
        __syncthreads();
        for (int o = AV_THREADS / 2; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) {
                const double a = red[threadIdx.x], c2 = red[threadIdx.x + o];
                red[threadIdx.x] = k == 6 ? fmax(a, c2) : (k == 7 ? fmin(a, c2) : a + c2);
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) out[k] = red[0];
        __syncthreads();
    }
}

__global__ void k_av_stats_finish(const double* __restrict__ part, int nblk, double* __restrict__ out) {
    const int b = blockIdx.x, k = threadIdx.x;
    if (k >= AV_NSTAT) return;
    double a = part[(size_t)b * nblk * AV_NSTAT + k];
    for (int i = 1; i < nblk; ++i) {
        const double v = part[((size_t)b * nblk + i) * AV_NSTAT + k];
        a = k == 6 ? fmax(a, v) : (k == 7 ? fmin(a, v) : a + v);
    }
    out[(size_t)b * AV_NSTAT + k] = a;
}

// ---- grids *= exp(-i s k . c_b), k = q (sin theta cos phi, sin theta sin phi, cos theta) (fxs_Projections.py:1436-1443) -------
__global__ void __launch_bounds__(AV_THREADS) k_av_phase(double2* __restrict__ g, const double* __restrict__ cc, double sgn,
                                                         const double* __restrict__ qs, const double* __restrict__ cost, int nt,
                                                         int np, long long G) {
    const int b = blockIdx.y;
    const double cx = cc[3 * b], cy = cc[3 * b + 1], cz = cc[3 * b + 2];
    const double dphi = 2.0 * 3.14159265358979323846 / np;
    double2* gb = g + (size_t)b * G;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < G; i += (long long)gridDim.x * blockDim.x) {
        const int p = (int)(i % np), th = (int)((i / np) % nt), q = (int)(i / ((long long)np * nt));
        const double ct = cost[th], st = sqrt(fmax(0.0, 1.0 - ct * ct));
        double sp, cp;
        sincos(dphi * p, &sp, &cp);
        const double kc = qs[q] * (st * cp * cx + st * sp * cy + ct * cz);
        double sn, cs;
        sincos(-sgn * kc, &sn, &cs);
        gb[i] = cmul(gb[i], make_double2(cs, sn));
    }
}

// ---- elementwise / over-the-stack combinations -----------------------------------------------------------------------------------
// 0 conj: dst[b] = conj(a[b]);  1 scale: dst[b] = a[b] * s[b] (complex per grid);  2 sum: dst = sum_b a[b];
// 3 abs2 sum: dst = sum_b |a[b]|^2 (as complex);  4 affine: dst[b] = (a[b] - s[0]) * s[1] (complex scalars)
enum { AV_CONJ = 0, AV_SCALE = 1, AV_SUM = 2, AV_ABS2SUM = 3, AV_AFFINE = 4 };
__global__ void __launch_bounds__(AV_THREADS) k_av_combine(int op, double2* dst, const double2* a,      // (dst may be a)
                                                           const double2* __restrict__ s, int n, long long G) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < G; i += (long long)gridDim.x * blockDim.x) {
        if (op == AV_SUM || op == AV_ABS2SUM) {
            double2 acc = make_double2(0.0, 0.0);
            for (int b = 0; b < n; ++b) {               // fixed order over the stack
                const double2 v = a[(size_t)b * G + i];
                if (op == AV_SUM) acc = cadd(acc, v);
                else acc.x += cabs2(v);
            }
            dst[i] = acc;
        } else {
            for (int b = 0; b < n; ++b) {
                const double2 v = a[(size_t)b * G + i];
                double2 r;
                if (op == AV_CONJ) r = make_double2(v.x, -v.y);
                else if (op == AV_SCALE) r = cmul(v, s[b]);
                else r = cmul(csub(v, s[0]), s[1]);
                dst[(size_t)b * G + i] = r;
            }
        }
    }
}

// ---- PRTF (resolution_metrics.py:62-78): nd = sqrt(a1 conj(a2) / (b1 conj(b2))) with b = sqrt(Re I) where both b are non-zero,
//      0 where a b vanishes and both a do not, 1 elsewhere; per shell the mean (complex) and the standard deviation over the sphere
__device__ __forceinline__ double2 csqrt_principal(double2 z) {
    const double m = sqrt(sqrt(cabs2(z)));
    if (m == 0.0) return make_double2(0.0, 0.0);
    const double ph = 0.5 * atan2(z.y, z.x);
    double sn, cs;
    sincos(ph, &sn, &cs);
    return make_double2(m * cs, m * sn);
}
__device__ __forceinline__ double2 prtf_point(double2 a1, double2 a2, double i1, double i2) {
    const double b1 = sqrt(i1), b2 = sqrt(i2);
    const bool nz = (b1 != 0.0) && (b2 != 0.0);
    double2 nd = make_double2(1.0, 0.0);
    if (nz) {
        const double den = b1 * b2;
        const double2 num = cmulc(a1, a2);
        nd = make_double2(num.x / den, num.y / den);
    } else if ((a1.x != 0.0 || a1.y != 0.0) && (a2.x != 0.0 || a2.y != 0.0)) {
        nd = make_double2(0.0, 0.0);
    }
    return csqrt_principal(nd);
}
__global__ void __launch_bounds__(AV_THREADS) k_av_prtf(const double2* __restrict__ a1, const double2* __restrict__ a2,
                                                        const double2* __restrict__ I1, const double2* __restrict__ I2, int npts,
                                                        double2* __restrict__ mean, double* __restrict__ sd) {
    __shared__ double rx[AV_THREADS], ry[AV_THREADS];
    const int q = blockIdx.x;
    const size_t o = (size_t)q * npts;
    double sx = 0.0, sy = 0.0;
    for (int i = threadIdx.x; i < npts; i += blockDim.x) {
        const double2 v = prtf_point(a1[o + i], a2[o + i], I1[o + i].x, I2[o + i].x);
        sx += v.x;
        sy += v.y;
    }
    rx[threadIdx.x] = sx;
    ry[threadIdx.x] = sy;
    __syncthreads();
    for (int k = AV_THREADS / 2; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) {
            rx[threadIdx.x] += rx[threadIdx.x + k];
            ry[threadIdx.x] += ry[threadIdx.x + k];
        }
        __syncthreads();
    }
    const double mx = rx[0] / npts, my = ry[0] / npts;
    __syncthreads();
    double s2 = 0.0;
    for (int i = threadIdx.x; i < npts; i += blockDim.x) {
        const double2 v = prtf_point(a1[o + i], a2[o + i], I1[o + i].x, I2[o + i].x);
        const double dx = v.x - mx, dy = v.y - my;
        s2 += dx * dx + dy * dy;
    }
    rx[threadIdx.x] = s2;
    __syncthreads();
    for (int k = AV_THREADS / 2; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) rx[threadIdx.x] += rx[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        mean[q] = make_double2(mx, my);
        sd[q] = sqrt(rx[0] / npts);
    }
}

// ----------------------------------------------------------------------------------------------------------------------------------
#define AV_FAIL(c, code, msg)   \
    do {                        \
        (c)->err = (msg);       \
        return (code);          \
    } while (0)
#define AV_HIP(c, call)                                                              \
    do {                                                                             \
        hipError_t e__ = (call);                                                     \
        if (e__ != hipSuccess) {                                                     \
            (c)->err = std::string(#call) + ": " + hipGetErrorString(e__);           \
            return MTIP_EHIP;                                                        \
        }                                                                            \
    } while (0)

static int av_blocks(const mtip_ctx* c) { return std::max(1, std::min(div_up((long long)c->G, AV_THREADS * 4), 4 * c->n_cu)); }

extern "C" {

int mtip_op_grid_stats(mtip_ctx* c, const mtip_cdouble* grids, int n, const mtip_cdouble* ref, const double* radial_w,
                       const double* theta_w, double* out) {
    if (!c) return MTIP_EINVAL;
    if (!grids || n <= 0 || !radial_w || !theta_w || !out) AV_FAIL(c, MTIP_EINVAL, "grid_stats: null buffer / empty stack");
    if (!c->have_radial || !c->have_angular) AV_FAIL(c, MTIP_ESTATE, "grid_stats: mtip_set_radial_grid / mtip_set_angular_grid have not been called");
    (void)hipSetDevice(c->device);
    AV_HIP(c, hipStreamSynchronize(c->stream));
    const int nblk = av_blocks(c);
    DevView vg(c, grids, (size_t)n * c->G * sizeof(double2), true, false);
    DevView vr(c, ref, c->G * sizeof(double2), true, false);
    DevView vw(c, radial_w, c->N * sizeof(double), true, false);
    DevView vt(c, theta_w, c->nt * sizeof(double), true, false);
    double* d_part = nullptr;
    AV_HIP(c, vg.err);
    AV_HIP(c, vr.err);
    AV_HIP(c, vw.err);
    AV_HIP(c, vt.err);
    AV_HIP(c, hipMalloc((void**)&d_part, ((size_t)n * nblk + n) * AV_NSTAT * sizeof(double)));
    double* d_out = d_part + (size_t)n * nblk * AV_NSTAT;
    hipLaunchKernelGGL(k_av_stats, dim3((unsigned)nblk, (unsigned)n), dim3(AV_THREADS), 0, c->stream, (const double2*)vg.dev,
                       (const double2*)vr.dev, (const double*)vw.dev, (const double*)vt.dev, (const double*)c->d_r,
                       (const double*)c->d_cost, c->nt, c->np, (long long)c->G, d_part);
    hipLaunchKernelGGL(k_av_stats_finish, dim3((unsigned)n), dim3(64), 0, c->stream, (const double*)d_part, nblk, d_out);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = mtip_copy(c, out, d_out, (size_t)n * AV_NSTAT * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(d_part);
    AV_HIP(c, e);
    AV_HIP(c, hipGetLastError());
    return MTIP_OK;
}

int mtip_op_grid_phase_ramp(mtip_ctx* c, mtip_cdouble* grids, int n, const double* centers_cartesian, double sign) {
    if (!c) return MTIP_EINVAL;
    if (!grids || n <= 0 || !centers_cartesian) AV_FAIL(c, MTIP_EINVAL, "grid_phase_ramp: null buffer / empty stack");
    if (!c->have_radial || !c->have_angular) AV_FAIL(c, MTIP_ESTATE, "grid_phase_ramp: mtip_set_radial_grid / mtip_set_angular_grid have not been called");
    (void)hipSetDevice(c->device);
    AV_HIP(c, hipStreamSynchronize(c->stream));
    DevView vg(c, grids, (size_t)n * c->G * sizeof(double2), true, true);
    DevView vc(c, centers_cartesian, (size_t)n * 3 * sizeof(double), true, false);
    AV_HIP(c, vg.err);
    AV_HIP(c, vc.err);
    hipLaunchKernelGGL(k_av_phase, dim3((unsigned)av_blocks(c), (unsigned)n), dim3(AV_THREADS), 0, c->stream, (double2*)vg.dev,
                       (const double*)vc.dev, sign, (const double*)c->d_q, (const double*)c->d_cost, c->nt, c->np, (long long)c->G);
    AV_HIP(c, hipStreamSynchronize(c->stream));
    AV_HIP(c, vg.finish());
    AV_HIP(c, hipGetLastError());
    return MTIP_OK;
}

int mtip_op_grid_combine(mtip_ctx* c, int op, mtip_cdouble* dst, const mtip_cdouble* a, int n, const mtip_cdouble* scalars) {
    if (!c) return MTIP_EINVAL;
    if (!dst || !a || n <= 0 || op < AV_CONJ || op > AV_AFFINE) AV_FAIL(c, MTIP_EINVAL, "grid_combine: null buffer / empty stack / unknown op");
    if ((op == AV_SCALE || op == AV_AFFINE) && !scalars) AV_FAIL(c, MTIP_EINVAL, "grid_combine: this op needs scalars");
    (void)hipSetDevice(c->device);
    AV_HIP(c, hipStreamSynchronize(c->stream));
    const bool reduce = op == AV_SUM || op == AV_ABS2SUM;
    const size_t gb = c->G * sizeof(double2);
    DevView va(c, a, (size_t)n * gb, true, false);
    DevView vd(c, dst, reduce ? gb : (size_t)n * gb, dst == a, true);
    DevView vs(c, scalars, (op == AV_SCALE ? (size_t)n : (size_t)2) * sizeof(double2), true, false);
    AV_HIP(c, va.err);
    AV_HIP(c, vd.err);
    AV_HIP(c, vs.err);
    // (dst == a on device memory: every point is read before it is written by the same thread)
    hipLaunchKernelGGL(k_av_combine, dim3((unsigned)av_blocks(c)), dim3(AV_THREADS), 0, c->stream, op, (double2*)vd.dev,
                       (const double2*)va.dev, (const double2*)vs.dev, n, (long long)c->G);
    AV_HIP(c, hipStreamSynchronize(c->stream));
    AV_HIP(c, vd.finish());
    AV_HIP(c, hipGetLastError());
    return MTIP_OK;
}

int mtip_op_prtf(mtip_ctx* c, const mtip_cdouble* a1, const mtip_cdouble* a2, const mtip_cdouble* I1, const mtip_cdouble* I2,
                 mtip_cdouble* mean, double* std_dev) {
    if (!c) return MTIP_EINVAL;
    if (!a1 || !a2 || !I1 || !I2 || !mean || !std_dev) AV_FAIL(c, MTIP_EINVAL, "prtf: null buffer");
    (void)hipSetDevice(c->device);
    AV_HIP(c, hipStreamSynchronize(c->stream));
    const size_t gb = c->G * sizeof(double2);
    DevView v1(c, a1, gb, true, false), v2(c, a2, gb, true, false), v3(c, I1, gb, true, false), v4(c, I2, gb, true, false);
    AV_HIP(c, v1.err);
    AV_HIP(c, v2.err);
    AV_HIP(c, v3.err);
    AV_HIP(c, v4.err);
    double* d_out = nullptr;
    AV_HIP(c, hipMalloc((void**)&d_out, (size_t)c->N * 3 * sizeof(double)));
    hipLaunchKernelGGL(k_av_prtf, dim3((unsigned)c->N), dim3(AV_THREADS), 0, c->stream, (const double2*)v1.dev, (const double2*)v2.dev,
                       (const double2*)v3.dev, (const double2*)v4.dev, c->nt * c->np, reinterpret_cast<double2*>(d_out), d_out + 2 * (size_t)c->N);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = mtip_copy(c, mean, d_out, (size_t)c->N * sizeof(double2), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = mtip_copy(c, std_dev, d_out + 2 * (size_t)c->N, (size_t)c->N * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    AV_HIP(c, e);
    AV_HIP(c, hipGetLastError());
    return MTIP_OK;
}

}  // extern "C"
