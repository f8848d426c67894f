// The 2-D (polar) variant of the hot path, operator level: the circular harmonic transforms, the polar
// Hankel transform, the Fourier pair built from them and the 2-D reciprocal projection of xFrame's fxs project
//   circularHarmonicTransform_{complex,real}_{forward,inverse}   xframe/library/mathLibrary.py:469-496
//   generate_polar_ht (midpoint branch)                          projects/fxs/projectLibrary/hankel_transforms.py:602-640
//   generate_ft, dimensions = 2                                  projectLibrary/fourier_transforms.py:49-88
//   approximate_unknowns / mtip_projection / fixed_projection    projectLibrary/fxs_Projections.py:723-745, 803-826, 855-863
// Layout: grids (B, Nq, n_phi) complex128, n_phi = 2 M + 1 (harmonic_transforms.py:44-47: odd, so the phi transform is a dense DFT,
// not a radix FFT); harmonic coefficients (B, Nq, n_phi) in numpy's FFT order (orders 0..M, -M..-1); coefficients of the real
// transform (B, Nq, M + 1).  A 2-D grid is 128 x 129 values: the whole problem of a restart is 260 KB, so these are bandwidth-trivial
// kernels -- one workgroup per (restart, shell) with the twiddles in LDS for the DFTs, coalesced reads of the (p, k, order) weight
// array for the Hankel transform (its 8.5 MB at 128 x M64 come from L2 / HBM once per application) -- written for parity, not tuned.
#include "mtip_internal.h"
#include <cmath>

struct mtip2d_ctx {
    int N = 0, n_phi = 0, M = 0, B = 0, device = 0;
    hipStream_t stream = nullptr;
    double2 *d_tw = nullptr;                       // n_phi: exp(-2 pi i j / n_phi)
    double2 *d_wf = nullptr, *d_wi = nullptr;      // (N, N, n_phi) forward / inverse Hankel weights (summed p, new k, order)
    uint8_t* d_unused = nullptr;                   // n_phi: order zeroed by the Hankel pair
    double2 *d_a = nullptr, *d_b = nullptr;        // (B, N, n_phi) work grids
    // projection
    int n_used = 0, zero_pos = -1, zero_id = -1;
    int so_pos = -1;                  // SO_freedom: position (among the used orders) of the order whose unknown is set to 1, or -1
    int* d_order_ids = nullptr;                    // n_used
    double2* d_pm = nullptr;                       // (n_used, N)
    uint8_t* d_rmask = nullptr;                    // (n_used, N): radial mask of the used orders
    double* d_q = nullptr;                         // N
    double2* d_unk = nullptr;                      // (B, n_used)
    double n_particles = 1.0;
    bool have_weights = false;
    // loop operators (step, shrink-wrap)
    RealParams rp{};
    double* d_errw = nullptr;                      // (N, n_phi) weights of the real error metric (integrator weights x metric mask)
    double2 *d_c = nullptr, *d_d = nullptr, *d_e = nullptr;   // more (B, N, n_phi) work grids
    uint8_t* d_sup = nullptr;                      // (B, N, n_phi)
    double* d_red = nullptr;                       // (B, 4) reductions
    bool have_errw = false;
    std::string err;
};

#define C2_CHECK(c, expr)                                                          \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) {                                                     \
            (c)->err = std::string(#expr) + ": " + hipGetErrorString(e_);           \
            return MTIP_EHIP;                                                       \
        }                                                                           \
    } while (0)

static hipError_t c2_copy(mtip2d_ctx* c, void* dst, const void* src, size_t n) {
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return e;
    e = hipMemcpy(dst, src, n, hipMemcpyDefault);
    return e != hipSuccess ? e : hipStreamSynchronize(nullptr);
}

// out[m] = scale * sum_p in[p] exp(sign 2 pi i m p / n) for one (restart, shell) per workgroup; n_out outputs (n for the complex
// transform, M + 1 for the real one); real_in: only Re(in) enters (circularHarmonicTransform_real_forward)
__global__ void __launch_bounds__(256) k2d_dft(const double2* __restrict__ in, double2* __restrict__ out, const double2* __restrict__ tw_g,
                                               int n, int n_out, int sign, double scale, int real_in) {
    HIP_DYNAMIC_SHARED(double2, sm)
    double2* tw = sm;                    // n
    double2* x = sm + n;                 // n
    const size_t row = blockIdx.x;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        tw[e] = tw_g[e];
        const double2 v = in[row * n + e];
        x[e] = real_in == 2 ? make_double2(v.x * v.x + v.y * v.y, 0.0) : (real_in ? make_double2(v.x, 0.0) : v);
    }
    __syncthreads();
    for (int m = threadIdx.x; m < n_out; m += blockDim.x) {
        double2 acc = make_double2(0.0, 0.0);
        int idx = 0;                                                   // (m p) mod n
        for (int p = 0; p < n; ++p) {
            double2 w = tw[idx];
            if (sign > 0) w.y = -w.y;
            acc = cadd(acc, cmul(x[p], w));
            idx += m;
            if (idx >= n) idx -= n;
        }
        out[row * n_out + m] = cscale(acc, scale);
    }
}

// circularHarmonicTransform_real_inverse: x[p] = irfft(c * n, n)[p] = Re c_0 + 2 sum_{m=1..M} Re(c_m e^{2 pi i m p / n}) (odd n); real output
__global__ void __launch_bounds__(256) k2d_irdft(const double2* __restrict__ coef, double* __restrict__ out, const double2* __restrict__ tw_g,
                                                 int n, int M) {
    HIP_DYNAMIC_SHARED(double2, sm)
    double2* tw = sm;
    double2* c = sm + n;
    const size_t row = blockIdx.x;
    for (int e = threadIdx.x; e < n; e += blockDim.x) tw[e] = tw_g[e];
    for (int e = threadIdx.x; e <= M; e += blockDim.x) c[e] = coef[row * (M + 1) + e];
    __syncthreads();
    for (int p = threadIdx.x; p < n; p += blockDim.x) {
        double acc = c[0].x;
        int idx = 0;
        for (int m = 1; m <= M; ++m) {
            idx += p;
            if (idx >= n) idx -= n;
            const double2 w = tw[idx];                                 // e^{-2 pi i m p / n}: conj for the inverse
            acc += 2.0 * (c[m].x * w.x + c[m].y * w.y);
        }
        out[row * n + p] = acc;
    }
}

// out[b, k, m] = sum_p W[p, k, m] c[b, p, m]; unused orders -> 0.  grid (k, b), threads over m (coalesced rows of W and c)
__global__ void __launch_bounds__(256) k2d_hankel(const double2* __restrict__ c, double2* __restrict__ out, const double2* __restrict__ W,
                                                  const uint8_t* __restrict__ unused, int N, int n) {
    const int k = blockIdx.x, b = blockIdx.y;
    for (int m = threadIdx.x; m < n; m += blockDim.x) {
        double2 acc = make_double2(0.0, 0.0);
        if (!unused[m])
            for (int p = 0; p < N; ++p) acc = cadd(acc, cmul(W[((size_t)p * N + k) * n + m], c[((size_t)b * N + p) * n + m]));
        out[((size_t)b * N + k) * n + m] = acc;
    }
}

// one workgroup per restart: u_j = <I[:, id_j], v_j>_q / |.| (1 when the scalar product vanishes), then I' (fxs_Projections.py:723-745,
// 803-826, 855-863)
__global__ void __launch_bounds__(256) k2d_project(const double2* __restrict__ I, double2* __restrict__ out, double2* __restrict__ unk,
                                                   const double2* __restrict__ pm, const uint8_t* __restrict__ rmask,
                                                   const int* __restrict__ order_ids, const double* __restrict__ q, int N, int n_coef,
                                                   int n_used, int zero_pos, int zero_id, double inv_sqrt_np, int so_pos) {
    HIP_DYNAMIC_SHARED(double2, sm)                // n_used unknowns
    const int b = blockIdx.x;
    const double2* Ib = I + (size_t)b * N * n_coef;
    double2* ob = out + (size_t)b * N * n_coef;
    for (int j = threadIdx.x; j < n_used; j += blockDim.x) {
        const int id = order_ids[j];
        double2 sp = make_double2(0.0, 0.0);
        for (int qq = 0; qq < N; ++qq) sp = cadd(sp, cscale(cmulc(Ib[(size_t)qq * n_coef + id], pm[(size_t)j * N + qq]), q[qq]));
        const double a = sqrt(cabs2(sp));
        double2 u = (sp.x != 0.0 || sp.y != 0.0) ? make_double2(sp.x / a, sp.y / a) : make_double2(1.0, 0.0);
        if (j == so_pos) u = make_double2(1.0, 0.0);                 // SO_freedom (fxs_Projections.py:744-750)
        sm[j] = u;
        unk[(size_t)b * n_used + j] = u;
    }
    for (int e = threadIdx.x; e < N * n_coef; e += blockDim.x) ob[e] = Ib[e];
    __syncthreads();
    for (int e = threadIdx.x; e < N * n_used; e += blockDim.x) {
        const int qq = e / n_used, j = e - qq * n_used;
        if (!rmask[(size_t)j * N + qq]) continue;
        const double2 v = pm[(size_t)j * N + qq];
        ob[(size_t)qq * n_coef + order_ids[j]] = (j == zero_pos) ? v : cmul(v, sm[j]);
    }
    __syncthreads();
    if (zero_id >= 0)
        for (int qq = threadIdx.x; qq < N; qq += blockDim.x) ob[(size_t)qq * n_coef + zero_id] = cscale(ob[(size_t)qq * n_coef + zero_id], inv_sqrt_np);
}

// F' = F sqrt(I' / |F|^2) where |F|^2 >= 0 and I' >= 0, else 0 (project_to_modified_intensity, fxs_Projections.py:899-909);
// I' real grid (B, N, n)
__global__ void __launch_bounds__(256) k2d_modulus(const double2* __restrict__ F, const double* __restrict__ Inew, double2* __restrict__ out,
                                                   long long total) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const double2 f = F[e];
    const double I = f.x * f.x + f.y * f.y, In = Inew[e];
    const bool ok = (I >= 0.0) && (In >= 0.0);
    out[e] = cscale(f, ok ? sqrt(In / I) : 0.0);
}

// the real-space stage of a step (one workgroup per restart): w = rho' (+ (rho_in - IFT(F)) above shell 0 with ft_stab:
// add_above_zero_index, misk.py:326-329), P = real projection, HIO / ER, and the l2_projection_diff sums with the metric weights
__global__ void __launch_bounds__(256) k2d_real_update(const double2* __restrict__ rho_p, const double2* __restrict__ rho_in,
                                                       const double2* __restrict__ rho_rt, const uint8_t* __restrict__ sup,
                                                       const double* __restrict__ errw, double2* __restrict__ out, double* __restrict__ red,
                                                       RealParams rp, int method, double beta, int ft_stab, int N, int n) {
    __shared__ double s_n[4], s_d[4];
    const int b = blockIdx.x;
    const size_t G = (size_t)N * n;
    double num = 0.0, den = 0.0;
    for (size_t e = threadIdx.x; e < G; e += blockDim.x) {
        const size_t i = (size_t)b * G + e;
        double2 w = rho_p[i];
        const double2 pv = rho_in[i];
        if (ft_stab && e >= (size_t)n) w = cadd(w, csub(pv, rho_rt[i]));
        double2 P;
        out[i] = real_update_point(rp, method, beta, w, pv, sup[i] != 0, P);
        const double wg = errw[e];
        const double dx = w.x - P.x, dy = w.y - P.y;
        num = fma(wg, dx * dx + dy * dy, num);
        den = fma(wg, w.x * w.x + w.y * w.y, den);
    }
    for (int o = 32; o > 0; o >>= 1) {
        num += __shfl_xor(num, o, 64);
        den += __shfl_xor(den, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        s_n[threadIdx.x >> 6] = num;
        s_d[threadIdx.x >> 6] = den;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, d = 0.0;
        for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) {
            a += s_n[wv];
            d += s_d[wv];
        }
        red[2 * b] = a;
        red[2 * b + 1] = d;
    }
}

// shrink-wrap (fxs_Projections.py:245-258, 294-298): |rho| -> FT -> x Gaussian(q, sigma) -> IFT is done by the caller's launches;
// these two kernels are the elementwise ends: abs, the Gaussian factor, and the threshold between min and max of the clamped result
__global__ void __launch_bounds__(256) k2d_abs(const double2* __restrict__ in, double2* __restrict__ out, long long total) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < total) out[e] = make_double2(sqrt(in[e].x * in[e].x + in[e].y * in[e].y), 0.0);
}

__global__ void __launch_bounds__(256) k2d_gauss(double2* __restrict__ F, const double* __restrict__ q, double sigma, int N, int n, long long total) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int qi = (int)((e / n) % N);
    const double a = 1.0 / (2.0 * sigma * sigma), pi = 3.14159265358979323846;
    const double q2 = q[qi] * q[qi];
    F[e] = cscale(F[e], sqrt(pi / a) * exp(-pi * pi * q2 * q2 / a));                  // gaussian_fourier_transformed_spherical: q^4 (mathLibrary.py:616-624)
}

__global__ void __launch_bounds__(256) k2d_sw_mask(const double2* __restrict__ conv, uint8_t* __restrict__ mask, double threshold, int N, int n) {
    __shared__ double s_lo[4], s_hi[4];
    const int b = blockIdx.x;
    const size_t G = (size_t)N * n;
    double lo = HUGE_VAL, hi = -HUGE_VAL;
    for (size_t e = threadIdx.x; e < G; e += blockDim.x) {
        const double c = fmax(conv[(size_t)b * G + e].x, 0.0);
        lo = fmin(lo, c);
        hi = fmax(hi, c);
    }
    for (int o = 32; o > 0; o >>= 1) {
        lo = fmin(lo, __shfl_xor(lo, o, 64));
        hi = fmax(hi, __shfl_xor(hi, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        s_lo[threadIdx.x >> 6] = lo;
        s_hi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    lo = s_lo[0];
    hi = s_hi[0];
    for (int wv = 1; wv < (int)(blockDim.x >> 6); ++wv) {
        lo = fmin(lo, s_lo[wv]);
        hi = fmax(hi, s_hi[wv]);
    }
    const double cut = lo + threshold * (hi - lo);
    for (size_t e = threadIdx.x; e < G; e += blockDim.x) mask[(size_t)b * G + e] = fmax(conv[(size_t)b * G + e].x, 0.0) >= cut ? 1 : 0;
}

static void c2_free(mtip2d_ctx* c) {
    for (void* p : {(void*)c->d_tw, (void*)c->d_wf, (void*)c->d_wi, (void*)c->d_unused, (void*)c->d_a, (void*)c->d_b, (void*)c->d_order_ids,
                    (void*)c->d_pm, (void*)c->d_rmask, (void*)c->d_q, (void*)c->d_unk, (void*)c->d_errw, (void*)c->d_c, (void*)c->d_d,
                    (void*)c->d_e, (void*)c->d_sup, (void*)c->d_red})
        if (p) (void)hipFree(p);
    if (c->stream) (void)hipStreamDestroy(c->stream);
}

static void c2_dft(mtip2d_ctx* c, const double2* in, double2* out, int inverse) {
    const size_t lds = (size_t)2 * c->n_phi * sizeof(double2);
    hipLaunchKernelGGL(k2d_dft, dim3((unsigned)(c->B * c->N)), dim3(256), lds, c->stream, in, out, (const double2*)c->d_tw, c->n_phi, c->n_phi,
                       inverse ? +1 : -1, inverse ? 1.0 : 1.0 / c->n_phi, 0);
}

static void c2_hankel(mtip2d_ctx* c, const double2* in, double2* out, int inverse) {
    hipLaunchKernelGGL(k2d_hankel, dim3((unsigned)c->N, (unsigned)c->B), dim3(256), 0, c->stream, in, out,
                       (const double2*)(inverse ? c->d_wi : c->d_wf), (const uint8_t*)c->d_unused, c->N, c->n_phi);
}

extern "C" {

mtip2d_ctx* mtip2d_create(int n_radial, int n_phi, int n_batch, int device) {
    if (n_radial < 2 || n_phi < 3 || !(n_phi & 1) || n_phi > 2047 || n_batch < 1) return nullptr;
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || device < 0 || device >= nd) return nullptr;
    mtip2d_ctx* c = new mtip2d_ctx();
    c->N = n_radial; c->n_phi = n_phi; c->M = (n_phi - 1) / 2; c->B = n_batch; c->device = device;
    (void)hipSetDevice(device);
    const size_t G = (size_t)n_batch * n_radial * n_phi;
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc((void**)&c->d_tw, n_phi * sizeof(double2)) == hipSuccess &&
              hipMalloc((void**)&c->d_wf, (size_t)n_radial * n_radial * n_phi * sizeof(double2)) == hipSuccess &&
              hipMalloc((void**)&c->d_wi, (size_t)n_radial * n_radial * n_phi * sizeof(double2)) == hipSuccess &&
              hipMalloc((void**)&c->d_unused, n_phi) == hipSuccess && hipMalloc((void**)&c->d_a, G * sizeof(double2)) == hipSuccess &&
              hipMalloc((void**)&c->d_b, G * sizeof(double2)) == hipSuccess;
    if (ok) {
        std::vector<double2> tw(n_phi);
        const double pi = 3.14159265358979323846;
        for (int j = 0; j < n_phi; ++j) tw[j] = make_double2(std::cos(2 * pi * j / n_phi), -std::sin(2 * pi * j / n_phi));
        ok = c2_copy(c, c->d_tw, tw.data(), n_phi * sizeof(double2)) == hipSuccess;
    }
    if (!ok) {
        c2_free(c);
        delete c;
        return nullptr;
    }
    return c;
}

void mtip2d_destroy(mtip2d_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    c2_free(c);
    delete c;
}

const char* mtip2d_last_error(const mtip2d_ctx* c) { return c ? c->err.c_str() : "null context"; }

int mtip2d_set_hankel_weights(mtip2d_ctx* c, const mtip_cdouble* forward, const mtip_cdouble* inverse, const uint8_t* unused_orders) {
    if (!c) return MTIP_EINVAL;
    if (!forward || !inverse || !unused_orders) {
        c->err = "hankel weights: null buffer";
        return MTIP_EINVAL;
    }
    (void)hipSetDevice(c->device);
    const size_t n = (size_t)c->N * c->N * c->n_phi * sizeof(double2);
    C2_CHECK(c, c2_copy(c, c->d_wf, forward, n));
    C2_CHECK(c, c2_copy(c, c->d_wi, inverse, n));
    C2_CHECK(c, c2_copy(c, c->d_unused, unused_orders, c->n_phi));
    c->have_weights = true;
    return MTIP_OK;
}

int mtip2d_set_projection(mtip2d_ctx* c, int n_used, const int32_t* order_ids, const mtip_cdouble* pm, const uint8_t* radial_mask,
                          const double* radial_points, double n_particles) {
    if (!c) return MTIP_EINVAL;
    if (n_used < 1 || n_used > c->M + 1 || !order_ids || !pm || !radial_mask || !radial_points || !(n_particles > 0)) {
        c->err = "projection: 1 <= n_used <= M + 1, buffers not null, n_particles > 0";
        return MTIP_EINVAL;
    }
    c->zero_pos = -1;
    c->zero_id = -1;
    for (int j = 0; j < n_used; ++j) {
        if (order_ids[j] < 0 || order_ids[j] > c->M || (j > 0 && order_ids[j] <= order_ids[j - 1])) {
            // (upstream assigns through boolean masks in row-major order: only ascending ids keep columns and vectors paired, 806-818)
            c->err = "projection: order ids must be ascending and within 0..M";
            return MTIP_EINVAL;
        }
        if (order_ids[j] == 0) {
            c->zero_pos = j;
            c->zero_id = 0;
        }
    }
    if (c->zero_id < 0) {
        c->err = "projection: order 0 must be among the used orders (fxs_Projections.py:852-861 indexes it unconditionally)";
        return MTIP_EINVAL;
    }
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (void* p : {(void*)c->d_order_ids, (void*)c->d_pm, (void*)c->d_rmask, (void*)c->d_q, (void*)c->d_unk})
        if (p) (void)hipFree(p);
    c->d_order_ids = nullptr; c->d_pm = nullptr; c->d_rmask = nullptr; c->d_q = nullptr; c->d_unk = nullptr;
    C2_CHECK(c, hipMalloc((void**)&c->d_order_ids, n_used * sizeof(int)));
    C2_CHECK(c, hipMalloc((void**)&c->d_pm, (size_t)n_used * c->N * sizeof(double2)));
    C2_CHECK(c, hipMalloc((void**)&c->d_rmask, (size_t)n_used * c->N));
    C2_CHECK(c, hipMalloc((void**)&c->d_q, c->N * sizeof(double)));
    C2_CHECK(c, hipMalloc((void**)&c->d_unk, (size_t)c->B * n_used * sizeof(double2)));
    C2_CHECK(c, c2_copy(c, c->d_order_ids, order_ids, n_used * sizeof(int)));
    C2_CHECK(c, c2_copy(c, c->d_pm, pm, (size_t)n_used * c->N * sizeof(double2)));
    C2_CHECK(c, c2_copy(c, c->d_rmask, radial_mask, (size_t)n_used * c->N));
    C2_CHECK(c, c2_copy(c, c->d_q, radial_points, c->N * sizeof(double)));
    c->n_used = n_used;
    c->n_particles = n_particles;
    c->so_pos = -1;
    return MTIP_OK;
}

int mtip2d_op_harmonic(mtip2d_ctx* c, const mtip_cdouble* in, mtip_cdouble* out, int inverse) {
    if (!c) return MTIP_EINVAL;
    if (!in || !out) {
        c->err = "harmonic transform: null buffer";
        return MTIP_EINVAL;
    }
    (void)hipSetDevice(c->device);
    const size_t n = (size_t)c->B * c->N * c->n_phi * sizeof(double2);
    C2_CHECK(c, c2_copy(c, c->d_a, in, n));
    c2_dft(c, c->d_a, c->d_b, inverse);
    C2_CHECK(c, c2_copy(c, out, c->d_b, n));
    C2_CHECK(c, hipGetLastError());
    return MTIP_OK;
}

int mtip2d_op_real_harmonic_forward(mtip2d_ctx* c, const mtip_cdouble* grid, mtip_cdouble* coef) {
    if (!c) return MTIP_EINVAL;
    if (!grid || !coef) {
        c->err = "real harmonic transform: null buffer";
        return MTIP_EINVAL;
    }
    (void)hipSetDevice(c->device);
    C2_CHECK(c, c2_copy(c, c->d_a, grid, (size_t)c->B * c->N * c->n_phi * sizeof(double2)));
    hipLaunchKernelGGL(k2d_dft, dim3((unsigned)(c->B * c->N)), dim3(256), (size_t)2 * c->n_phi * sizeof(double2), c->stream, (const double2*)c->d_a,
                       c->d_b, (const double2*)c->d_tw, c->n_phi, c->M + 1, -1, 1.0 / c->n_phi, 1);
    C2_CHECK(c, c2_copy(c, coef, c->d_b, (size_t)c->B * c->N * (c->M + 1) * sizeof(double2)));
    C2_CHECK(c, hipGetLastError());
    return MTIP_OK;
}

int mtip2d_op_real_harmonic_inverse(mtip2d_ctx* c, const mtip_cdouble* coef, double* grid) {
    if (!c) return MTIP_EINVAL;
    if (!grid || !coef) {
        c->err = "real harmonic transform: null buffer";
        return MTIP_EINVAL;
    }
    (void)hipSetDevice(c->device);
    C2_CHECK(c, c2_copy(c, c->d_a, coef, (size_t)c->B * c->N * (c->M + 1) * sizeof(double2)));
    hipLaunchKernelGGL(k2d_irdft, dim3((unsigned)(c->B * c->N)), dim3(256), (size_t)2 * c->n_phi * sizeof(double2), c->stream, (const double2*)c->d_a,
                       reinterpret_cast<double*>(c->d_b), (const double2*)c->d_tw, c->n_phi, c->M);
    C2_CHECK(c, c2_copy(c, grid, c->d_b, (size_t)c->B * c->N * c->n_phi * sizeof(double)));
    C2_CHECK(c, hipGetLastError());
    return MTIP_OK;
}

int mtip2d_op_hankel(mtip2d_ctx* c, const mtip_cdouble* in, mtip_cdouble* out, int inverse) {
    if (!c) return MTIP_EINVAL;
    if (!c->have_weights) {
        c->err = "mtip2d_set_hankel_weights has not been called";
        return MTIP_ESTATE;
    }
    if (!in || !out) {
        c->err = "hankel: null buffer";
        return MTIP_EINVAL;
    }
    (void)hipSetDevice(c->device);
    const size_t n = (size_t)c->B * c->N * c->n_phi * sizeof(double2);
    C2_CHECK(c, c2_copy(c, c->d_a, in, n));
    c2_hankel(c, c->d_a, c->d_b, inverse);
    C2_CHECK(c, c2_copy(c, out, c->d_b, n));
    C2_CHECK(c, hipGetLastError());
    return MTIP_OK;
}

/* generate_ft (fourier_transforms.py:57-88): harmonic forward, Hankel (forward / inverse weights), harmonic inverse */
int mtip2d_op_fourier_transform(mtip2d_ctx* c, const mtip_cdouble* in, mtip_cdouble* out, int inverse) {
    if (!c) return MTIP_EINVAL;
    if (!c->have_weights) {
        c->err = "mtip2d_set_hankel_weights has not been called";
        return MTIP_ESTATE;
    }
    if (!in || !out) {
        c->err = "fourier transform: null buffer";
        return MTIP_EINVAL;
    }
    (void)hipSetDevice(c->device);
    const size_t n = (size_t)c->B * c->N * c->n_phi * sizeof(double2);
    C2_CHECK(c, c2_copy(c, c->d_a, in, n));
    c2_dft(c, c->d_a, c->d_b, 0);
    c2_hankel(c, c->d_b, c->d_a, inverse);
    c2_dft(c, c->d_a, c->d_b, 1);
    C2_CHECK(c, c2_copy(c, out, c->d_b, n));
    C2_CHECK(c, hipGetLastError());
    return MTIP_OK;
}

int mtip2d_set_so_freedom(mtip2d_ctx* c, int position) {
    if (!c) return MTIP_EINVAL;
    if (position < -1 || position >= c->n_used) {
        c->err = "so_freedom: position among the used orders of the current projection, or -1";
        return MTIP_EINVAL;
    }
    c->so_pos = position;
    return MTIP_OK;
}

int mtip2d_op_project(mtip2d_ctx* c, const mtip_cdouble* I, mtip_cdouble* out, mtip_cdouble* unknowns) {
    if (!c) return MTIP_EINVAL;
    if (c->n_used == 0) {
        c->err = "mtip2d_set_projection has not been called";
        return MTIP_ESTATE;
    }
    if (!I || !out) {
        c->err = "project: null buffer";
        return MTIP_EINVAL;
    }
    (void)hipSetDevice(c->device);
    const int n_coef = c->M + 1;
    const size_t n = (size_t)c->B * c->N * n_coef * sizeof(double2);
    C2_CHECK(c, c2_copy(c, c->d_a, I, n));
    hipLaunchKernelGGL(k2d_project, dim3((unsigned)c->B), dim3(256), (size_t)c->n_used * sizeof(double2), c->stream, (const double2*)c->d_a, c->d_b,
                       c->d_unk, (const double2*)c->d_pm, (const uint8_t*)c->d_rmask, (const int*)c->d_order_ids, (const double*)c->d_q, c->N,
                       n_coef, c->n_used, c->zero_pos, c->zero_id, 1.0 / std::sqrt(c->n_particles), c->so_pos);
    C2_CHECK(c, c2_copy(c, out, c->d_b, n));
    if (unknowns) C2_CHECK(c, c2_copy(c, unknowns, c->d_unk, (size_t)c->B * c->n_used * sizeof(double2)));
    C2_CHECK(c, hipGetLastError());
    return MTIP_OK;
}

int mtip2d_set_real_constraints(mtip2d_ctx* c, uint32_t flags, double lo, double hi, double imag_thr, uint32_t hio_flags) {
    if (!c) return MTIP_EINVAL;
    c->rp.flags = flags; c->rp.hio_flags = hio_flags; c->rp.lo = lo; c->rp.hi = hi; c->rp.imag_thr = imag_thr;
    return MTIP_OK;
}

/* weights (Nq, n_phi) of the real error metric: integrator weights times the metric's mask (fxs_IO_methods.py:97-128 with
 * PolarIntegrator, mathLibrary.py:1242-1265) */
int mtip2d_set_error_weights(mtip2d_ctx* c, const double* weights) {
    if (!c || !weights) return MTIP_EINVAL;
    (void)hipSetDevice(c->device);
    const size_t G = (size_t)c->N * c->n_phi, BG = (size_t)c->B * G;
    // (each buffer on its own: a call that failed half way is completed by the next one)
    if (!c->d_errw) C2_CHECK(c, hipMalloc((void**)&c->d_errw, G * sizeof(double)));
    if (!c->d_c) C2_CHECK(c, hipMalloc((void**)&c->d_c, BG * sizeof(double2)));
    if (!c->d_d) C2_CHECK(c, hipMalloc((void**)&c->d_d, BG * sizeof(double2)));
    if (!c->d_e) C2_CHECK(c, hipMalloc((void**)&c->d_e, BG * sizeof(double2)));
    if (!c->d_sup) C2_CHECK(c, hipMalloc((void**)&c->d_sup, BG));
    if (!c->d_red) C2_CHECK(c, hipMalloc((void**)&c->d_red, (size_t)c->B * 4 * sizeof(double)));
    C2_CHECK(c, c2_copy(c, c->d_errw, weights, G * sizeof(double)));
    c->have_errw = true;
    return MTIP_OK;
}

/* one HIO (method 0) / ER (method 1) step of the 2-D loop (sketches reconstruct.py:518-528, 576-593 with the 2-D operators):
 * F = FT(rho); I_m = real harmonic transform of |F|^2; unknowns + projection; I' back on the grid; F' = F sqrt(I' / |F|^2);
 * rho' = IFT(F') (+ rho - IFT(F) above shell 0 with ft_stab); real-space projection with `support` (n_batch, Nq, n_phi; the
 * effective one) + HIO / ER; error = l2_projection_diff.  F_new, rho_new (n_batch, Nq, n_phi), err (n_batch), unknowns
 * (n_batch, n_used) or NULL. */
int mtip2d_op_step(mtip2d_ctx* c, int method, int ft_stab, double beta, const mtip_cdouble* rho, const uint8_t* support,
                   mtip_cdouble* F_new, mtip_cdouble* rho_new, double* err, mtip_cdouble* unknowns) {
    return mtip2d_op_step_ex(c, method, ft_stab, beta, rho, support, nullptr, F_new, rho_new, err, unknowns, nullptr, nullptr);
}

int mtip2d_op_step_ex(mtip2d_ctx* c, int method, int ft_stab, double beta, const mtip_cdouble* rho, const uint8_t* support,
                      const double* fixed_intensity, mtip_cdouble* F_new, mtip_cdouble* rho_new, double* err, mtip_cdouble* unknowns,
                      mtip_cdouble* F_out, mtip_cdouble* I_out) {
    if (!c) return MTIP_EINVAL;
    if (!c->have_weights || c->n_used == 0 || !c->have_errw) {
        c->err = "step: hankel weights, projection and error weights must be set first";
        return MTIP_ESTATE;
    }
    const bool fxs = (method == MTIP_HIO || method == MTIP_ER);
    if (method < 0 || method > 3 || !rho || !support || !F_new || !rho_new || !err) {
        c->err = "step: method 0 (HIO), 1 (ER), 2 (HIO_non_FXS) or 3 (ER_non_FXS), buffers not null";
        return MTIP_EINVAL;
    }
    if (!fxs && !fixed_intensity) {
        c->err = "step: the *_non_FXS methods need the fixed intensity grid";
        return MTIP_EINVAL;
    }
    (void)hipSetDevice(c->device);
    const int N = c->N, n = c->n_phi, M1 = c->M + 1, B = c->B;
    const size_t BG = (size_t)B * N * n;
    const long long total = (long long)BG;
    const size_t lds = (size_t)2 * n * sizeof(double2);
    const unsigned rows = (unsigned)(B * N);
    C2_CHECK(c, c2_copy(c, c->d_e, rho, BG * sizeof(double2)));                      // rho_in
    C2_CHECK(c, c2_copy(c, c->d_sup, support, BG));
    // F = FT(rho) -> d_c
    c2_dft(c, c->d_e, c->d_a, 0);
    c2_hankel(c, c->d_a, c->d_b, 0);
    c2_dft(c, c->d_b, c->d_c, 1);
    if (F_out) C2_CHECK(c, c2_copy(c, F_out, c->d_c, BG * sizeof(double2)));
    if (!fxs) {
        // MTIP_start_non_FXS (reconstruct.py:530-535): F' = F sqrt(fixed / |F|^2), no harmonic transform and no unknowns
        C2_CHECK(c, c2_copy(c, c->d_a, fixed_intensity, BG * sizeof(double)));
        unknowns = nullptr;
    } else {
    // I_m of |F|^2 -> d_a (B, N, M + 1); projection -> d_b; I' (real grid) -> d_a (as doubles)
    hipLaunchKernelGGL(k2d_dft, dim3(rows), dim3(256), lds, c->stream, (const double2*)c->d_c, c->d_a, (const double2*)c->d_tw, n, M1, -1, 1.0 / n, 2);
    if (I_out) C2_CHECK(c, c2_copy(c, I_out, c->d_a, (size_t)B * N * M1 * sizeof(double2)));
    hipLaunchKernelGGL(k2d_project, dim3((unsigned)B), dim3(256), (size_t)c->n_used * sizeof(double2), c->stream, (const double2*)c->d_a, c->d_b,
                       c->d_unk, (const double2*)c->d_pm, (const uint8_t*)c->d_rmask, (const int*)c->d_order_ids, (const double*)c->d_q, N, M1,
                       c->n_used, c->zero_pos, c->zero_id, 1.0 / std::sqrt(c->n_particles), c->so_pos);
    hipLaunchKernelGGL(k2d_irdft, dim3(rows), dim3(256), lds, c->stream, (const double2*)c->d_b, reinterpret_cast<double*>(c->d_a),
                       (const double2*)c->d_tw, n, c->M);
    }
    // F' -> d_d
    hipLaunchKernelGGL(k2d_modulus, dim3((unsigned)div_up(total, 256)), dim3(256), 0, c->stream, (const double2*)c->d_c,
                       (const double*)reinterpret_cast<double*>(c->d_a), c->d_d, total);
    // rho' = IFT(F') -> d_a
    c2_dft(c, c->d_d, c->d_a, 0);
    c2_hankel(c, c->d_a, c->d_b, 1);
    c2_dft(c, c->d_b, c->d_a, 1);
    C2_CHECK(c, c2_copy(c, F_new, c->d_d, BG * sizeof(double2)));                   // F' delivered: d_d is free for the outputs below
    if (ft_stab) {                                                                  // IFT(F) -> d_b (through d_d)
        c2_dft(c, c->d_c, c->d_b, 0);
        c2_hankel(c, c->d_b, c->d_d, 1);
        c2_dft(c, c->d_d, c->d_b, 1);
    }
    hipLaunchKernelGGL(k2d_real_update, dim3((unsigned)B), dim3(256), 0, c->stream, (const double2*)c->d_a, (const double2*)c->d_e,
                       (const double2*)c->d_b, (const uint8_t*)c->d_sup, (const double*)c->d_errw, c->d_d, c->d_red, c->rp, method & 1, beta,
                       ft_stab ? 1 : 0, N, n);
    C2_CHECK(c, c2_copy(c, rho_new, c->d_d, BG * sizeof(double2)));
    std::vector<double> red((size_t)B * 2);
    C2_CHECK(c, c2_copy(c, red.data(), c->d_red, red.size() * sizeof(double)));
    for (int b = 0; b < B; ++b) err[b] = red[2 * b + 1] != 0.0 ? red[2 * b] / red[2 * b + 1] : HUGE_VAL;   // fxs_IO_methods.py:121-126
    if (unknowns) C2_CHECK(c, c2_copy(c, unknowns, c->d_unk, (size_t)B * c->n_used * sizeof(double2)));
    C2_CHECK(c, hipGetLastError());
    return MTIP_OK;
}

/* the SW sketch (reconstruct.py:598-605; fxs_Projections.py:245-258, 294-298): mask = c >= min + threshold (max - min) of
 * c = max(Re IFT(FT(|rho|) gaussian(q, sigma)), 0); mask (n_batch, Nq, n_phi) */
int mtip2d_op_shrinkwrap(mtip2d_ctx* c, const mtip_cdouble* rho, double sigma, double threshold, uint8_t* mask) {
    if (!c) return MTIP_EINVAL;
    if (!c->have_weights || !c->have_errw || c->d_q == nullptr) {
        c->err = "shrinkwrap: hankel weights, projection (radial points) and error weights must be set first";
        return MTIP_ESTATE;
    }
    if (!rho || !mask || !(sigma > 0.0)) {
        c->err = "shrinkwrap: null buffer or sigma <= 0";
        return MTIP_EINVAL;
    }
    (void)hipSetDevice(c->device);
    const size_t BG = (size_t)c->B * c->N * c->n_phi;
    const long long total = (long long)BG;
    C2_CHECK(c, c2_copy(c, c->d_e, rho, BG * sizeof(double2)));
    hipLaunchKernelGGL(k2d_abs, dim3((unsigned)div_up(total, 256)), dim3(256), 0, c->stream, (const double2*)c->d_e, c->d_c, total);
    c2_dft(c, c->d_c, c->d_a, 0);
    c2_hankel(c, c->d_a, c->d_b, 0);
    c2_dft(c, c->d_b, c->d_c, 1);
    hipLaunchKernelGGL(k2d_gauss, dim3((unsigned)div_up(total, 256)), dim3(256), 0, c->stream, c->d_c, (const double*)c->d_q, sigma, c->N, c->n_phi, total);
    c2_dft(c, c->d_c, c->d_a, 0);
    c2_hankel(c, c->d_a, c->d_b, 1);
    c2_dft(c, c->d_b, c->d_c, 1);
    hipLaunchKernelGGL(k2d_sw_mask, dim3((unsigned)c->B), dim3(256), 0, c->stream, (const double2*)c->d_c, c->d_sup, threshold, c->N, c->n_phi);
    C2_CHECK(c, c2_copy(c, mask, c->d_sup, BG));
    C2_CHECK(c, hipGetLastError());
    return MTIP_OK;
}

}  // extern "C"
