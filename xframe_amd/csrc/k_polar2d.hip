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
    int* d_order_ids = nullptr;                    // n_used
    double2* d_pm = nullptr;                       // (n_used, N)
    uint8_t* d_rmask = nullptr;                    // (n_used, N): radial mask of the used orders
    double* d_q = nullptr;                         // N
    double2* d_unk = nullptr;                      // (B, n_used)
    double n_particles = 1.0;
    bool have_weights = false;
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
        x[e] = real_in ? make_double2(v.x, 0.0) : v;
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
                                                   int n_used, int zero_pos, int zero_id, double inv_sqrt_np) {
    HIP_DYNAMIC_SHARED(double2, sm)                // n_used unknowns
    const int b = blockIdx.x;
    const double2* Ib = I + (size_t)b * N * n_coef;
    double2* ob = out + (size_t)b * N * n_coef;
    for (int j = threadIdx.x; j < n_used; j += blockDim.x) {
        const int id = order_ids[j];
        double2 sp = make_double2(0.0, 0.0);
        for (int qq = 0; qq < N; ++qq) sp = cadd(sp, cscale(cmulc(Ib[(size_t)qq * n_coef + id], pm[(size_t)j * N + qq]), q[qq]));
        const double a = sqrt(cabs2(sp));
        const double2 u = (sp.x != 0.0 || sp.y != 0.0) ? make_double2(sp.x / a, sp.y / a) : make_double2(1.0, 0.0);
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

static void c2_free(mtip2d_ctx* c) {
    for (void* p : {(void*)c->d_tw, (void*)c->d_wf, (void*)c->d_wi, (void*)c->d_unused, (void*)c->d_a, (void*)c->d_b, (void*)c->d_order_ids,
                    (void*)c->d_pm, (void*)c->d_rmask, (void*)c->d_q, (void*)c->d_unk})
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
                       n_coef, c->n_used, c->zero_pos, c->zero_id, 1.0 / std::sqrt(c->n_particles));
    C2_CHECK(c, c2_copy(c, out, c->d_b, n));
    if (unknowns) C2_CHECK(c, c2_copy(c, unknowns, c->d_unk, (size_t)c->B * c->n_used * sizeof(double2)));
    C2_CHECK(c, hipGetLastError());
    return MTIP_OK;
}

}  // extern "C"
