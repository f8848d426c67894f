// Pieces shared by the register-FFT spherical-harmonic kernels (k_sht_reg.hip, k_sht_chain.hip): small in-register FFTs,
// radix pairs of the supported n_phi, pass sizes.
#pragma once
#include "mtip_internal.h"
#include <algorithm>

#define SR_THREADS 256

// exp(-2 pi i j / 16), j = 0..7
__device__ __forceinline__ double2 tw16(int j) {
    const double c1 = 0.92387953251128673848, s1 = 0.38268343236508978178, h = 0.70710678118654752440;
    switch (j) {
        case 0: return make_double2(1.0, 0.0);
        case 1: return make_double2(c1, -s1);
        case 2: return make_double2(h, -h);
        case 3: return make_double2(s1, -c1);
        case 4: return make_double2(0.0, -1.0);
        case 5: return make_double2(-s1, -c1);
        case 6: return make_double2(-h, -h);
        default: return make_double2(-c1, -s1);
    }
}

// in-register FFT of R points (natural order in and out), decimation in time, fully unrolled
template <int R, bool INV>
struct SmallFFT {
    static __device__ __forceinline__ void run(double2 (&v)[R]) {
        double2 e[R / 2], o[R / 2];
#pragma unroll
        for (int i = 0; i < R / 2; ++i) {
            e[i] = v[2 * i];
            o[i] = v[2 * i + 1];
        }
        SmallFFT<R / 2, INV>::run(e);
        SmallFFT<R / 2, INV>::run(o);
#pragma unroll
        for (int k = 0; k < R / 2; ++k) {
            const int j = k * (16 / R);
            double2 t;
            if (j == 0) {
                t = o[k];
            } else if (j == 4) {
                t = INV ? make_double2(-o[k].y, o[k].x) : make_double2(o[k].y, -o[k].x);   // * (+-i)
            } else {
                double2 w = tw16(j);
                if (INV) w.y = -w.y;
                t = cmul(o[k], w);
            }
            v[k] = cadd(e[k], t);
            v[k + R / 2] = csub(e[k], t);
        }
    }
};
template <bool INV>
struct SmallFFT<1, INV> {
    static __device__ __forceinline__ void run(double2 (&)[1]) {}
};


// Half of an R-point FFT: the outputs n = 2 j + half (j < R / 2, returned in y[j]) of the inputs x[0..R) -- one radix-2
// decimation-in-frequency stage, then an R/2-point FFT.  Two threads share one transform (`half` is wave-uniform where this is
// used: no divergence); each does a little more than half of its work and, more to the point, every wave of the workgroup has work
// in the two FFT steps that 16-point transforms otherwise leave to half of them.
template <int R, bool INV>
__device__ __forceinline__ void half_fft(const double2 (&x)[R], int half, double2 (&y)[R / 2]) {
    static_assert(R == 16 || R == 8, "twiddles come from tw16");
#pragma unroll
    for (int k = 0; k < R / 2; ++k) {
        if (half == 0) {
            y[k] = cadd(x[k], x[k + R / 2]);
        } else {
            const double2 d = csub(x[k], x[k + R / 2]);
            const int j = k * (16 / R);                 // exp(-/+ 2 pi i k / R) = tw16(k 16 / R)
            if (j == 0) {
                y[k] = d;
            } else if (j == 4) {
                y[k] = INV ? make_double2(-d.y, d.x) : make_double2(d.y, -d.x);
            } else {
                double2 w = tw16(j);
                if (INV) w.y = -w.y;
                y[k] = cmul(d, w);
            }
        }
    }
    SmallFFT<R / 2, INV>::run(y);
}

#define SW_THREADS 512

static inline bool reg_radices(int np, int* r1, int* r2) {
    switch (np) {
        case 16: *r1 = 4; *r2 = 4; return true;
        case 32: *r1 = 4; *r2 = 8; return true;
        case 64: *r1 = 8; *r2 = 8; return true;
        case 128: *r1 = 8; *r2 = 16; return true;
        case 256: *r1 = 16; *r2 = 16; return true;
        default: return false;
    }
}

static inline int largest_even_divisor_le(int nt, int cap) {
    for (int v = std::min(nt, cap) & ~1; v >= 2; v -= 2)
        if (nt % v == 0) return v;
    return 0;
}
