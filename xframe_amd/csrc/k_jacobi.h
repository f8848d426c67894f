// Pieces shared by the one-sided Jacobi polar-factor kernels (k_proj.hip: complex X_l; k_projr.hip: the real form of
// X_l for real projection matrices): constants, reciprocal square roots, DPP lane sums, pairing-schedule flags.
#pragma once
#include "mtip_internal.h"

#define JAC_MAX_SWEEPS 40
#define JAC_TOL 1e-14
#define JAC_DEFLATE 1e-15   // columns below this fraction of the largest column are numerical zeros
#define JL_EARLY 1e-6       // a sweep that stayed below this leaves ~1e-12 of non-orthogonality (operator tolerance 1e-10)
// pairing-schedule entry (build_jacobi_schedule): resident | mover << 8 | flags
#define JS_ACTIVE (1 << 17)
#define JS_WB (1 << 16)

// 1/sqrt(x) to full double precision from the hardware estimate (two Newton steps); the rotation only needs
// cs^2 + sn^2 = 1 and |em| = 1 to rounding, not a correctly rounded quotient.
__device__ __forceinline__ double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    return y;
}
__device__ __forceinline__ double fast_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = y * (2.0 - x * y);
    y = y * (2.0 - x * y);
    return y;
}

// sum over the 8 lanes of a pair-group with DPP moves (3 VALU ops per stage instead of two ds_bpermute each):
// xor 1 = quad_perm [1,0,3,2], xor 2 = quad_perm [2,3,0,1], then row_half_mirror (lane i <-> 7-i) crosses quads.
template <int CTRL>
__device__ __forceinline__ double dpp_add(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return v + __hiloint2double(hi2, lo2);
}
template <int TG>
__device__ __forceinline__ double group_sum(double v) {
    v = dpp_add<0xB1>(v);
    v = dpp_add<0x4E>(v);
    v = dpp_add<0x141>(v);
    if (TG == 16) v = dpp_add<0x140>(v);                   // row_mirror (lane i <-> 15-i) joins the two halves
    return v;
}

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi2, lo2);
}

// Sums of FOUR values over the 16 lanes of a pair-group at once ("transposed" reduction): after the xor-1 step a lane
// carries two of the four partial sums, after the xor-2 step one, the steps across quads (row_ror 4, 8 keep lane & 3)
// finish that one sum, and four quad broadcasts hand every lane all four totals: 35 VALU ops instead of 68 for
// four independent butterflies.
template <int TG>
__device__ __forceinline__ void group_sum4(double& v0, double& v1, double& v2, double& v3) {
    if (TG != 16) {
        v0 = group_sum<TG>(v0);
        v1 = group_sum<TG>(v1);
        v2 = group_sum<TG>(v2);
        v3 = group_sum<TG>(v3);
        return;
    }
    const int lane = threadIdx.x;
    const bool p = (lane & 1) != 0, p2 = (lane & 2) != 0;
    // xor 1: odd lanes keep (v2, v3), even lanes keep (v0, v1); the other two go to the neighbour
    const double s0 = p ? v0 : v2, s1 = p ? v1 : v3;
    double u0 = p ? v2 : v0, u1 = p ? v3 : v1;
    u0 += dpp_mov<0xB1>(s0);
    u1 += dpp_mov<0xB1>(s1);
    // xor 2: lanes with bit 1 keep u1, the others u0
    const double s = p2 ? u0 : u1;
    double w = p2 ? u1 : u0;
    w += dpp_mov<0x4E>(s);
    // across the four quads of the group.  Opposite quads first (row_ror 8), then neighbours (row_ror 4): every quad then adds
    // (W_q + W_q+2) + (W_q+1 + W_q+3) -- the same two operands in either order, so all 16 lanes end with bitwise identical sums.
    // (Neighbours first gave the even and the odd quads differently associated sums, one ulp apart: the lanes of ONE column pair
    // then derived different rotations for their rows -- harmless for small angles, a 1e-8 stall for the 45 degree rotations
    // inside a cluster of equal singular values, found with k_sym_eig.)
    w += dpp_mov<0x128>(w);
    w += dpp_mov<0x124>(w);
    // lane & 3 = p + 2 p2 holds value 2 p + p2:  quad lane 0 -> v0, 2 -> v1, 1 -> v2, 3 -> v3
    v0 = dpp_mov<0x00>(w);
    v1 = dpp_mov<0xAA>(w);
    v2 = dpp_mov<0x55>(w);
    v3 = dpp_mov<0xFF>(w);
}

