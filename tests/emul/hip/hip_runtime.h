// CPU emulation of the small HIP subset used by xframe_amd/csrc  --  TEST INFRASTRUCTURE ONLY.
//
// Purpose: run the *unchanged* kernel sources of xframe_amd/csrc on the host (g++ -x c++ -I tests/emul)
// at toy sizes, so that index arithmetic, LDS usage, barrier placement and out-of-bounds accesses can be
// checked (also under -fsanitize=address) in the build container, which has no GPU.  One OS thread per
// GPU thread, one block at a time; __syncthreads = std::barrier; wave collectives (shuffles, f64 MFMA)
// are emulated with a per-wave exchange buffer.  Nothing here is ever loaded by the product: the
// Python loader only opens libmtip_hip.so, and the -m gpu parity tests run the real HIP build.
#pragma once
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __align__(n) __attribute__((aligned(n)))

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct double2 { double x, y; };
static inline double2 make_double2(double x, double y) { return double2{x, y}; }
struct uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
struct int2 { int x, y; };
static inline int2 make_int2(int x, int y) { return int2{x, y}; }

typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };

namespace emul {
// One OS worker thread runs one block at a time; the GPU threads of that block are cooperative fibers
// (ucontext) that yield at __syncthreads() / wave collectives.  Deterministic and fast enough for CI.
constexpr int WAVE = 64;
constexpr size_t DYN_SMEM_MAX = 160 * 1024;
constexpr size_t DYN_SMEM_GUARD = 16 * 1024;    // canary behind a block's dynamic LDS (stores beyond the allocation abort)
struct alignas(16) WaveBuf { double d[WAVE * 4]; long long i[WAVE]; };
struct Worker;
extern thread_local Worker* W;
extern thread_local dim3 t_idx;          // of the fiber currently running on this worker
extern thread_local int t_linear;
extern thread_local dim3 b_idx;
extern thread_local unsigned char* dyn_smem;
extern dim3 b_dim, g_dim;
void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body);
void block_barrier();
void wave_barrier();
WaveBuf& wave_buf();
void fiber_yield();                 // s_sleep inside a spin wait: let the other threads of the block run
int wave_all(int pred);             // wave vote over the lanes that are still running
unsigned long long wave_ballot(int pred);
inline int lane() { return t_linear % WAVE; }
inline int wave() { return t_linear / WAVE; }
}  // namespace emul

#define threadIdx (emul::t_idx)
#define blockIdx (emul::b_idx)
#define blockDim (emul::b_dim)
#define gridDim (emul::g_dim)
static const int warpSize = 64;

#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(emul::dyn_smem);
#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...) \
    emul::launch((grid), (block), (smem), [&]() { kernel(__VA_ARGS__); })

static inline void __syncthreads() { emul::block_barrier(); }

template <typename T>
static inline T __shfl(T v, int src_lane, int width = 64) {
    static_assert(sizeof(T) <= 8, "shfl type");
    long long raw = 0;
    std::memcpy(&raw, &v, sizeof(T));
    emul::wave_buf().i[emul::lane()] = raw;
    emul::wave_barrier();
    int base = (emul::lane() / width) * width;
    long long r = emul::wave_buf().i[base + (src_lane % width)];
    emul::wave_barrier();
    T out;
    std::memcpy(&out, &r, sizeof(T));
    return out;
}
template <typename T>
static inline T __shfl_xor(T v, int mask, int width = 64) { return __shfl(v, (emul::lane() % width) ^ mask, width); }
template <typename T>
static inline T __shfl_down(T v, unsigned delta, int width = 64) {
    int l = emul::lane() % width;
    return __shfl(v, (l + (int)delta < width) ? l + (int)delta : l, width);
}
template <typename T>
static inline T __shfl_up(T v, unsigned delta, int width = 64) {
    int l = emul::lane() % width;
    return __shfl(v, (l - (int)delta >= 0) ? l - (int)delta : l, width);
}

typedef double emul_v4f64 __attribute__((vector_size(32)));
// v_mfma_f64_16x16x4_f64: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D reg r of lane l = D[(l>>4)+4r][l&15]
static inline emul_v4f64 __builtin_amdgcn_mfma_f64_16x16x4f64(double a, double b, emul_v4f64 c, int, int, int) {
    int l = emul::lane();
    emul::wave_buf().d[l] = a;
    emul::wave_buf().d[64 + l] = b;
    emul::wave_barrier();
    auto& buf = emul::wave_buf();
    emul_v4f64 d = c;
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) + 4 * r;
        double acc = c[r];
        for (int k = 0; k < 4; ++k) acc = std::fma(buf.d[row + 16 * k], buf.d[64 + col + 16 * k], acc);
        d[r] = acc;
    }
    emul::wave_barrier();
    return d;
}

// DPP lane permutations used by the kernels: quad_perm (ctrl < 0x100), row_mirror 0x140, row_half_mirror 0x141, row_ror:n 0x120+n
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int, int, bool) {
    (void)old;
    const int l = emul::lane();
    int from;
    if (ctrl < 0x100) from = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);
    else if (ctrl == 0x141) from = (l & ~7) | (7 - (l & 7));
    else if (ctrl == 0x140) from = (l & ~15) | (15 - (l & 15));
    else if (ctrl > 0x120 && ctrl <= 0x12F) from = (l & ~15) | ((l - (ctrl - 0x120)) & 15);   // row_ror:n
    else { std::fprintf(stderr, "emul: unsupported dpp ctrl 0x%x\n", ctrl); std::abort(); }
    return __shfl(src, from, 64);
}
// v_permlane16_swap_b32 / v_permlane32_swap_b32 (gfx950), semantics measured on the MI355X (scripts/microbench/permlane_swap.hip):
// (vdst_old A, src0_old B) -> {vdst_new, src0_new}; rows of 16 lanes: 16: {[A0 B0 A2 B2], [A1 B1 A3 B3]}, 32: {[A0 A1 B0 B1], [A2 A3 B2 B3]}
struct emul_u2 {
    unsigned v[2];
    unsigned operator[](int i) const { return v[i]; }
};
static inline emul_u2 __builtin_amdgcn_permlane16_swap(unsigned a, unsigned b, bool, bool) {
    const int l = emul::lane(), row = l >> 4, col = l & 15;
    // vdst_new row r: even r -> A[r], odd r -> B[r - 1];  src0_new row r: even r -> A[r + 1], odd r -> B[r]
    const unsigned a_from0 = __shfl(a, (row & ~1) * 16 + col, 64), b_from0 = __shfl(b, (row & ~1) * 16 + col, 64);
    const unsigned a_from1 = __shfl(a, (row | 1) * 16 + col, 64), b_from1 = __shfl(b, (row | 1) * 16 + col, 64);
    emul_u2 r;
    r.v[0] = (row & 1) ? b_from0 : a_from0;
    r.v[1] = (row & 1) ? b_from1 : a_from1;
    return r;
}
static inline emul_u2 __builtin_amdgcn_permlane32_swap(unsigned a, unsigned b, bool, bool) {
    const int l = emul::lane(), half = l >> 5, col = l & 31;
    const unsigned a_lo = __shfl(a, col, 64), a_hi = __shfl(a, 32 + col, 64), b_lo = __shfl(b, col, 64), b_hi = __shfl(b, 32 + col, 64);
    emul_u2 r;
    r.v[0] = half ? b_lo : a_lo;
    r.v[1] = half ? b_hi : a_hi;
    return r;
}
// wave-uniform lane reads (the lane index is uniform in the kernels: an SGPR on the GPU)
static inline int __builtin_amdgcn_readlane(int v, int src_lane) { return __shfl(v, src_lane, 64); }
#define MTIP_PIN_VGPRS4(a, b, c, d)      // register-scheduling fence of the device build: nothing to do on the host
#define MTIP_WAIT_LDS()
#define MTIP_WAVE_LDS_SYNC() emul::wave_barrier()
static inline long long wall_clock64() { static thread_local long long t = 0; return t += 1000; }
static inline int __builtin_amdgcn_readfirstlane(int v) { return __shfl(v, 0, 64); }
// workgroup-scope atomics on LDS words: the fibers of a block share one OS thread, plain accesses are atomic enough
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
// (a polling loop may spin on such a load without s_sleep: let the other fibers of the block run)
#ifndef __HIP_MEMORY_SCOPE_AGENT
#define __HIP_MEMORY_SCOPE_AGENT 4
#endif
template <typename T> static inline T __hip_atomic_load(const T* p, int, int) { emul::fiber_yield(); return *const_cast<const volatile T*>(p); }
template <typename T> static inline void __hip_atomic_store(T* p, T v, int, int) { *const_cast<volatile T*>(p) = v; }
static inline int __double2loint(double v) { long long u; std::memcpy(&u, &v, 8); return (int)(u & 0xffffffffll); }
static inline int __double2hiint(double v) { long long u; std::memcpy(&u, &v, 8); return (int)(u >> 32); }
static inline double __hiloint2double(int hi, int lo) {
    unsigned long long u = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
    double v; std::memcpy(&v, &u, 8); return v;
}

static std::mutex emul_atomic_mutex;
static inline double atomicAdd(double* p, double v) {
    std::lock_guard<std::mutex> g(emul_atomic_mutex);
    double o = *p; *p = o + v; return o;
}
static inline int atomicMin(int* p, int v) {
    const int old = *p;
    if (v < old) *p = v;
    return old;
}
static inline long long __double_as_longlong(double v) { long long r; std::memcpy(&r, &v, 8); return r; }
static inline int atomicAdd(int* p, int v) {
    std::lock_guard<std::mutex> g(emul_atomic_mutex);
    int o = *p; *p = o + v; return o;
}
static inline unsigned atomicAdd(unsigned* p, unsigned v) {
    std::lock_guard<std::mutex> g(emul_atomic_mutex);
    unsigned o = *p; *p = o + v; return o;
}
static inline long long clock64() { static thread_local long long t = 0; return t += 7; }   // a running (per host thread) clock: timers carry values
static inline unsigned __builtin_amdgcn_s_getreg(int) { return 0u; }
static inline void __threadfence() { std::atomic_thread_fence(std::memory_order_seq_cst); }
static inline void __threadfence_block() { std::atomic_thread_fence(std::memory_order_seq_cst); }
static inline void __builtin_amdgcn_s_sleep(int) { emul::fiber_yield(); }
static inline void __builtin_amdgcn_wave_barrier() { emul::wave_barrier(); }
static inline int __all(int pred) { return emul::wave_all(pred); }
static inline unsigned long long __ballot(int pred) { return emul::wave_ballot(pred); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline double rsqrt(double x) { return 1.0 / std::sqrt(x); }
// hardware estimates are only ~single precision: emulate that so the Newton refinement is really exercised
static inline double emul_trunc_mantissa(double v) {
    unsigned long long u;
    std::memcpy(&u, &v, 8);
    u &= ~((1ull << 29) - 1);
    std::memcpy(&v, &u, 8);
    return v;
}
static inline double __builtin_amdgcn_rsq(double x) { return emul_trunc_mantissa(1.0 / std::sqrt(x)); }
static inline double __builtin_amdgcn_rcp(double x) { return emul_trunc_mantissa(1.0 / x); }
static inline double fmin_(double a, double b) { return a < b ? a : b; }
using std::fma;
using std::sqrt;
using std::fabs;
using std::fmax;
using std::fmin;
using std::min;
using std::max;

// ---- runtime API ----------------------------------------------------------------------------
static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emul error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
// every pointer of the emulation is "device" memory (host arrays are used in place)
enum hipMemoryType { hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
struct hipPointerAttribute_t { hipMemoryType type; };
static inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void*) { a->type = hipMemoryTypeDevice; return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = std::calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <typename T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
static inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
enum { hipStreamNonBlocking = 1 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 0, hipDeviceAttributeWallClockRate = 1 };
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int) { *v = a == hipDeviceAttributeWallClockRate ? 1000 : 2; return hipSuccess; }   // tiny "device": persistent grids loop
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
#define hipEventDisableSystemFence 0x20000000
#define hipEventDisableTiming 0x2
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }   // streams run in call order here
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
