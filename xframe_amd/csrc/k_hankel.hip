// Radial Hankel step (rows a1-a3 of SURVEY section 8): the reference's only live GPU kernel,
// OpenCL `apply_weights` (xframe/projects/fxs/projectLibrary/hankel_transforms.py:702-731 midpoint,
// 671-700 trapz):   out[k, lm] = c_l * sum_p W[l(lm)][p][k] * in[p (+1), lm]
// W is the *real* raw weight array (hankel_transforms.py:399-410); the complex prefactor
// c_l = (-/+ i)^l * scale of assemble_weights_mid (426-452) is applied once in the epilogue, so the
// contraction is real-matrix x complex-panel.
#include "mtip_internal.h"
#include <vector>

void launch_hankel_mfma(mtip_ctx* c, const double2* in, double2* out, int inverse);

void launch_hankel(mtip_ctx* c, const double2* in, double2* out, int inverse) {
    ProfScope ps(c, "hankel");
    launch_hankel_mfma(c, in, out, inverse);
}

// out = a - b for shells > 0, out = a for shell 0   (ft_stab add-back folded into coefficient space,
// misk.py:326-329 add_above_zero_index; used by the fused step only)
__global__ void __launch_bounds__(256) k_coeff_diff(const double2* __restrict__ a, const double2* __restrict__ b,
                                                    double2* __restrict__ out, int N, int nlm, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int q = (int)((idx / nlm) % N);
    double2 v = a[idx];
    if (q > 0) v = csub(v, b[idx]);
    out[idx] = v;
}

void launch_coeff_diff(mtip_ctx* c, const double2* a, const double2* b, double2* out) {
    const long long total = (long long)c->B * c->N * c->nlm;
    hipLaunchKernelGGL(k_coeff_diff, dim3((unsigned)div_up(total, 256)), dim3(256), 0, c->stream, a, b, out, c->N,
                       c->nlm, total);
}

// ------------------------------------------------------------------------------------------------------
// Per order l the contraction is a real GEMM  D[k][col] = sum_p W_l[p][k] * X[p][col] with the (batch, m, re/im) axes
// flattened into columns (B (4l+2) of them).  v_mfma_f64_16x16x4_f64: lane j holds A[i = j&15][kk = j>>4] = W_l[p0+kk][k0+i],
// B[kk = j>>4][col = j&15] = X[p0+kk][col0+col], D reg r = D[(j>>4) + 4r][j&15].  The (-/+ i)^l * scale prefactor is applied in
// the epilogue: multiplying by +-i swaps the re/im columns, i.e. neighbouring lanes (shfl_xor 1).
// ---- workgroup-tiled variant ---------------------------------------------------------------------------------
// (rounds 1-2 had one 16-column tile per WAVE, fragments straight from L2: the panel was re-read once per row group and W_l once per column tile, PMC 3.3 x the algorithmic
// bytes), and what bounds these kernels is the rate at which a CU can pull bytes when every CU pulls (~11 B/clk,
// L2 hits included).  Here a 512-thread workgroup owns 80 columns x all (<= 128) output shells of one order -- 218
// workgroups at 8 restarts, L = 32: one per CU, all resident, W_l read once per 80 columns and the panel once.  Per
// chunk of 16 input shells W_l[16][Nq] and the panel [16][80] are staged in LDS with coalesced loads (operands of the
// next two chunks in flight in registers); wave w multiplies row tile w with the five column tiles (one A and five B
// fragments from LDS per k-step).  LDS row strides = 16 doubles mod 32, so the two input shells a half-wave reads
// land on disjoint banks.
#define HT_CT_MAX 5                 // 16-column MFMA tiles per workgroup: template parameter CT in {1, 2, 3, 5}
#define HT_KC 16                    // input shells per chunk (4 MFMA k-steps)
#define HT_ROWS 128                 // output shells per workgroup (8 row tiles = 8 waves)
#define HT_WS (HT_ROWS + 16)        // LDS row strides (doubles)
#define HT_THREADS 512
#define HT_NW (HT_KC * HT_ROWS / HT_THREADS)                       // W doubles per thread and chunk (4)

struct HankelTile32 { int l, cflat0; };

// SUB: out = H(in - in_sub) on output shells > 0 and H(in) on shell 0 in ONE pass (the ft_stab step needs
// IFT(F') - IFT(F) above shell 0 and IFT(F') on it, misk.py:326-329): the staged panel is the difference, and the wave
// that owns output row 0 adds H(in_sub)[0] back with a second MFMA chain whose A fragment is W_l masked to row 0.
template <int HT_CT, bool SUB>
__global__ void __launch_bounds__(HT_THREADS) k_hankel_tile(const double* __restrict__ in, double* __restrict__ out,
                                                            const double* __restrict__ W,
                                                            const HankelTile32* __restrict__ tiles, int N, int Np, int L,
                                                            int B, int poffs, double scale, int sign,
                                                            const double* __restrict__ in_sub, const uint8_t* __restrict__ sub_mask) {
    constexpr int HT_COLS = 16 * HT_CT;
    constexpr int HT_XS = HT_COLS + 16;
    constexpr int HT_NX = (HT_KC * HT_COLS + HT_THREADS - 1) / HT_THREADS;    // panel doubles per thread and chunk
    __shared__ double Ws[2][HT_KC][HT_WS];
    __shared__ double Xs[2][HT_KC][HT_XS];
    __shared__ double Ss[SUB ? 2 : 1][SUB ? HT_KC : 1][HT_XS];   // the subtracted panel itself (row-0 correction)
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const HankelTile32 tinfo = tiles[blockIdx.x];
    const int l = tinfo.l;
    const int k_base = blockIdx.y * HT_ROWS;                      // row block (Nq > 128)
    const int ncl = 4 * l + 2;                                    // doubles per (batch, shell) of this order
    const int nlm2 = 2 * (L + 1) * (L + 1);                       // doubles per (batch, shell)
    const double* Wl = W + (size_t)l * Np * N;
    // ---- staging roles: W chunk = 16 x 128 doubles (row tid / 32, 4 consecutive columns),
    //      panel chunk = 16 x 80 doubles, element e = tid + j * 512 -> (row e / 80, column e % 80)
    const int wrow = tid >> 5, wcol = (tid & 31) * 4;
    int xrow[HT_NX], xc[HT_NX];
    size_t xoff[HT_NX];
    bool xok[HT_NX], xsub[HT_NX];        // xsub: this column's restart takes the subtraction (per-restart ft_stab; all without a mask)
#pragma unroll
    for (int j = 0; j < HT_NX; ++j) {
        const int e = tid + j * HT_THREADS;
        xrow[j] = e / HT_COLS;
        xc[j] = e - xrow[j] * HT_COLS;
        const int cflat = tinfo.cflat0 + xc[j];
        xok[j] = xrow[j] < HT_KC && cflat < B * ncl;
        const int b = xok[j] ? cflat / ncl : 0;
        const int within = xok[j] ? cflat - b * ncl : 0;
        xoff[j] = (size_t)b * N * nlm2 + 2 * (size_t)l * l + within;
        xsub[j] = SUB && (sub_mask == nullptr || sub_mask[b] != 0);
    }
    // two register sets: the operands of the next TWO chunks are in flight while one is multiplied
    double rw0[HT_NW], rx0[HT_NX], rw1[HT_NW], rx1[HT_NX];
    double rs0[SUB ? HT_NX : 1], rs1[SUB ? HT_NX : 1];
    auto request = [&](int p0, double (&rw)[HT_NW], double (&rx)[HT_NX], double (&rs)[SUB ? HT_NX : 1]) {
        const int p = p0 + wrow;
        const bool p_ok = p < Np;
        const double* wr = Wl + (size_t)(p_ok ? p : 0) * N + k_base + wcol;
#pragma unroll
        for (int j = 0; j < HT_NW; ++j) rw[j] = (p_ok && k_base + wcol + j < N) ? wr[j] : 0.0;
#pragma unroll
        for (int j = 0; j < HT_NX; ++j) {
            const bool ok = xok[j] && p0 + xrow[j] < Np;
            const size_t o = xoff[j] + (size_t)(p0 + xrow[j] + poffs) * nlm2;
            rx[j] = ok ? in[o] : 0.0;
            if (SUB) {
                rs[j] = (ok && xsub[j]) ? in_sub[o] : 0.0;
                rx[j] -= rs[j];
            }
        }
    };
    auto deposit = [&](int buf, const double (&rw)[HT_NW], const double (&rx)[HT_NX], const double (&rs)[SUB ? HT_NX : 1]) {
#pragma unroll
        for (int j = 0; j < HT_NW; ++j) Ws[buf][wrow][wcol + j] = rw[j];
#pragma unroll
        for (int j = 0; j < HT_NX; ++j)
            if (xrow[j] < HT_KC) {
                Xs[buf][xrow[j]][xc[j]] = rx[j];
                if (SUB) Ss[buf][xrow[j]][xc[j]] = rs[j];
            }
    };
    // ---- MFMA roles: wave = row tile, all HT_CT column tiles
    const int li = lane & 15, kk = lane >> 4;
    v4f64 acc[HT_CT];
#pragma unroll
    for (int t = 0; t < HT_CT; ++t) acc[t] = v4f64{0.0, 0.0, 0.0, 0.0};
    auto multiply = [&](int buf) {
#pragma unroll
        for (int s = 0; s < HT_KC / 4; ++s) {
            const double af = Ws[buf][4 * s + kk][wave * 16 + li];
#pragma unroll
            for (int t = 0; t < HT_CT; ++t) {
                const double bf = Xs[buf][4 * s + kk][16 * t + li];
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc[t], 0, 0, 0);
            }
            if (SUB && wave == 0 && k_base == 0) {               // wave-uniform: output row 0 lives here
                const double a0 = li == 0 ? af : 0.0;
#pragma unroll
                for (int t = 0; t < HT_CT; ++t) {
                    const double sf = Ss[buf][4 * s + kk][16 * t + li];
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, sf, acc[t], 0, 0, 0);
                }
            }
        }
    };
    const int n_chunks = (Np + HT_KC - 1) / HT_KC;
    request(0, rw0, rx0, rs0);
    deposit(0, rw0, rx0, rs0);
    if (n_chunks > 1) request(HT_KC, rw1, rx1, rs1);
    if (n_chunks > 2) request(2 * HT_KC, rw0, rx0, rs0);
    __syncthreads();
    for (int i = 0; i < n_chunks; i += 2) {
        // chunk i from buffer 0; chunk i+1 (set 1) goes to buffer 1; then set 1 asks for chunk i+3
        multiply(0);
        if (i + 1 < n_chunks) deposit(1, rw1, rx1, rs1);
        __syncthreads();
        if (i + 3 < n_chunks) request((i + 3) * HT_KC, rw1, rx1, rs1);
        if (i + 1 >= n_chunks) break;
        // chunk i+1 from buffer 1; chunk i+2 (set 0) goes to buffer 0; then set 0 asks for chunk i+4
        multiply(1);
        if (i + 2 < n_chunks) deposit(0, rw0, rx0, rs0);
        __syncthreads();
        if (i + 4 < n_chunks) request((i + 4) * HT_KC, rw0, rx0, rs0);
    }
    // epilogue: * scale * (-/+ i)^l, store
    int r = l & 3;
    if (sign < 0) r = (4 - r) & 3;
    const bool is_im = (li & 1) != 0;
#pragma unroll
    for (int t = 0; t < HT_CT; ++t) {
        const int cflat = tinfo.cflat0 + 16 * t + li;
        const bool col_ok = cflat < B * ncl;
        const int b = col_ok ? cflat / ncl : 0;
        const int within = col_ok ? cflat - b * ncl : 0;
        const size_t col_off = (size_t)b * N * nlm2 + 2 * (size_t)l * l + within;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double v = acc[t][j] * scale;
            const double partner = __shfl_xor(v, 1, 64);
            double o;
            if (r == 0) o = v;
            else if (r == 2) o = -v;
            else if (r == 1) o = is_im ? partner : -partner;       // * i : (a+bi) i = -b + a i
            else o = is_im ? -partner : partner;                   // * -i: (a+bi)(-i) = b - a i
            const int k = k_base + wave * 16 + kk + 4 * j;
            if (col_ok && k < N) out[col_off + (size_t)k * nlm2] = o;
        }
    }
}

bool hankel_has_difference(const mtip_ctx* c) { return c->d_htiles32 != nullptr; }

// out = H(in - in_sub) above output shell 0, H(in) on it (in_sub == nullptr: plain transform)
void launch_hankel_mfma_sub(mtip_ctx* c, const double2* in, const double2* in_sub, double2* out, int inverse, const uint8_t* sub_mask) {
    const dim3 grid((unsigned)c->n_htiles32, (unsigned)div_up(c->N, HT_ROWS));
#define HT_LAUNCH(CT, SUBF)                                                                                                  \
    hipLaunchKernelGGL((k_hankel_tile<CT, SUBF>), grid, dim3(HT_THREADS), 0, c->stream, reinterpret_cast<const double*>(in), \
                       reinterpret_cast<double*>(out), (const double*)c->d_W, (const HankelTile32*)c->d_htiles32, c->N,      \
                       c->Np, c->L, c->B, c->cfg.hankel_trapz ? 1 : 0, inverse ? c->inv_scale : c->fwd_scale,                \
                       inverse ? +1 : -1, reinterpret_cast<const double*>(in_sub), sub_mask)
    if (in_sub != nullptr) {
        switch (c->htile_ct) {
            case 1: HT_LAUNCH(1, true); break;
            case 2: HT_LAUNCH(2, true); break;
            case 3: HT_LAUNCH(3, true); break;
            default: HT_LAUNCH(5, true); break;
        }
    } else {
        switch (c->htile_ct) {
            case 1: HT_LAUNCH(1, false); break;
            case 2: HT_LAUNCH(2, false); break;
            case 3: HT_LAUNCH(3, false); break;
            default: HT_LAUNCH(5, false); break;
        }
    }
#undef HT_LAUNCH
}

void launch_hankel_mfma(mtip_ctx* c, const double2* in, double2* out, int inverse) {
    launch_hankel_mfma_sub(c, in, nullptr, out, inverse);
}

int build_hankel_tiles(mtip_ctx* c) {
    // widest workgroup tile that still gives (about) one workgroup per CU: W_l is re-read once per tile
    std::vector<HankelTile32> t32;
    const int cts[4] = {5, 3, 2, 1};
    for (int ci = 0; ci < 4; ++ci) {
        t32.clear();
        c->htile_ct = cts[ci];
        for (int l = c->L; l >= 0; --l) {                   // heavy orders first
            const int ncols = c->B * (4 * l + 2);
            for (int c0 = 0; c0 < ncols; c0 += 16 * cts[ci]) t32.push_back(HankelTile32{l, c0});
        }
        if (c->htile_force > 0 ? cts[ci] == c->htile_force : (int)t32.size() * div_up(c->N, HT_ROWS) * 5 >= c->n_cu * 4) break;
    }
    {
        // XCD-aware order: consecutive workgroup ids go round-robin to the 8 XCDs (one L2 each), so the tiles of one order
        // are dealt to ONE residue class mod 8 and W_l is fetched into one L2 instead of up to eight.  Orders go to the
        // least loaded XCD, heaviest first; a queue that runs dry takes tiles from the tail of the longest one.
        const int X = 8;
        std::vector<std::vector<HankelTile32>> q(X);
        size_t i0 = 0;
        while (i0 < t32.size()) {
            size_t i1 = i0;
            while (i1 < t32.size() && t32[i1].l == t32[i0].l) ++i1;
            int best = 0;
            for (int x = 1; x < X; ++x)
                if (q[x].size() < q[best].size()) best = x;
            q[best].insert(q[best].end(), t32.begin() + i0, t32.begin() + i1);
            i0 = i1;
        }
        std::vector<size_t> head(X, 0);
        std::vector<HankelTile32> order;
        while (order.size() < t32.size())
            for (int x = 0; x < X && order.size() < t32.size(); ++x) {
                if (head[x] < q[x].size()) { order.push_back(q[x][head[x]++]); continue; }
                int longest = 0;
                for (int y = 1; y < X; ++y)
                    if (q[y].size() - head[y] > q[longest].size() - head[longest]) longest = y;
                order.push_back(q[longest].back());
                q[longest].pop_back();
            }
        t32.swap(order);
    }
    c->n_htiles32 = (int)t32.size();
    if (hipMalloc((void**)&c->d_htiles32, t32.size() * sizeof(HankelTile32)) != hipSuccess) return MTIP_ENOMEM;
    (void)mtip_copy(c, c->d_htiles32, t32.data(), t32.size() * sizeof(HankelTile32), hipMemcpyHostToDevice);
    return MTIP_OK;
}
