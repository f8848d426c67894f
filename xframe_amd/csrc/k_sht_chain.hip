// Inverse -> forward spherical-harmonic transform of a shell in ONE kernel (rows a4-a6, a10-a12 of SURVEY section 8).
//
// In a phasing step the grid an inverse transform produces is, three times, the grid the next forward transform reads
// (reconstruct.py:518-528, 576-593 as decoded in SURVEY 3.3):
//     F  = iSHT(Hankel(SHT(rho)))          ->  I_lm  = SHT(|F|^2)                 (square_grid, misk.py:159-168)
//     F' = F sqrt(iSHT(I'_lm) / |F|^2)     ->  SHT(F')  (first half of IFT(F'))   (fxs_Projections.py:899-909)
//     rho_new = real-space update          ->  SHT(rho_new) (first half of the next step's FT(rho))
// With separate kernels each of these grids (16.8 MB per restart at 128 x L32) is written once and read back once, and the
// forward kernel's own phases (row loads, FFTs, Legendre sums) run while nothing else of that shell is in flight.  Here the
// shell stays in the workgroup: the inverse transform's pass structure (k_sht_inv_wide: Legendre synthesis of all rows by
// the in-register three-term recurrence, then per pass of RP rows two register-FFT steps around one LDS transpose, epilogue,
// coalesced store) hands its step-2 registers -- thread (row r, n2) holds the R1 values x[R2 n1 + n2] of its row, the row of
// theta_j and its mirror R2 lanes apart -- straight to the forward transform's phase 1 (mirror fold through one lane
// exchange, R1-point FFTs, twiddle, the SAME transpose buffer, R2-point FFTs, (theta, m) panel, table-driven Legendre sums
// in registers).  The stores of a pass drain while the forward half of that pass computes.  Per shell: three launches and
// three grid re-reads per restart-step less (12 -> 9 launches, 173 -> ~123 MB of HBM traffic at 128 x L32).
// The grid values written are bit-identical to k_sht_inv_wide's; the coefficients differ from k_sht_fwd_pair's only in the
// order of the theta sum (<= 1e-15 relative).
#include "mtip_internal.h"
#include "k_sht_common.h"
#include "k_sht_legendre.h"

// phase stamps (DBG instantiations only): s_memtime into scalar registers at the phase boundaries, no branch and no store
// until the kernel's last instruction -- stamps that branched and stored on the spot made the allocator spill the table registers
// of the last phase (238 of them), which is not what one wants to time
#define CHAIN_STAMP(i)                           \
    if constexpr (DBG) stamp[i] = clock64();
// inside the pass loop: durations of the six segments of a pass, summed over the passes (static slots 4..9)
#define CHAIN_SEG(i)                             \
    if constexpr (DBG) {                         \
        const long long n_ = clock64();          \
        stamp[i] += n_ - tprev;                  \
        tprev = n_;                              \
    }

struct ChainArgs {
    // inverse half (as k_sht_inv_wide, one workgroup per shell)
    const double2* coeff;
    double2* grid;
    const double* P;
    const double2* AB;
    const double* cost;
    const double2* twN_g;
    const double2* Fin;
    const int* slot;
    const uint16_t* mk;                 // packed masks (3, B, Nq, nt, R2): bit n1 = support, bit 8 + n1 = initial support of point R2 n1 + n2
    RealEpi re;
    int npairs, nt, L, Nq, which, B;
    // forward half
    double2* coeff_out;
    const double* PT;
    const int* lmtab;
    const double* PTc;                  // chunk layout of PT and lmtab (CHK instantiations)
    const int* lmc;
    const double* gw;
    double norm;
    long long* dbg;                     // diagnostic: (shells, CHAIN_DBG_SLOTS) clock64 stamps of wave 0 at the phase boundaries, or null
    int gsz, thg;                       // threads per accumulation group (a power of two), theta pairs of the shell per group
};

// EPI: epilogue of the inverse half (EPI_STORE / EPI_MODULUS / EPI_REAL_UPDATE); PRE: prologue of the forward half on the value
// just written (MTIP_PRE_NONE / MTIP_PRE_SQUARE); MAXI: (l, m) pairs per thread of an accumulation group; THG > 0: theta pairs
// per group at compile time, their table rows requested together into registers; THG == 0: run-time count, table values
// loaded where they are used (small grids)
// LC: L_max at compile time (0: run-time value).  With the benchmark's L = 32 the row strides of the spectra / panel become immediates, the
// zero-padded inputs of the inverse 16-point FFTs (7 of 16) and the unused outputs of the forward ones fold away, loop bounds are constants.
// CHK: chunk layout of the Legendre sums (a thread = up to three orders l of one (m, parity): one pair of LDS reads per theta serves all three)
template <int EPI, int PRE, int R1, int R2, int MAXI, int THG, bool DBG, int LC, bool CHK>
__global__ void __launch_bounds__(SW_THREADS) k_sht_chain(ChainArgs a) {
    constexpr int N = R1 * R2;
    constexpr int AS = R2 + 1;
    constexpr int GSR = R1 * AS;                    // panel row stride (>= n_phi + 1 > 2 L + 1)
    HIP_DYNAMIC_SHARED(double2, sm)
    long long stamp[DBG ? MTIP_CHAIN_DBG_SLOTS : 1];
    long long tprev = 0;
    if constexpr (DBG) {
#pragma unroll
        for (int i = 0; i < MTIP_CHAIN_DBG_SLOTS; ++i) stamp[i] = 0;
    }
    const int L = LC > 0 ? LC : a.L, nt = a.nt, npairs = LC > 0 ? (LC + 1) * (LC + 2) / 2 : a.npairs, Nq = a.Nq, B = a.B;
    const int nm = 2 * L + 1;
    const int nlm = (L + 1) * (L + 1);
    double2* twN = sm;                              // N
    double2* Gs = sm + N;                           // nt * nm        spectra: row 2j = theta_j, 2j+1 = its mirror
    double2* ABs = Gs + (size_t)nt * nm;            // npairs         recurrence coefficients
    double2* cl = ABs + npairs;                     // nlm            (Legendre phase)
    double2* Bm = cl;                               // nw * RW * R1 * AS   the waves' transpose buffers (both directions); at the end the groups' partial sums
    const int tid = threadIdx.x;
    const long long shell = blockIdx.x;
    const int q = (int)(shell % Nq);
    const double2* csrc = a.coeff + (size_t)shell * nlm;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
    CHAIN_STAMP(0)
    LegendreStart ls;
    legendre_prefetch_first(ls, a.P, nt, L, nt >> 1, 0, wave, tid & 63);
    for (int e = tid; e < N; e += blockDim.x) sm[e] = a.twN_g[e];
    for (int e = tid; e < npairs; e += blockDim.x) ABs[e] = a.AB[e];
    for (int e = tid; e < nlm; e += blockDim.x) cl[e] = csrc[e];
    long long dst_shell = shell;
    if (a.slot != nullptr && a.which >= 0) dst_shell += (long long)a.slot[(shell / Nq) * SL_N + a.which] * B * Nq;
    double2* gdst = a.grid + (size_t)dst_shell * nt * N;
    const double2* fsrc = a.Fin ? a.Fin + (size_t)shell * nt * N : nullptr;
    const double2* rprev = nullptr;
    const uint16_t* rmk = nullptr;
    double err_num = 0.0, err_den = 0.0;
    if (EPI == EPI_REAL_UPDATE) {
        const int bb = (int)(shell / Nq);
        const int* sl = a.slot + bb * SL_N;
        const size_t gsh = (size_t)nt * N;
        rprev = a.re.prev + ((size_t)sl[SL_CUR] * B * Nq + shell) * gsh;
        gdst = a.re.out + ((size_t)sl[SL_OUT] * B * Nq + shell) * gsh;
        rmk = a.mk + ((size_t)sl[SL_SUP] * B * Nq + shell) * nt * R2;
    }
    // forward half: its (theta, m) panel rows overwrite the spectra rows a pass has consumed (same row order: 2j = even part,
    // 2j+1 = odd part of theta pair j), and ONE Legendre-sum phase follows the last pass: accumulation group g owns theta pairs
    // [g thg, (g + 1) thg) of the shell, thread t of it the (l, m) pairs t + u gsz.  (Sums per pass were built first: table
    // rows and accumulators live across the 16-point FFTs spill.)
    const int gsz = a.gsz;
    const int thg = THG > 0 ? THG : a.thg;
    const int grp = __builtin_amdgcn_readfirstlane(tid / gsz);      // wave-uniform (gsz >= 64): table rows through scalar bases
    const int tg = tid - grp * gsz;
    // ---- the FFT steps: WAVE-PRIVATE row groups.  A wave owns RW = 64 / R2 whole rows (theta pairs with their mirrors) through all
    // four FFT steps -- inverse R2-point FFTs over k2, transpose, inverse R1-point FFTs + epilogue + store, mirror fold, forward
    // R1-point FFTs, transpose, forward R2-point FFTs -- and a private slice of the transpose buffer, so nothing between the
    // Legendre synthesis and the Legendre sums needs a workgroup barrier: LDS operations of one wave complete in order.  (With
    // workgroup-wide passes of 32 rows, eight barriers per shell kept the eight waves in lock step: one resource at a time --
    // LDS, vector ALU, memory pipe -- and a fifth of the workgroup's lifetime waiting for the slowest wave,
    // profiles/r04_chain_phase_timers.txt.)  Lane roles: steps 2 / forward 1: (row, n2); steps 1 / forward 2: (half, row, k1)
    // where the 16-point transforms are split over two lanes (half_fft).
    constexpr bool SPLIT = R2 == 16;
    constexpr int RW = 64 / R2;                             // rows per group
    constexpr int S1 = RW * R1;                             // lanes of one half in steps 1 / forward 2
    static_assert(S1 * (SPLIT ? 2 : 1) <= 64, "lane roles");
    const int lane = tid & 63;
    const int n_grp = (nt + RW - 1) / RW;
    double2* Bw = Bm + (size_t)wave * (RW * R1 * AS);       // this wave's transpose buffer
    const int r2l = lane / R2, n2 = lane - r2l * R2;
    const int hf = SPLIT ? lane / S1 : 0;
    const int t1 = lane - hf * S1;
    const bool role1 = SPLIT || lane < S1;
    const int r1l = (t1 / R1) % RW, k1 = t1 % R1;
    // epilogue operands of a lane's step-2 outputs in group g (previous density / F: R1 values, the packed masks: one load),
    // requested a group ahead: those of the wave's first group before the Legendre synthesis (which touches no global memory),
    // those of the next group as soon as the epilogue has consumed its own
    constexpr bool HAS_PRE = EPI == EPI_MODULUS || EPI == EPI_REAL_UPDATE;
    double2 pre[HAS_PRE ? R1 : 1];
    unsigned pre_m = 0;
    double pre_w = 0.0;                                     // theta weight of the error integral of this lane's row
    auto load_pre = [&](int g) {
        const int rr = g * RW + r2l;
        if (HAS_PRE && g < n_grp && rr < nt) {
            const int th = rr >> 1;
            const int row = (rr & 1) ? (nt - 1 - th) : th;
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) {
                const size_t o = (size_t)row * N + R2 * n1 + n2;
                if (EPI == EPI_MODULUS) pre[n1] = fsrc[o];
                if (EPI == EPI_REAL_UPDATE) pre[n1] = rprev[o];
            }
            if (EPI == EPI_REAL_UPDATE) {
                pre_m = rmk[row * R2 + n2];
                pre_w = a.re.wt[row];
            }
        }
    };
    // (l, m) of this thread's pairs in the Legendre sums: loaded here, used 40 k cycles later (a global round trip at that point
    // was most of the closing loop)
    int my_lm[MAXI];
#pragma unroll
    for (int u = 0; u < MAXI; ++u) my_lm[u] = CHK ? a.lmc[u * 256 + tg] : a.lmtab[min(tg + u * gsz, npairs - 1)];
    __syncthreads();
    CHAIN_STAMP(1)
    // (behind the staging copies: vmcnt counts in order, requested before them these loads made the copies wait)
    load_pre(wave);
    const RealFlags rflags = real_flags(a.re.rp, a.re.method);
    const bool add_prev = EPI == EPI_REAL_UPDATE && a.re.add_prev && q > 0 && (a.re.add_mask == nullptr || a.re.add_mask[shell / Nq] != 0);
    const double wr_q = EPI == EPI_REAL_UPDATE ? a.re.wr[q] : 0.0;
    // ---- Legendre synthesis of every row (k_sht_legendre.h)
    legendre_synthesis_rows(ls, Gs, cl, ABs, a.P, a.cost, nt, L, nt >> 1, 0, wave, nw, tid & 63);
    CHAIN_STAMP(2)
    __syncthreads();
    CHAIN_STAMP(3)
    if constexpr (DBG) tprev = stamp[3];
    for (int g = wave; g < n_grp; g += nw) {
        // ---- step 1: inverse R2-point FFTs over k2 of the zero padded spectrum, twiddle, transpose store
        {
            const int rr = g * RW + r1l;
            if (role1 && rr < nt) {
                const double2* gr = Gs + (size_t)rr * nm + L;
                double2 uv[R2];
#pragma unroll
                for (int k2 = 0; k2 < R2; ++k2) {
                    // branch-free (clamped index + select): a `cond ? load : 0` compiles to a branch with a full wait at its
                    // join, sixteen LDS round trips one after the other
                    const int k = k1 + R1 * k2;
                    const bool lo = k <= L, hi = k >= N - L;
                    const double2 v = gr[lo ? k : (hi ? k - N : 0)];
                    uv[k2] = (lo || hi) ? v : make_double2(0.0, 0.0);
                }
                double2* br = Bw + (size_t)(r1l * R1 + k1) * AS;
                if constexpr (SPLIT) {
                    double2 yv[R2 / 2];
                    half_fft<R2, true>(uv, hf, yv);
#pragma unroll
                    for (int j = 0; j < R2 / 2; ++j) {
                        const int m2 = 2 * j + hf;
                        double2 w = twN[m2 * k1];
                        w.y = -w.y;
                        br[m2] = cmul(yv[j], w);
                    }
                } else {
                    SmallFFT<R2, true>::run(uv);
#pragma unroll
                    for (int m2 = 0; m2 < R2; ++m2) {
                        double2 w = twN[m2 * k1];
                        w.y = -w.y;
                        br[m2] = cmul(uv[m2], w);
                    }
                }
            }
        }
        MTIP_WAVE_LDS_SYNC();
        CHAIN_SEG(4)
        // ---- step 2: inverse R1-point FFTs over k1, epilogue + coalesced store, the forward half's prologue
        const int rr2 = g * RW + r2l;
        const bool act2 = rr2 < nt;
        double2 vv[R1];
#pragma unroll
        for (int q1 = 0; q1 < R1; ++q1) vv[q1] = make_double2(0.0, 0.0);
        if (act2) {
            const double2* br = Bw + (size_t)r2l * R1 * AS + n2;
#pragma unroll
            for (int q1 = 0; q1 < R1; ++q1) vv[q1] = br[q1 * AS];
            SmallFFT<R1, true>::run(vv);
            const int th = rr2 >> 1;
            const int row = (rr2 & 1) ? (nt - 1 - th) : th;
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) {
                double2 v = vv[n1];
                const size_t o = (size_t)row * N + R2 * n1 + n2;
                if (EPI == EPI_MODULUS) {
                    // project_to_modified_intensity, fxs_Projections.py:899-909
                    const double2 Fv = pre[n1];
                    const double I = cabs2(Fv);
                    const bool ok = (I >= 0.0) && (v.x >= 0.0);
                    const double mult = ok ? sqrt(v.x / I) : 0.0;
                    v = cscale(Fv, mult);
                } else if (EPI == EPI_REAL_UPDATE) {
                    const double2 pv = pre[n1];
                    const double2 w = add_prev ? cadd(v, pv) : v;
                    double2 Pj;
                    v = real_update_point_flat(a.re.rp, rflags, a.re.beta, w, pv, ((pre_m >> n1) & 1u) != 0, Pj);
                    if (!a.re.err_use_mask || ((pre_m >> (8 + n1)) & 1u)) {   // l2_projection_diff, fxs_IO_methods.py:97-128
                        const double wg = wr_q * pre_w;
                        const double dx = w.x - Pj.x, dy = w.y - Pj.y;
                        err_num = fma(wg, dx * dx + dy * dy, err_num);
                        err_den = fma(wg, w.x * w.x + w.y * w.y, err_den);
                    }
                }
                gdst[o] = v;
                if (PRE == MTIP_PRE_SQUARE) v = make_double2(cabs2(v), 0.0);       // square_grid, misk.py:159-168
                vv[n1] = v;
            }
        }
        load_pre(g + nw);
        // ---- forward phase 1: mirror fold (row 2j = theta_j, 2j+1 = its mirror: R2 lanes apart, same wave), R1-point FFTs, twiddle
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) {
            const double px = __shfl_xor(vv[n1].x, R2, 64), py = __shfl_xor(vv[n1].y, R2, 64);
            vv[n1] = (r2l & 1) ? make_double2(px - vv[n1].x, py - vv[n1].y)      // odd part  north - south
                               : make_double2(vv[n1].x + px, vv[n1].y + py);    // even part north + south
        }
        if (act2) {
            SmallFFT<R1, false>::run(vv);
            double2* ar = Bw + (size_t)r2l * R1 * AS + n2;
#pragma unroll
            for (int q1 = 0; q1 < R1; ++q1) ar[q1 * AS] = cmul(vv[q1], twN[n2 * q1]);
        }
        MTIP_WAVE_LDS_SYNC();
        CHAIN_SEG(6)
        // ---- forward phase 2: R2-point FFTs over n2; keep |m| <= L, Gauss weight; the panel rows go where this group's spectra
        //      were (consumed by step 1 of the same wave)
        {
            const int rr = g * RW + r1l;
            if (role1 && rr < nt) {
                double2 uv[R2];
                const double2* ar = Bw + (size_t)(r1l * R1 + k1) * AS;
#pragma unroll
                for (int qq = 0; qq < R2; ++qq) uv[qq] = ar[qq];
                const double sc = a.gw[rr >> 1] * a.norm;
                double2* gr = Gs + (size_t)rr * nm + L;
                if constexpr (SPLIT) {
                    double2 yv[R2 / 2];
                    half_fft<R2, false>(uv, hf, yv);
#pragma unroll
                    for (int j = 0; j < R2 / 2; ++j) {
                        const int k = k1 + R1 * (2 * j + hf);
                        if (k <= L) gr[k] = cscale(yv[j], sc);
                        else if (k >= N - L) gr[k - N] = cscale(yv[j], sc);
                    }
                } else {
                    SmallFFT<R2, false>::run(uv);
#pragma unroll
                    for (int k2 = 0; k2 < R2; ++k2) {
                        const int k = k1 + R1 * k2;
                        if (k <= L) gr[k] = cscale(uv[k2], sc);
                        else if (k >= N - L) gr[k - N] = cscale(uv[k2], sc);
                    }
                }
            }
        }
        MTIP_WAVE_LDS_SYNC();
        CHAIN_SEG(8)
    }
    __syncthreads();                                    // the panel of the shell is complete
    CHAIN_SEG(9)
    // the twiddles are dead: the per-shell error sums go to the head of the LDS block
    if (EPI == EPI_REAL_UPDATE) {
        // fixed order (bitwise reproducible): wave butterflies, then the waves
        for (int o = 32; o > 0; o >>= 1) {
            err_num += __shfl_xor(err_num, o, 64);
            err_den += __shfl_xor(err_den, o, 64);
        }
        double* red = reinterpret_cast<double*>(sm);
        if ((tid & 63) == 0) {
            red[2 * (tid >> 6)] = err_num;
            red[2 * (tid >> 6) + 1] = err_den;
        }
    }
    // ---- Legendre sums over the whole panel: c_lm += P_lm(theta_j) E/O[j][m], table rows of the group requested together
    double2 accp[MAXI], accm[MAXI];
    {
        // table rows of this thread's (l, m) pairs and theta pairs, requested together.  (Requested before the passes instead --
        // they fit beside them since the 16-point FFTs are split -- the sums get 4 k cycles shorter and step 1 of the passes 4 k
        // longer: measured, workgroup lifetime unchanged, profiles/r04_chain_phase_timers.txt.)
        double tab[MAXI][THG > 0 ? THG : 1];
        if (THG > 0) {
#pragma unroll
            for (int u = 0; u < MAXI; ++u)
#pragma unroll
                for (int jj = 0; jj < (THG > 0 ? THG : 1); ++jj) {
                    if constexpr (CHK) {
                        tab[u][jj] = a.PTc[(size_t)(grp * THG + jj) * 768 + u * 256 + tg];
                    } else {
                        const double* row = a.PT + (size_t)(grp * THG + jj) * npairs;
                        tab[u][jj] = row[(unsigned)min(tg + u * gsz, npairs - 1)];   // clamped: branch-free loads
                    }
                }
        }
        const double2 *srcp[MAXI], *srcm[MAXI];
#pragma unroll
        for (int u = 0; u < MAXI; ++u) {
            const int lm = CHK ? max(my_lm[0], 0) : my_lm[u];                 // (CHK: the chunk's orders share m and the parity of l - m)
            const int l = lm & 0xff, m = lm >> 8;
            srcp[u] = Gs + (size_t)(((l + m) & 1) + 2 * grp * thg) * nm + L + m;
            srcm[u] = srcp[u] - 2 * m;
            accp[u] = make_double2(0.0, 0.0);
            accm[u] = make_double2(0.0, 0.0);
        }
        if (THG > 0) {
            // four theta pairs at a time, fenced: left alone the scheduler requests every panel value of the phase first (96 LDS
            // reads in flight next to the table registers) and spills
#pragma unroll
            for (int jc = 0; jc < (THG > 0 ? THG : 1); jc += 4) {
                if constexpr (CHK) {
#pragma unroll
                    for (int jj = jc; jj < jc + 4 && jj < (THG > 0 ? THG : 1); ++jj) {
                        const double2 vp = srcp[0][2 * jj * nm];
                        const double2 vm = srcm[0][2 * jj * nm];
#pragma unroll
                        for (int u = 0; u < MAXI; ++u) {
                            const double p = tab[u][jj];
                            accp[u].x = fma(p, vp.x, accp[u].x);
                            accp[u].y = fma(p, vp.y, accp[u].y);
                            accm[u].x = fma(p, vm.x, accm[u].x);
                            accm[u].y = fma(p, vm.y, accm[u].y);
                        }
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < MAXI; ++u)
#pragma unroll
                        for (int jj = jc; jj < jc + 4 && jj < (THG > 0 ? THG : 1); ++jj) {
                            const double p = tab[u][jj];
                            const double2 vp = srcp[u][2 * jj * nm];
                            const double2 vm = srcm[u][2 * jj * nm];
                            accp[u].x = fma(p, vp.x, accp[u].x);
                            accp[u].y = fma(p, vp.y, accp[u].y);
                            accm[u].x = fma(p, vm.x, accm[u].x);
                            accm[u].y = fma(p, vm.y, accm[u].y);
                        }
                }
                asm volatile("" ::: "memory");
            }
        } else {
#pragma unroll
            for (int u = 0; u < MAXI; ++u) {
                const double* pt = a.PT + (size_t)(grp * thg) * npairs + min(tg + u * gsz, npairs - 1);
                double2 ap = accp[u], am = accm[u];
                for (int jj = 0; jj < thg; ++jj) {
                    const double p = pt[(size_t)jj * npairs];
                    const double2 vp = srcp[u][2 * jj * nm];
                    const double2 vm = srcm[u][2 * jj * nm];
                    ap.x = fma(p, vp.x, ap.x);
                    ap.y = fma(p, vp.y, ap.y);
                    am.x = fma(p, vm.x, am.x);
                    am.y = fma(p, vm.y, am.y);
                }
                accp[u] = ap;
                accm[u] = am;
            }
        }
    }
    CHAIN_STAMP(16)
    // the other groups' partial coefficients go where the transpose buffers were (other waves may still be reading the panel);
    // group 0 adds them to its own in a fixed order and stores
    double2* racc = Bm;                                 // (groups - 1, MAXI, 2, gsz)
    if (grp > 0) {
#pragma unroll
        for (int u = 0; u < MAXI; ++u) {
            racc[(size_t)(((grp - 1) * MAXI + u) * 2) * gsz + tg] = accp[u];
            racc[(size_t)(((grp - 1) * MAXI + u) * 2 + 1) * gsz + tg] = accm[u];
        }
    }
    __syncthreads();
    if (EPI == EPI_REAL_UPDATE && tid == 0) {
        const double* red = reinterpret_cast<const double*>(sm);
        double sn = 0.0, sd = 0.0;
        for (int wv = 0; wv < nw; ++wv) {
            sn += red[2 * wv];
            sd += red[2 * wv + 1];
        }
        a.re.partial[(size_t)shell * 2] = sn;
        a.re.partial[(size_t)shell * 2 + 1] = sd;
    }
    if (grp == 0) {
        const int ngrp = blockDim.x / gsz;
        double2* cdst = a.coeff_out + (size_t)shell * nlm;
#pragma unroll
        for (int u = 0; u < MAXI; ++u) {
            double2 sp = accp[u], sq = accm[u];
            for (int g = 1; g < ngrp; ++g) {
                const double2 vp = racc[(size_t)(((g - 1) * MAXI + u) * 2) * gsz + tg];
                const double2 vq = racc[(size_t)(((g - 1) * MAXI + u) * 2 + 1) * gsz + tg];
                sp.x += vp.x; sp.y += vp.y;
                sq.x += vq.x; sq.y += vq.y;
            }
            if (CHK ? my_lm[u] >= 0 : tg + u * gsz < npairs) {
                const int l = my_lm[u] & 0xff, m = my_lm[u] >> 8;
                cdst[l * (l + 1) + m] = sp;
                if (m > 0) cdst[l * (l + 1) - m] = (m & 1) ? make_double2(-sq.x, -sq.y) : sq;
            }
        }
    }
    CHAIN_STAMP(17)
    if constexpr (DBG) {
        if (a.dbg != nullptr && tid == 0) {
#pragma unroll
            for (int i = 0; i < MTIP_CHAIN_DBG_SLOTS; ++i) a.dbg[(size_t)shell * MTIP_CHAIN_DBG_SLOTS + i] = stamp[i];
        }
    }
}

// ------------------------------------------------------------------------------------------------------
struct ChainGeom {
    int r1 = 0, r2 = 0, gsz = 0, thg = 0, maxi = 0;     // maxi: the kernel's MAXI (>= pairs per thread)
    bool reg_tab = false;
    size_t tb = 0;                      // double2 of the waves' transpose buffers
    size_t lds = 0;
    bool ok = false;
};

static ChainGeom chain_geom(const mtip_ctx* c) {
    ChainGeom g;
    if (!sht_reg_supported(c) || !c->sht_wide || c->d_AB == nullptr || c->d_PT == nullptr || c->d_lmtab == nullptr) return g;
    if (!reg_radices(c->np, &g.r1, &g.r2) || (c->nt & 1)) return g;
    // every wave owns 64 / R2 rows at a time and a transpose buffer for them; it aliases the coefficient block
    g.tb = (size_t)(SW_THREADS / 64) * (64 / g.r2) * g.r1 * (g.r2 + 1);
    const size_t fixed = (size_t)c->np + (size_t)c->nt * c->nm + c->npairs;
    const int TP = c->nt / 2;                       // theta pairs of a shell
    // accumulation groups of the Legendre-sum phase: the smallest power-of-two group whose threads hold <= 3 (l, m) pairs each
    // and whose count divides the theta pairs
    for (int gsz = 64; gsz <= SW_THREADS; gsz *= 2) {
        const int ngrp = SW_THREADS / gsz;
        if (TP % ngrp != 0) continue;
        const int maxi = div_up(c->npairs, gsz);
        if (maxi > 3) continue;
        g.gsz = gsz;
        g.thg = TP / ngrp;
        // the instantiation launch_chain_r picks: table rows in registers for the 128-point grids, else the run-time variants
        g.reg_tab = c->np == 128 && g.thg == 16 && maxi >= 2;
        g.maxi = g.reg_tab ? maxi : (maxi == 1 ? 1 : 3);
        break;
    }
    if (g.gsz == 0) return g;
    // transpose buffer / panel staging aliases the coefficient block; the groups' partial sums land there at the end
    const size_t un = std::max(std::max((size_t)c->nlm, g.tb), (size_t)SW_THREADS * g.maxi * 2);
    g.lds = (fixed + un) * sizeof(double2);
    g.ok = g.lds <= 158 * 1024;
    return g;
}

bool sht_chain_supported(const mtip_ctx* c) {
    return c->sht_chain && chain_geom(c).ok;
}

template <int EPI, int PRE, int R1, int R2>
static void launch_chain_r(mtip_ctx* c, const ChainGeom& g, const ChainArgs& a) {
    const dim3 gr((unsigned)(c->B * c->N)), bl(SW_THREADS);
#define CHAIN_GO(MAXI, THG) hipLaunchKernelGGL((k_sht_chain<EPI, PRE, R1, R2, MAXI, THG, false, 0, false>), gr, bl, g.lds, c->stream, a)
    if constexpr (R1 * R2 == 128) {
        const bool metric_grid = g.reg_tab && g.maxi == 3 && c->L == 32 && c->nt == 64 && c->chain_chunks > 0;
        if (metric_grid && a.dbg != nullptr) {                   // phase stamps: the metric's grid only
            hipLaunchKernelGGL((k_sht_chain<EPI, PRE, R1, R2, 3, 16, true, 32, true>), gr, bl, g.lds, c->stream, a);
            return;
        }
        if (metric_grid && c->sht_chain_lc) {                   // the metric's grid: L at compile time, chunk layout of the sums
            hipLaunchKernelGGL((k_sht_chain<EPI, PRE, R1, R2, 3, 16, false, 32, true>), gr, bl, g.lds, c->stream, a);
            return;
        }
        if (g.reg_tab && g.maxi == 3) { CHAIN_GO(3, 16); return; }
        if (g.reg_tab && g.maxi == 2) { CHAIN_GO(2, 16); return; }
    }
    if constexpr (R1 * R2 < 256) {                  // a 256-point shell does not fit one CU (chain_geom)
        if (g.maxi == 1) CHAIN_GO(1, 0);
        else CHAIN_GO(3, 0);
    }
#undef CHAIN_GO
}

template <int EPI, int PRE>
static void launch_chain_p(mtip_ctx* c, const ChainGeom& g, const ChainArgs& a) {
    switch (c->np) {
        case 16: launch_chain_r<EPI, PRE, 4, 4>(c, g, a); break;
        case 32: launch_chain_r<EPI, PRE, 4, 8>(c, g, a); break;
        case 64: launch_chain_r<EPI, PRE, 8, 8>(c, g, a); break;
        case 128: launch_chain_r<EPI, PRE, 8, 16>(c, g, a); break;
        default: launch_chain_r<EPI, PRE, 16, 16>(c, g, a); break;
    }
}

// grid = epilogue(iSHT(coeff)), coeff_out = SHT(prologue(grid)).  Epilogues: EPI_STORE (prologue |.|^2: the F -> I_lm link of a
// step), EPI_MODULUS and EPI_REAL_UPDATE (no prologue).  Caller checks sht_chain_supported().
void launch_sht_chain(mtip_ctx* c, const double2* coeff, double2* grid, const InvEpilogue& epi, int prologue, double2* coeff_out) {
    ProfScope ps(c, epi.mode == EPI_REAL_UPDATE ? "sht_chain_real" : epi.mode == EPI_MODULUS ? "sht_chain_modulus" : "sht_chain");
    const ChainGeom g = chain_geom(c);
    ChainArgs a;
    a.coeff = coeff;
    a.grid = grid;
    a.P = c->d_P;
    a.AB = c->d_AB;
    a.cost = c->d_cost;
    a.twN_g = c->d_twN;
    a.Fin = epi.F;
    a.slot = (epi.out_slot >= 0 || epi.mode == EPI_REAL_UPDATE) ? c->d_slot : nullptr;
    a.mk = c->d_mk;
    a.re = epi.real;
    a.npairs = c->npairs;
    a.nt = c->nt;
    a.L = c->L;
    a.Nq = c->N;
    a.which = epi.out_slot;
    a.B = c->B;
    a.coeff_out = coeff_out;
    a.PT = c->d_PT;
    a.lmtab = c->d_lmtab;
    a.PTc = c->d_PTc;
    a.lmc = c->d_lmc;
    a.gw = c->d_gw;
    a.norm = 2.0 * 3.14159265358979323846 / c->np;
    a.gsz = g.gsz;
    a.thg = g.thg;
    a.dbg = c->d_chain_dbg ? c->d_chain_dbg + (size_t)(epi.mode == EPI_REAL_UPDATE ? 2 : epi.mode == EPI_MODULUS ? 1 : 0) * c->B * c->N * MTIP_CHAIN_DBG_SLOTS
                           : nullptr;
    if (epi.mode == EPI_REAL_UPDATE) launch_chain_p<EPI_REAL_UPDATE, MTIP_PRE_NONE>(c, g, a);
    else if (epi.mode == EPI_MODULUS) launch_chain_p<EPI_MODULUS, MTIP_PRE_NONE>(c, g, a);
    else if (prologue == MTIP_PRE_SQUARE) launch_chain_p<EPI_STORE, MTIP_PRE_SQUARE>(c, g, a);
    else launch_chain_p<EPI_STORE, MTIP_PRE_NONE>(c, g, a);
}
