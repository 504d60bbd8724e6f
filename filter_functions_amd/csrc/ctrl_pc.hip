// ctrl_pc.hip -- K3p: the control-matrix accumulation for small d with SPECIALISED wavefronts.
// Same mathematics, inputs, output layout and per-lane arithmetic as ctrl.hip (one frequency per
// lane, 64 per wave; Y_a(w) = sum_g T_g^dag [Bbar_a o E_g(w)] T_g), different division of labour:
// every sub-chunk of a block consists of ONE producer wavefront, which generates the whole integral
// tile e^{i w t_g} I^(g)(w) of the next segment (and moves the segment's operands and table rows
// from global memory into LDS), and NC consumer wavefronts, one per noise operator, which do
// nothing but the two d x d products on the tile of the current segment.  In ctrl.hip every wave
// alternates between the two phases and all waves of a block are in the same phase at the same
// time; the latency-bound generation and the FMA-bound contraction never overlap there
// (profiles/r01_c_*: generation alone 43 us, contraction alone 80 us, together 111 us).  Here they
// run side by side on every SIMD.  Producers sit on different SIMDs (sub-chunk s -> wave s of its
// group): with four sub-chunks of one producer and three consumers every SIMD hosts exactly one
// producer and three consumers, and with 13 entries the producer's ~555 instructions per segment
// match a consumer's 576.
// The consumers take T_g through SCALAR loads (wave-uniform addresses, issued one segment ahead at
// the end of the previous one): 32 doubles in SGPRs feed v_fma_f64 directly instead of occupying
// 64 VGPRs, which brings the kernel to 104 VGPRs, i.e. four wavefronts per SIMD.  (The first
// version of ctrl.hip fed ALL operands from scalar loads issued right before their use and spent
// 60 % of its time waiting for them; here the loads have a whole barrier interval to land.)
// Measured at config 2 on one box: symmetric kernel 108.4 us, 3 sub-chunks with T_g from LDS
// 104.5 us (166 VGPRs), 3 sub-chunks scalar T_g 106.9 us, 4 sub-chunks scalar T_g 100.8 us.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "ffk_internal.h"
#include "ffk_mfma_util.h"

namespace ffk {
namespace {

#ifndef FFK_PC_WFOLD            /* 1: consumers contract with Bbar_mn T_nj from LDS (no Bbar o E pass) */
#define FFK_PC_WFOLD 0
#endif
#ifndef FFK_PC_PRIO_PRODUCER   /* 0..3; tuning builds override */
#define FFK_PC_PRIO_PRODUCER 1
#endif
#if defined(FFK_PC_SUB)       /* tuning builds */
constexpr int kPcSub = FFK_PC_SUB;
#else
constexpr int kPcSub = 4;   // sub-chunks per block: 4 x (1 + 3) = 16 waves = 4 per SIMD
#endif

template <int D>
struct PcEntries {          // every entry once, the (coinciding) diagonal entries as slot 0
    static constexpr int build(int* out) {
        int n = 0;
        for (int e = 0; e < D*D; ++e) {
            if (e != 0 && e / D == e % D) continue;
            if (out) out[n] = e;
            ++n;
        }
        return n;
    }
    static constexpr int count = build(nullptr);
    struct Slots {
        int v[D*D];
    };
    static constexpr Slots make() {
        Slots s{};
        build(s.v);
        return s;
    }
    static constexpr Slots slots = make();
    // position of entry e = m*D + n in the tile: the list index (diagonal entries: 0)
    static constexpr Slots make_table() {
        Slots t{};
        for (int e = 0; e < D*D; ++e) {
            int k = 0;
            for (int x = 0; x < e; ++x)
                if (x == 0 || x / D != x % D) ++k;
            t.v[e] = (e / D == e % D) ? 0 : k;
        }
        return t;
    }
    static constexpr Slots slot_table = make_table();
};

// MF != 0: the consumers contract on the FP64 matrix cores (v_mfma_f64_4x4x4_4b) instead of
// v_fma_f64.  MF = 1: 16 frequencies on the instruction's columns, four groups per wavefront, a
// 4 x 4 transpose across lanes between the two products (layout in ffk_mfma_util.h).  MF = 2: one
// frequency per 4 x 4 x 4 block, the first product's result is the second's A operand as it stands.
// The integral tile's slots are TS complex apart: 64 frequencies (+ 4 of padding for MF = 2, whose
// lanes read 16 slots x 4 frequencies at once).
constexpr int pc_tile_stride(int mf) { return mf == 2 ? 68 : 64; }
template <int D, int NC, int MF = 0>
__global__ __launch_bounds__((NC + 1)*kPcSub*64, kPcSub) void ctrl_accumulate_pc_kernel(
    const double* __restrict__ omega, int W, const double* __restrict__ segtab,
    const cplx* __restrict__ ops, int G, int A, int chunk_len, cplx* __restrict__ Ypart) {
    constexpr int GS = kPcSub, NWS = NC + 1;
    constexpr int S = seg_stride(D), DD = D*D;
    // cplx per integral tile: the D (D - 1) + 1 distinct entries only.  (With all D^2 slots the block
    // held 140.5 KiB of LDS; a kernel of another pass needing more than ~8 KiB -- the scan: 9.5 KiB,
    // the prologue: 17.5 KiB -- could then not be placed beside it and waited the whole 83 us for it
    // to retire, tools/corun.hip and profiles/r02_q_*.)
    constexpr int TS = pc_tile_stride(MF);
    constexpr int TILE = PcEntries<D>::count*TS;
#if FFK_PC_WFOLD
    constexpr int OPS = DD + NC*D*DD;                 // T_g | W_a[m][n][j] = Bbar_a[m][n] T_g[n][j]
#else
    constexpr int OPS = (1 + NC)*DD;                  // T_g | Bbar_0 ..
#endif
    constexpr int BUF = TILE + OPS;                   // cplx per buffer: tile | operands
    constexpr int SUB = 2*BUF + S;                    // cplx per sub-chunk: 2 buffers | 2 table rows
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

    const int lane = threadIdx.x & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = wave_all / NWS, wl = wave_all % NWS;
    const int pw = sub % NWS;                         // producer's position inside the group
    const bool producer = wl == pw;
    const int cidx = wl < pw ? wl : wl - 1;           // consumer index 0..NC-1
    const int alpha0 = blockIdx.y*NC;
    const int alpha = alpha0 + cidx;
    const bool active = !producer && alpha < A;
    const int n_alpha = min(NC, A - alpha0);
    cplx* lds = reinterpret_cast<cplx*>(lds_raw) + static_cast<size_t>(sub)*SUB;
    double* rows = reinterpret_cast<double*>(lds + 2*BUF);
    const int iw = blockIdx.x*64 + lane;
    const double om = omega[iw < W ? iw : W - 1];
    const int sub_len = (chunk_len + GS - 1)/GS;
    const int g0 = blockIdx.z*chunk_len + sub*sub_len;
    const int g1 = min(min(G, static_cast<int>(blockIdx.z + 1)*chunk_len), g0 + sub_len);

    // Tile of the FIRST segment: generated by all waves of the sub-chunk together (every NWS-th entry
    // each, the per-frequency trigonometry recomputed by each wave), after the producer has staged
    // the table rows.  Left to the producer alone, its dependent chains kept the twelve consumer
    // waves of a block waiting ~2.5 us of the block's ~83 us.
    auto generate_first_share = [&]() {
        const double* st = rows;                          // slot 0 = row g0
        cplx* tile = lds + lane;                          // buffer 0
        const double dtg = st[0];
        cplx ph;
        sincos_pi<true>(om*st[1], &ph.im, &ph.re);
        double sa, ca;
        sincos_pi<true>(0.5*(om*dtg), &sa, &ca);
        constexpr auto& sl = PcEntries<D>::slots;
        const PhasedFrequency pf = phased_frequency(om, dtg, ph, sa, ca);
        for (int k = wl; k < PcEntries<D>::count; k += NWS) {
            const int e = sl.v[k];
            const double* r = st + seg_rec(e);
            tile[k*TS] = phased_integral_aa(pf, r[0], r[1], r[2]);
        }
    };

    if (producer) {
        // ---- producer: operands + table rows -> LDS, integral tile of the next segment ----------
        // Static issue priority: the producer's work is a chain of dependent operations (argument
        // reduction -> polynomial -> reciprocal -> entries); at equal priority it competes with three
        // consumers' independent FMAs for every issue slot, finishes last and all 16 waves wait for it
        // at the barrier.  One s_setprio before the loop, no per-segment flips: accumulate 92.0 ->
        // 86.2 us at config 2 (priority 3: the same; consumers at priority 1 instead: 94.6 us).
        __builtin_amdgcn_s_setprio(FFK_PC_PRIO_PRODUCER);
        const int n_ops = (1 + n_alpha)*DD;           // <= 64: one element per lane
        auto load_ops = [&](int g) -> cplx {
            const cplx* src = ops + static_cast<size_t>(g)*(1 + A)*DD;
            return lane < n_ops ? src[lane < DD ? lane : lane + alpha0*DD] : cplx{0.0, 0.0};
        };
        // operands of one segment into LDS; lane l holds element l of [T | Bbar_0 | Bbar_1 ..]
        auto store_ops = [&](cplx* dst, cplx o) {
#if FFK_PC_WFOLD
            // W_a[m][n][j] = Bbar_a[m][n] T[n][j], (m, n, j) = this lane's index: the factors come
            // from the lanes that loaded them
            if (lane < DD) dst[lane] = o;
            const int src_t = lane % DD;                          // T[n][j]
            const cplx tv = {__shfl(o.re, src_t, 64), __shfl(o.im, src_t, 64)};
#pragma unroll
            for (int a = 0; a < NC; ++a) {
                const int src_b = DD + a*DD + lane / D;           // Bbar_a[m][n]
                const cplx bv = {__shfl(o.re, src_b, 64), __shfl(o.im, src_b, 64)};
                dst[DD + a*D*DD + lane] = cmul(bv, tv);
            }
#else
            if (lane < n_ops) dst[lane] = o;
#endif
        };
        auto load_row = [&](int g) -> cplx {
            const cplx* src = reinterpret_cast<const cplx*>(segtab + static_cast<size_t>(g)*S);
            return lane < S/2 ? src[lane] : cplx{0.0, 0.0};
        };
        auto generate = [&](int buf, int slot) {
            const double* st = rows + slot*S;
            cplx* tile = lds + static_cast<size_t>(buf)*BUF + lane;
            const double dtg = st[0];
            cplx ph;
            sincos_pi<true>(om*st[1], &ph.im, &ph.re);
            double sa, ca;
            sincos_pi<true>(0.5*(om*dtg), &sa, &ca);
            const PhasedFrequency pf = phased_frequency(om, dtg, ph, sa, ca);
#pragma unroll
            for (int k = 0; k < PcEntries<D>::count; ++k) {
                constexpr auto& sl = PcEntries<D>::slots;
                const int e = sl.v[k];
                const double* r = st + seg_rec(e);
                tile[k*TS] = phased_integral_aa(pf, r[0], r[1], r[2]);
            }
        };
        static_assert((1 + NC)*DD <= 64 && S/2 <= 64, "one staging element per lane");
        static_assert(!FFK_PC_WFOLD || D*DD == 64, "W_a has one element per lane");
        // prologue: rows g0, g0+1 and operands g0 straight in, tile g0
        if (g0 < g1) {
            const cplx r0 = load_row(g0), o0 = load_ops(g0);
            const cplx r1 = g0 + 1 < g1 ? load_row(g0 + 1) : cplx{0.0, 0.0};
            if (lane < S/2) {
                reinterpret_cast<cplx*>(rows)[lane] = r0;
                reinterpret_cast<cplx*>(rows + S)[lane] = r1;
            }
            store_ops(lds + TILE, o0);
        }
        __syncthreads();                              // rows of g0 visible to the whole sub-chunk
        if (g0 < g1) generate_first_share();
        __syncthreads();
        for (int it = 0; it < sub_len; ++it) {
            const int g = g0 + it;
            const int nb = (it + 1) & 1;
            if (g + 1 < g1) {
                // operands of g+1 and the table row of g+2: loads now, LDS stores after the tile
                const cplx o = load_ops(g + 1);
                const cplx r = g + 2 < g1 ? load_row(g + 2) : cplx{0.0, 0.0};
#if !(defined(FFK_PC_ABLATE) && FFK_PC_ABLATE == 1)   /* diagnostic: no generation */
                generate(nb, nb);                     // row g+1 lives in slot (it+1) & 1
#endif
                store_ops(lds + static_cast<size_t>(nb)*BUF + TILE, o);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();      // row g+1 fully read before it is replaced
                if (lane < S/2) reinterpret_cast<cplx*>(rows + (it & 1)*S)[lane] = r;
            }
            __syncthreads();
        }
    } else if constexpr (MF == 1) {
        // ---- matrix-core consumers ---------------------------------------------------------------
        // lane (cl = lane & 15, q = lane >> 4); per group wg of 16 frequencies (column cl):
        //   step 1:  Z_m[j = q]  = sum_n T[n, j] X_m[n],  X_m[n = q] = Bbar[m, q] E[m, q]     (m = 0..3)
        //   4 x 4 transpose of (m, q) across the 16-lane rows
        //   step 2:  Y[i = q, j] += sum_m conj(T[m, i]) Z_m[j]
        // Both steps take the SAME A operand, T[q][cl & 3] (conjugation through the NEG bits): one
        // complex per lane and segment.  32 matrix instructions per group replace 128 v_fma_f64.
        static_assert(D == 4, "one 4 x 4 block per matrix");
        const int cl = lane & 15, q = lane >> 4, c4 = cl & 3;
        double Yr[4][D], Yi[4][D];                    // [frequency group][column j], row i = q
#pragma unroll
        for (int wg = 0; wg < 4; ++wg)
#pragma unroll
            for (int j = 0; j < D; ++j) {
                Yr[wg][j] = 0.0;
                Yi[wg][j] = 0.0;
            }
        // tile slot of entry (m, n = q): the diagonal entries share slot 0
        int slot[D];
#pragma unroll
        for (int m = 0; m < D; ++m) {
            const int e = m*D + q;
            slot[m] = (m == q) ? 0 : e - (e > 5) - (e > 10);
        }
        __syncthreads();
        if (g0 < g1) generate_first_share();
        __syncthreads();
        for (int it = 0; it < sub_len; ++it) {
            const int g = g0 + it;
            if (active && g < g1) {
                const cplx* tile = lds + static_cast<size_t>(it & 1)*BUF;
                const cplx* opT = tile + TILE;
                const cplx* opB = opT + (1 + cidx)*DD;
                const cplx t = opT[q*D + c4];
                cplx b[D];
#pragma unroll
                for (int m = 0; m < D; ++m) b[m] = opB[m*D + q];
#pragma unroll
                for (int wg = 0; wg < 4; ++wg) {
                    const cplx* ecol = tile + 16*wg + cl;
                    double zr[4], zi[4];
#pragma unroll
                    for (int m = 0; m < D; ++m) {
                        const cplx x = cmul(b[m], ecol[slot[m]*64]);
                        zr[m] = __builtin_amdgcn_mfma_f64_4x4x4f64(t.re, x.re, 0.0, 0, 0, 0);
                        zi[m] = __builtin_amdgcn_mfma_f64_4x4x4f64(t.re, x.im, 0.0, 0, 0, 0);
                        zr[m] = __builtin_amdgcn_mfma_f64_4x4x4f64(t.im, x.im, zr[m], 0, 0, 1);
                        zi[m] = __builtin_amdgcn_mfma_f64_4x4x4f64(t.im, x.re, zi[m], 0, 0, 0);
                    }
                    transpose_rows(zr);
                    transpose_rows(zi);
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        Yr[wg][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(t.re, zr[j], Yr[wg][j], 0, 0, 0);
                        Yi[wg][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(t.re, zi[j], Yi[wg][j], 0, 0, 0);
                    }
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        Yr[wg][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(t.im, zi[j], Yr[wg][j], 0, 0, 0);
                        Yi[wg][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(t.im, zr[j], Yi[wg][j], 0, 0, 1);
                    }
                }
            }
            __syncthreads();
        }
        // sub-chunks > 0 hand their accumulators to sub-chunk 0 through LDS (tiles are dead now)
        cplx* red = reinterpret_cast<cplx*>(lds_raw);
        constexpr int YSZ = DD*64;
#pragma unroll
        for (int s = 1; s < GS; ++s) {
            if (sub == s) {
                cplx* dst = red + static_cast<size_t>(cidx)*YSZ + lane;
#pragma unroll
                for (int wg = 0; wg < 4; ++wg)
#pragma unroll
                    for (int j = 0; j < D; ++j) dst[(wg*D + j)*64] = {Yr[wg][j], Yi[wg][j]};
            }
            __syncthreads();
            if (sub == 0) {
                const cplx* srcy = red + static_cast<size_t>(cidx)*YSZ + lane;
#pragma unroll
                for (int wg = 0; wg < 4; ++wg)
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        const cplx v = srcy[(wg*D + j)*64];
                        Yr[wg][j] += v.re;
                        Yi[wg][j] += v.im;
                    }
            }
            __syncthreads();
        }
        if (sub == 0 && active) {
#pragma unroll
            for (int wg = 0; wg < 4; ++wg) {
                const int iwm = blockIdx.x*64 + 16*wg + cl;
                if (iwm < W) {
                    cplx* out = Ypart + ((static_cast<size_t>(blockIdx.z)*A + alpha)*DD)*W + iwm;
#pragma unroll
                    for (int j = 0; j < D; ++j) out[static_cast<size_t>(q*D + j)*W] = {Yr[wg][j], Yi[wg][j]};
                }
            }
        }
        return;
    } else if constexpr (MF == 2) {
        // ---- matrix-core consumers, one frequency per 4 x 4 x 4 block ----------------------------
        // lane (c = lane & 15, q = lane >> 4) supplies A_b[c & 3][q], B_b[q][c & 3] of block
        // b = c >> 2 and receives D_b[q][c & 3]: a product's result is the transpose of an A operand.
        //   step 1:  P[n, i] = sum_m X[m, n] conj(T[m, i])     A = X^T: lane holds X[q][c & 3]
        //   step 2:  Y[i, j] += sum_n P[n, i] T[n, j]          A = P^T: step 1's registers
        // B is T[q][c & 3] in both steps (conjugation through the NEG bit), Bbar[q][c & 3] is read
        // once per segment: 16 + 2 LDS reads and 16 complex products per lane and segment, 8 matrix
        // instructions per set of four frequencies, 16 sets per wavefront.
        static_assert(D == 4, "one 4 x 4 block per matrix");
        const int cl = lane & 15, q = lane >> 4, c4 = cl & 3, b = cl >> 2;
        double Yr[16], Yi[16];                        // [set]: Y[q][c4] at frequency 4 set + b
#pragma unroll
        for (int set = 0; set < 16; ++set) {
            Yr[set] = 0.0;
            Yi[set] = 0.0;
        }
        const int e_mine = q*D + c4;
        const int slot = (q == c4) ? 0 : e_mine - (e_mine > 5) - (e_mine > 10);
        __syncthreads();
        if (g0 < g1) generate_first_share();
        __syncthreads();
        for (int it = 0; it < sub_len; ++it) {
            const int g = g0 + it;
            if (active && g < g1) {
                const cplx* tile = lds + static_cast<size_t>(it & 1)*BUF;
                const cplx* opT = tile + TILE;
                const cplx t = opT[e_mine];
                const cplx bb = opT[(1 + cidx)*DD + e_mine];
                const cplx* ecol = tile + slot*TS + b;
#pragma unroll
                for (int set = 0; set < 16; ++set) {
                    const cplx x = cmul(bb, ecol[4*set]);
                    double pr = __builtin_amdgcn_mfma_f64_4x4x4f64(x.re, t.re, 0.0, 0, 0, 0);
                    double pi = __builtin_amdgcn_mfma_f64_4x4x4f64(x.im, t.re, 0.0, 0, 0, 0);
                    pr = __builtin_amdgcn_mfma_f64_4x4x4f64(x.im, t.im, pr, 0, 0, 0);
                    pi = __builtin_amdgcn_mfma_f64_4x4x4f64(x.re, t.im, pi, 0, 0, 1);
                    Yr[set] = __builtin_amdgcn_mfma_f64_4x4x4f64(pr, t.re, Yr[set], 0, 0, 0);
                    Yi[set] = __builtin_amdgcn_mfma_f64_4x4x4f64(pr, t.im, Yi[set], 0, 0, 0);
                    Yr[set] = __builtin_amdgcn_mfma_f64_4x4x4f64(pi, t.im, Yr[set], 0, 0, 1);
                    Yi[set] = __builtin_amdgcn_mfma_f64_4x4x4f64(pi, t.re, Yi[set], 0, 0, 0);
                }
            }
            __syncthreads();
        }
        // sub-chunks > 0 hand their accumulators to sub-chunk 0 through LDS (tiles are dead now)
        cplx* red = reinterpret_cast<cplx*>(lds_raw);
        constexpr int YSZ = DD*64;
#pragma unroll
        for (int s = 1; s < GS; ++s) {
            if (sub == s) {
                cplx* dst = red + static_cast<size_t>(cidx)*YSZ + lane;
#pragma unroll
                for (int set = 0; set < 16; ++set) dst[set*64] = {Yr[set], Yi[set]};
            }
            __syncthreads();
            if (sub == 0) {
                const cplx* srcy = red + static_cast<size_t>(cidx)*YSZ + lane;
#pragma unroll
                for (int set = 0; set < 16; ++set) {
                    const cplx v = srcy[set*64];
                    Yr[set] += v.re;
                    Yi[set] += v.im;
                }
            }
            __syncthreads();
        }
        if (sub == 0 && active) {
            cplx* out = Ypart + ((static_cast<size_t>(blockIdx.z)*A + alpha)*DD + e_mine)*W;
#pragma unroll
            for (int set = 0; set < 16; ++set) {
                const int iws = blockIdx.x*64 + 4*set + b;
                if (iws < W) out[iws] = {Yr[set], Yi[set]};
            }
        }
        return;
    } else {
        // ---- consumers: Y += T^dag [Bbar o E] T on the tile of the current segment -------------
#if defined(FFK_PC_PRIO_CONSUMER)     /* tuning builds */
        __builtin_amdgcn_s_setprio(FFK_PC_PRIO_CONSUMER);
#endif
        cplx Y[D][D];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) Y[i][j] = {0.0, 0.0};
#if !defined(FFK_PC_T_FROM_LDS)
        // T_g through scalar loads (wave-uniform addresses -> SGPRs feeding v_fma_f64 directly):
        // 64 VGPRs less per consumer; the loads for segment g+1 are issued at the end of segment g
        cplx Ts[D][D];
        auto load_T = [&](int g) {
            const cplx* tg = ops + static_cast<size_t>(g)*(1 + A)*DD;
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) Ts[i][j] = tg[i*D + j];
        };
        if (g0 < g1) load_T(g0);
#endif
        __syncthreads();
        if (g0 < g1) generate_first_share();
        __syncthreads();
        for (int it = 0; it < sub_len; ++it) {
            const int g = g0 + it;
#if defined(FFK_PC_ABLATE) && FFK_PC_ABLATE == 2      /* diagnostic: no contraction */
            if (false) {
#else
            if (active && g < g1) {
#endif
                const cplx* tile = lds + static_cast<size_t>(it & 1)*BUF;
                const cplx* src = tile + lane;
                const cplx* opT = tile + TILE;                       // T[n][j]
#if FFK_PC_WFOLD
                const cplx* opW = opT + DD + cidx*D*DD;              // Bbar_alpha[m][n] T[n][j]
#else
                const cplx* opB = opT + (1 + cidx)*DD;               // Bbar_alpha[m][n]
#endif
#pragma unroll
                for (int m = 0; m < D; ++m) {
                    cplx Z[D];
#pragma unroll
                    for (int j = 0; j < D; ++j) Z[j] = {0.0, 0.0};
#if FFK_PC_WFOLD
                    // Z_j = sum_n E_mn (Bbar_mn T_nj): no separate Bbar o E pass
#pragma unroll
                    for (int n = 0; n < D; ++n) {
                        const int slot = PcEntries<D>::slot_table.v[m*D + n];
                        const cplx e = src[slot*64];
#pragma unroll
                        for (int j = 0; j < D; ++j) cmac(Z[j], opW[(m*D + n)*D + j], e);
                    }
#else
                    cplx X[D];
#pragma unroll
                    for (int n = 0; n < D; ++n) {
                        const int slot = PcEntries<D>::slot_table.v[m*D + n];
                        X[n] = cmul(opB[m*D + n], src[slot*64]);
                    }
#pragma unroll
                    for (int n = 0; n < D; ++n)
#pragma unroll
                        for (int j = 0; j < D; ++j) {
#if !defined(FFK_PC_T_FROM_LDS)
                            cmac(Z[j], Ts[n][j], X[n]);
#else
                            cmac(Z[j], opT[n*D + j], X[n]);
#endif
                        }
#endif
#pragma unroll
                    for (int i = 0; i < D; ++i) {
#if !defined(FFK_PC_T_FROM_LDS)
                        const cplx t = Ts[m][i];
#else
                        const cplx t = opT[m*D + i];
#endif
#pragma unroll
                        for (int j = 0; j < D; ++j) cmac_conj(Y[i][j], t, Z[j]);
                    }
                }
            }
#if !defined(FFK_PC_T_FROM_LDS)
            if (g + 1 < g1) load_T(g + 1);
#endif
            __syncthreads();
        }
        // sub-chunks > 0 hand their accumulators to sub-chunk 0 through LDS (tiles are dead now)
        cplx* red = reinterpret_cast<cplx*>(lds_raw);
        constexpr int YSZ = DD*64;
#pragma unroll
        for (int s = 1; s < GS; ++s) {
            if (sub == s) {
                cplx* dst = red + static_cast<size_t>(cidx)*YSZ + lane;
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) dst[(i*D + j)*64] = Y[i][j];
            }
            __syncthreads();
            if (sub == 0) {
                const cplx* srcy = red + static_cast<size_t>(cidx)*YSZ + lane;
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        const cplx v = srcy[(i*D + j)*64];
                        Y[i][j].re += v.re;
                        Y[i][j].im += v.im;
                    }
            }
            __syncthreads();
        }
        if (sub == 0 && active && iw < W) {
            cplx* out = Ypart + ((static_cast<size_t>(blockIdx.z)*A + alpha)*DD)*W + iw;
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) out[static_cast<size_t>(i*D + j)*W] = Y[i][j];
        }
        return;
    }
    // producers keep the consumers' epilogue barriers company
#pragma unroll
    for (int s = 1; s < GS; ++s) {
        __syncthreads();
        __syncthreads();
    }
}

// FFK_TUNE_PC_MFMA: 0 vector consumers (default), 1 / 2 the matrix-core forms (tuning / A-B)
int pc_consumer_form() {
    static const int form = [] {
        const char* e = std::getenv("FFK_TUNE_PC_MFMA");
        return (e != nullptr && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 0;
    }();
    return form;
}

template <int D, int NC>
hipError_t launch_pc(const double* omega, int W, const double* segtab, const cplx* ops, int G, int A,
                     int chunks, int chunk_len, cplx* Ypart, hipStream_t stream) {
    const int lds = pc_accumulate_lds_bytes(D, NC);
    const int form = pc_consumer_form();
    auto kern = form == 2   ? ctrl_accumulate_pc_kernel<D, NC, 2>
                : form == 1 ? ctrl_accumulate_pc_kernel<D, NC, 1>
                            : ctrl_accumulate_pc_kernel<D, NC, 0>;
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (err != hipSuccess) return err;
    const dim3 grid((W + 63)/64, (A + NC - 1)/NC, chunks);
    hipLaunchKernelGGL(kern, grid, dim3((NC + 1)*kPcSub*64), lds, stream, omega, W, segtab, ops, G, A,
                       chunk_len, Ypart);
    return hipGetLastError();
}

}  // namespace

bool pc_accumulate_supported(int d, int A) { return d == 4 && A >= 1; }
int pc_accumulate_ops_per_block(int A) { return A >= 3 ? 3 : A; }
int pc_accumulate_subchunks() { return kPcSub; }
int pc_accumulate_lds_bytes(int d, int nc) {
    const int S = seg_stride(d), dd = d*d;
    const int ops = FFK_PC_WFOLD ? dd + nc*d*dd : (1 + nc)*dd;
    const int entries = d*(d - 1) + 1;                // distinct integral entries (PcEntries<D>::count)
    return static_cast<int>(kPcSub*(2*(entries*pc_tile_stride(pc_consumer_form()) + ops) + S)*sizeof(cplx));
}

hipError_t launch_accumulate_pc(const double* omega, int W, const double* segtab, const cplx* ops,
                                int G, int d, int A, int nc, int chunks, int chunk_len, cplx* Ypart,
                                hipStream_t stream) {
    if (d != 4) return hipErrorInvalidValue;
    switch (nc) {
        case 1: return launch_pc<4, 1>(omega, W, segtab, ops, G, A, chunks, chunk_len, Ypart, stream);
        case 2: return launch_pc<4, 2>(omega, W, segtab, ops, G, A, chunks, chunk_len, Ypart, stream);
        case 3: return launch_pc<4, 3>(omega, W, segtab, ops, G, A, chunks, chunk_len, Ypart, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ffk
