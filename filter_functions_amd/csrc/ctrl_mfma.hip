// ctrl_mfma.hip -- K3m: the control-matrix accumulation for large Hilbert spaces (d = 12, 16; d = 4, 8
// on request) on the FP64 matrix cores.  Same mathematics, inputs and output layout as ctrl.hip:
//     Y_a(w) = sum_g T_g^dag [ Bbar_a^(g) o E^(g)(w) ] T_g ,   E = e^{i w t_g} I^(g)(w),
// with the two d x d products per (segment, frequency, noise operator) on MFMA.  The waves of a
// block share the generated E tile (d^2 entries x 16 frequencies) through LDS.  Three forms, all kept:
//   * ctrl_accumulate_mfma4_kernel<D, JH, MAXW, BF = true> -- THE DEFAULT for d = 12, 16 (round 3).
//     v_mfma_f64_4x4x4_4b with ONE FREQUENCY PER 4 x 4 x 4 BLOCK: the result of the first product
//     P = X^T conj(T) is, register for register, the transposed A operand of the second,
//     Y += P^T T -- nothing moves between the two products, both take the same entries of T (held in
//     registers for the whole segment) as B operand, X = Bbar o E is formed once per operator.  The
//     JH wavefronts of an operator split the tile's 16 frequencies.  See the comment above the
//     kernel; d = 16: 4.65 ms at config 5 (0.71 of the FP64 peak), d = 12: 1.80 ms.
//   * ... BF = false -- the same instruction with the FREQUENCY AS THE COLUMN of the tile (round 2):
//       step 1, per row m:    Z_m[j, w]  = sum_n  T[n, j]        X_m[n, w],   X_m[n, w] = Bbar[m, n] E[m, n](w)
//       step 2, per column j: Y[i, j, w] += sum_m conj(T[m, i]) Z_m[j, w].
//     A operands are entries of T (read from LDS per use), B operands the per-frequency X / Z.  Step 1
//     leaves Z_m[j = q + 4r, w_c] in lane (c, q) (c = lane & 15, q = lane >> 4), step 2 wants
//     Z_{m = 4 mg + q}[j, w_c] there: a 4 x 4 transpose across the four 16-lane rows of the wavefront
//     (v_permlane16_swap / v_permlane32_swap, ffk_mfma_util.h).  The columns of Y are split over JH
//     wavefronts per operator.  5.18 ms / 2.41 ms on the same shapes (round-3 A/B build).
//   * ctrl_accumulate_mfma_kernel<D> -- v_mfma_f64_16x16x4, one wavefront per noise operator with
//     all d x d x 16 accumulators (one wavefront per SIMD): 6.7 ms at d = 16; a tuning reference
//     (round-3 A/B build).
#include <algorithm>
#include <cstdlib>

#include <cstdio>

#include "ffk_internal.h"
#include "ffk_mfma_util.h"

namespace ffk {
namespace {

#ifndef FFK_MFMA_BF_DEFAULT     /* 1: the block-frequency form is the default for d = 12, 16 */
#define FFK_MFMA_BF_DEFAULT 1
#endif
using f64x4 = __attribute__((ext_vector_type(4))) double;
constexpr int kMW = 4;   // wavefronts (= noise operators) per block, one per SIMD

#ifdef FFK_MFMA_CLOCK   /* tuning build: where a wavefront of the d = 12, 16 kernel spends its cycles */
// sums over every wavefront of the last launches, shader-clock cycles: [0] wait at the barrier that
// frees the tile, [1] issue of the staging loads + generation, [2] parking the staged operands,
// [3] wait at the barrier that publishes the tile, [4] contraction, [5] segment-steps counted
__device__ unsigned long long g_mfma_phase[8];
#define FFK_MC_T() __builtin_amdgcn_s_memtime()
// (summed per wavefront in scalar registers, one atomic per phase at the end of the kernel: six atomics
// per step on six addresses stalled the staging loads behind them and tripled the kernel's time)
#define FFK_MC_DECL() unsigned long long mc_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define FFK_MC_ADD(slot, t0, t1) mc_sum[slot] += static_cast<unsigned long long>((t1) - (t0))
#define FFK_MC_FLUSH() \
    do { if (lane == 0) for (int mc_i = 0; mc_i < 8; ++mc_i) atomicAdd(&g_mfma_phase[mc_i], mc_sum[mc_i]); } while (0)
#else
#define FFK_MC_T() 0ull
#define FFK_MC_DECL()
#define FFK_MC_ADD(slot, t0, t1)
#define FFK_MC_FLUSH()
#endif

template <int D>
struct MfmaLayout {
    static constexpr int S = seg_stride(D);
    static constexpr int DD = D*D;
    static constexpr int kTile = DD*16;                   // cplx: E[m*D + n][16 frequencies]
    static constexpr int kOps = (1 + kMW)*DD;             // cplx per operand buffer
    static constexpr int kStageElems = kOps + S/2;        // cplx staged per segment
    static constexpr int kStagePerThread = (kStageElems + kMW*64 - 1)/(kMW*64);
    static constexpr size_t lds_bytes = (static_cast<size_t>(kTile) + 2*kOps)*sizeof(cplx) +
                                        2*static_cast<size_t>(S)*sizeof(double);
};

template <int D>
__global__ __launch_bounds__(kMW*64, 1) void ctrl_accumulate_mfma_kernel(
    const double* __restrict__ omega, int W, const double* __restrict__ segtab,
    const cplx* __restrict__ ops, int G, int A, int chunk_len, cplx* __restrict__ Ypart) {
    static_assert(D % 4 == 0 && D >= 8 && D <= 16, "d must be 8, 12 or 16");
    using L = MfmaLayout<D>;
    constexpr int S = L::S, DD = L::DD, NS = D/4;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    cplx* tile = reinterpret_cast<cplx*>(lds_raw);
    cplx* opsb = tile + L::kTile;
    double* rows = reinterpret_cast<double*>(opsb + 2*L::kOps);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    const int iw = blockIdx.x*16 + c;
    const double om = omega[iw < W ? iw : W - 1];
    const int alpha0 = blockIdx.y*kMW;
    const int alpha = alpha0 + wave;
    const bool active = alpha < A;
    const int n_alpha = min(kMW, A - alpha0);
    const int n_ops = (1 + n_alpha)*DD;
    const int g0 = blockIdx.z*chunk_len;
    const int g1 = min(G, g0 + chunk_len);

    f64x4 Yr[D], Yi[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        Yr[j] = {0.0, 0.0, 0.0, 0.0};
        Yi[j] = {0.0, 0.0, 0.0, 0.0};
    }

    // staging copy of segment g: operands (T_g, Bbar of this block's operators) -> opsb[buf],
    // table row -> rows[slot]; loads issued before the contraction, parked after it
    cplx staged[L::kStagePerThread];
    auto issue_stage = [&](int g) {
        const cplx* src_ops = ops + static_cast<size_t>(g)*(1 + A)*DD;
        const cplx* src_tab = reinterpret_cast<const cplx*>(segtab + static_cast<size_t>(g)*S);
#pragma unroll
        for (int k = 0; k < L::kStagePerThread; ++k) {
            const int e = tid + k*kMW*64;
            // (one unconditional assignment per element: with the two guarded ones the array stayed in
            // scratch memory)
            const cplx* src = e < n_ops ? src_ops + (e < DD ? e : e + alpha0*DD) : src_tab + (e - n_ops);
            staged[k] = e < n_ops + S/2 ? *src : cplx{0.0, 0.0};
        }
    };
    auto park = [&](int buf) {
        cplx* dst_ops = opsb + buf*L::kOps;
        cplx* dst_tab = reinterpret_cast<cplx*>(rows + buf*S);
#pragma unroll
        for (int k = 0; k < L::kStagePerThread; ++k) {
            const int e = tid + k*kMW*64;
            if (e < n_ops)
                dst_ops[e] = staged[k];
            else if (e < n_ops + S/2)
                dst_tab[e - n_ops] = staged[k];
        }
    };

    // generation: thread (wave, q, c) -> entries (wave*4 + q) + 16 s of frequency c
    auto generate = [&](int slot) {
        const double* st = rows + slot*S;
        const double dtg = st[0];
        cplx ph;
        sincos_pi<true>(om*st[1], &ph.im, &ph.re);
        double sa, ca;
        sincos_pi<true>(0.5*(om*dtg), &sa, &ca);
        const int eg = wave*4 + q;
        const PhasedFrequency pf = phased_frequency(om, dtg, ph, sa, ca);
#pragma unroll 4
        for (int s = 0; s < DD/16; ++s) {
            const int e = eg + 16*s;
            const double* r = st + seg_rec(e);
            tile[e*16 + c] = phased_integral_aa(pf, r[0], r[1], r[2]);
        }
    };

    auto contract = [&](int buf) {
        const cplx* opT = opsb + buf*L::kOps;
        const cplx* opB = opT + (1 + wave)*DD;
        // A operand of both steps: T[4 s + q][c]  (rows beyond d contribute zeros)
        double tr[NS], ti[NS], nti[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            cplx t = {0.0, 0.0};
            if (c < D) t = opT[(4*s + q)*D + c];
            tr[s] = t.re;
            ti[s] = t.im;
            nti[s] = -t.im;
        }
#pragma unroll
        for (int mg = 0; mg < NS; ++mg) {
            f64x4 Zr[4], Zi[4];
#pragma unroll
            for (int mm = 0; mm < 4; ++mm) {
                Zr[mm] = {0.0, 0.0, 0.0, 0.0};
                Zi[mm] = {0.0, 0.0, 0.0, 0.0};
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int n = 4*s + q;
                cplx x[4];
#pragma unroll
                for (int mm = 0; mm < 4; ++mm) {
                    const int m = 4*mg + mm;
#if defined(FFK_M_ABLATE) && FFK_M_ABLATE == 4      /* diagnostic: no LDS operands */
                    x[mm] = {om + m, om - n};
#else
                    x[mm] = cmul(opB[m*D + n], tile[(m*D + n)*16 + c]);
#endif
                }
                // eight independent accumulators between two uses of the same one
#pragma unroll
                for (int mm = 0; mm < 4; ++mm) {
                    Zr[mm] = __builtin_amdgcn_mfma_f64_16x16x4f64(tr[s], x[mm].re, Zr[mm], 0, 0, 0);
                    Zi[mm] = __builtin_amdgcn_mfma_f64_16x16x4f64(tr[s], x[mm].im, Zi[mm], 0, 0, 0);
                }
#pragma unroll
                for (int mm = 0; mm < 4; ++mm) {
                    Zr[mm] = __builtin_amdgcn_mfma_f64_16x16x4f64(nti[s], x[mm].im, Zr[mm], 0, 0, 0);
                    Zi[mm] = __builtin_amdgcn_mfma_f64_16x16x4f64(ti[s], x[mm].re, Zi[mm], 0, 0, 0);
                }
            }
            // lane (c, q) holds Z_{4 mg + mm}[j = q + 4 r]; step 2 needs Z_{4 mg + q}[j = qo + 4 r]
#pragma unroll
            for (int r = 0; r < (D + 3)/4; ++r) {
                double zr[4] = {Zr[0][r], Zr[1][r], Zr[2][r], Zr[3][r]};
                double zi[4] = {Zi[0][r], Zi[1][r], Zi[2][r], Zi[3][r]};
#if !(defined(FFK_M_ABLATE) && FFK_M_ABLATE == 3)   /* diagnostic: no transposes (wrong results) */
                transpose_rows(zr);
                transpose_rows(zi);
#endif
#pragma unroll
                for (int qo = 0; qo < 4; ++qo) {
                    const int j = qo + 4*r;
                    if (j >= D) continue;
                    // Y[i, j] += conj(T[m, i]) Z_m[j]
                    Yr[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(tr[mg], zr[qo], Yr[j], 0, 0, 0);
                    Yi[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(tr[mg], zi[qo], Yi[j], 0, 0, 0);
                    Yr[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ti[mg], zi[qo], Yr[j], 0, 0, 0);
                    Yi[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(nti[mg], zr[qo], Yi[j], 0, 0, 0);
                }
            }
        }
    };

    // prologue: segment g0's operands and table row straight into buffer 0
    if (g0 < g1) {
        issue_stage(g0);
        park(0);
    }
    for (int g = g0; g < g1; ++g) {
        const int buf = (g - g0) & 1;
        __syncthreads();                 // tile free, staged data of segment g visible
#if !(defined(FFK_M_ABLATE) && FFK_M_ABLATE == 1)   /* diagnostic build 1: no generation */
        generate(buf);
#endif
        __syncthreads();
        if (g + 1 < g1) issue_stage(g + 1);
#if !(defined(FFK_M_ABLATE) && FFK_M_ABLATE == 2)   /* diagnostic build 2: no contraction */
        if (active) contract(buf);
#endif
        if (g + 1 < g1) park(buf ^ 1);
    }

    if (active && iw < W) {
        cplx* out = Ypart + ((static_cast<size_t>(blockIdx.z)*A + alpha)*DD)*W + iw;
#pragma unroll
        for (int j = 0; j < D; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = q + 4*r;
                if (i < D) out[static_cast<size_t>(i*D + j)*W] = {Yr[j][r], Yi[j][r]};
            }
    }
}

// ---------------------------------------------------------------------------------------------
// d = 8: the same scheme on v_mfma_f64_4x4x4_4b (4 blocks of 4x4x4 per instruction), which has no
// 16-row tile to leave half empty.  Layout (tools/mfma4_layout_probe.hip): with c = lane & 15,
// q = lane >> 4 and block b = c >> 2,  A[i = c & 3][k = q] (per block),  B[k = q][col = c],
// D[i = q][col = c]: again 16 frequencies as columns, 4 rows per instruction, A replicated over the
// blocks.  Accumulators are one f64 per lane and (row group, column): 2 d^2/4 registers, so the
// kernel runs at 2-3 waves per SIMD with NW wavefronts (= noise operators) per block.
// ---------------------------------------------------------------------------------------------
// JH > 1: the columns j of Y are split over JH wavefronts per noise operator (each computes Z only
// for its own row groups -- the 4-row instruction has no tile to leave half empty), which halves
// the accumulators per wave: d = 16 fits 256 registers, i.e. two wavefronts per SIMD.
// `nw` counts wavefronts; a block serves nw / JH noise operators.
// ---------------------------------------------------------------------------------------------
// BF ("block frequency"): the other way to lay the problem on the same instruction -- ONE frequency
// per 4 x 4 x 4 block (b = c >> 2) instead of one per column.  Lane (c, q) supplies A_b[c & 3][q] and
// B_b[q][c & 3] and receives D_b[q][c & 3], so a product's result is, register for register, the
// transpose of an A operand:
//     step 1:  P[n, i] = sum_m X[m, n] conj(T[m, i])      A = X^T (formed per lane), B = conj T
//     step 2:  Y[i, j] += sum_n P[n, i] T[n, j]            A = P^T = step 1's registers, B = T
// No transposes between the steps, both take the same (d/4)^2 entries of T per lane as B operand
// (held in registers for the whole segment), X is formed once per operator, and a wavefront's
// frequencies come as 4 / JH sets of four: the JH wavefronts of an operator split the tile's 16
// frequencies, not the columns of Y.  The tile's slots are 20 complex apart (16 frequencies + 4 of
// padding: a 16-lane row reads 4 slots x 4 frequencies, banks (4 (c & 3) + (c >> 2)) mod 16).
constexpr int mfma4_tile_stride(bool bf, int tw = 16) { return bf ? tw + 4 : 16; }
// TW (block-frequency form only): frequencies per tile.  16: a block = eight wavefronts, two per SIMD, all in the same
// phase -- generate, barrier, contract, barrier -- so the matrix pipe is idle while the tile is generated (7.1 k of a
// 40.7 k-cycle step, profiles/r04_o_*) and the wavefront of a SIMD that finishes its contraction first waits for its
// sibling.  8 (round 6): a block = FOUR wavefronts, one per SIMD, one operator each (JH = 1), 76 KB of LDS (operands
// single-buffered: copied by LDS-DMA behind the barrier that ends the contraction, in flight during the generation)
// -- TWO INDEPENDENT BLOCKS per CU, each with its own barriers: the SIMD's older wavefront wins the arbiter, runs
// ahead, and from then on one block generates its tile while the other's matrix instructions have the pipe.
template <int D, int JH, int MAXW = 8, bool BF = false, int TW = 16>
__global__ __launch_bounds__(MAXW*64, TW == 8 ? 2 : 1) void ctrl_accumulate_mfma4_kernel(
    const double* __restrict__ omega, int W, const double* __restrict__ segtab,
    const cplx* __restrict__ ops, int G, int A, int chunk_len, int nw, cplx* __restrict__ Ypart,
    int alpha_base, int alpha_end, ExpandEpilogue ep) {
    static_assert(D % 4 == 0 && D >= 4 && D <= 16, "d must be a multiple of 4");
    static_assert(BF ? (TW/4) % JH == 0 : (D/4) % JH == 0, "row groups / frequency sets must split evenly");
    static_assert(TW == 16 || (TW == 8 && BF), "8-frequency tiles: block-frequency form only");
    constexpr int S = seg_stride(D), DD = D*D, NS = D/4;
    constexpr int NJG = BF ? NS : NS/JH;   // row groups (of 4 columns of Y) owned by this wave
    constexpr int NSET = BF ? (TW/4)/JH : 1;    // BF: sets of four frequencies owned by this wave
    constexpr int TS = mfma4_tile_stride(BF, TW);
    constexpr bool kSingleOps = TW == 8;         // one operand buffer (see above)
    constexpr int kMaxStage = D == 16 ? (MAXW < 8 ? 8 : 4) : 8;   // staged elements per thread (see launch_d4)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int na = nw/JH;             // noise operators per block
    const int kops = (1 + na)*DD;
    cplx* tile = reinterpret_cast<cplx*>(lds_raw);
    cplx* opsb = tile + DD*TS;
    double* rows = reinterpret_cast<double*>(opsb + (kSingleOps ? 1 : 2)*kops);

    const int tid = threadIdx.x;
    const int nthreads = nw*64;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4, c4 = c & 3;
    const int iw = blockIdx.x*TW + (c & (TW - 1));      // the frequency this lane GENERATES entries for
    const double om = omega[iw < W ? iw : W - 1];
    const int alpha0 = alpha_base + blockIdx.y*na;   // the launch serves operators [alpha_base, alpha_end)
    const int alpha_l = wave / JH, jh = wave % JH;
    const int alpha = alpha0 + alpha_l;
    const bool active = alpha < alpha_end;
    const int n_alpha = min(na, alpha_end - alpha0);
    const int n_ops = (1 + n_alpha)*DD;
    const int g0 = blockIdx.z*chunk_len;
    const int g1 = min(G, g0 + chunk_len);

    // [row group ig][own column]; BF: [set*NS + ig][jg] = Y[4 ig + q][4 jg + (c & 3)] of frequency
    // 4 (jh NSET + set) + (c >> 2)
    constexpr int YA = BF ? NSET*NS : NS, YB = BF ? NS : 4*NJG;
    double Yr[YA][YB], Yi[YA][YB];
#pragma unroll
    for (int ig = 0; ig < YA; ++ig)
#pragma unroll
        for (int j = 0; j < YB; ++j) {
            Yr[ig][j] = 0.0;
            Yi[ig][j] = 0.0;
        }

    cplx staged[kMaxStage];
    auto issue_stage = [&](int g) {
        const cplx* src_ops = ops + static_cast<size_t>(g)*(1 + A)*DD;
        const cplx* src_tab = reinterpret_cast<const cplx*>(segtab + static_cast<size_t>(g)*S);
        // The element numbers through an OPAQUE copy of the thread index: as loop invariants the source
        // offsets and range conditions of the staged elements were hoisted out of the segment loop,
        // kept alive across the contraction -- i.e. spilled -- and reloaded here: scratch load, wait,
        // global load, four times in a row, each wait draining the load before it (2.5-3 k cycles of
        // a 40.7 k-cycle step, profiles/r04_o_*).  Recomputing them costs a dozen instructions.
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));
#pragma unroll
        for (int k = 0; k < kMaxStage; ++k) {
            const int e = tid_o + k*nthreads;
            // (one unconditional assignment per element: with the two guarded ones the array stayed in
            // scratch memory)
            const cplx* src = e < n_ops ? src_ops + (e < DD ? e : e + alpha0*DD) : src_tab + (e - n_ops);
            staged[k] = e < n_ops + S/2 ? *src : cplx{0.0, 0.0};
        }
    };
    auto park = [&](int buf) {
        cplx* dst_ops = opsb + buf*kops;
        cplx* dst_tab = reinterpret_cast<cplx*>(rows + buf*S);
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));
#pragma unroll
        for (int k = 0; k < kMaxStage; ++k) {
            const int e = tid_o + k*nthreads;
            if (e < n_ops)
                dst_ops[e] = staged[k];
            else if (e < n_ops + S/2)
                dst_tab[e - n_ops] = staged[k];
        }
    };
    // BF, round 6: the staging copy by LDS-DMA (ffk_mfma_util.h lds_dma16): pieces of 64 complex numbers dealt over the
    // wavefronts, requested at the top of a step into the buffer the step before has finished with, in flight during
    // the generation AND the contraction, waited for before the barrier at the top of the next step.  No staging
    // registers (16 of 256, next to 5 spilled), no ds_write_b128 (35-45 cycles of issue each, tools/lds_issue_probe.py).
    // (g_ops / g_row: the segments whose operands / table row are copied, -1 = none; obuf / rbuf: their buffers)
    auto dma_stage2 = [&](int g_ops, int obuf, int g_row, int rbuf) {
        const int n_och = g_ops >= 0 ? (n_ops + 63) >> 6 : 0, n_rch = g_row >= 0 ? (S/2 + 63) >> 6 : 0;
        for (int ch = wave; ch < n_och + n_rch; ch += nw) {
            if (ch < n_och) {
                const cplx* src_ops = ops + static_cast<size_t>(g_ops)*(1 + A)*DD;
                const int e = 64*ch + lane;
                if (e < n_ops) lds_dma16(src_ops + (e < DD ? e : e + alpha0*DD), opsb + obuf*kops + 64*ch);
            } else {
                const cplx* src_tab = reinterpret_cast<const cplx*>(segtab + static_cast<size_t>(g_row)*S);
                const int e = 64*(ch - n_och) + lane;
                if (e < S/2) lds_dma16(src_tab + e, reinterpret_cast<cplx*>(rows + rbuf*S) + 64*(ch - n_och));
            }
        }
    };
    auto dma_stage = [&](int g, int buf) { dma_stage2(g, buf, g, buf); };
    auto generate = [&](int slot) {
        const double* st = rows + slot*S;
        const double dtg = st[0];
        cplx ph;
        sincos_pi<true>(om*st[1], &ph.im, &ph.re);
        double sa, ca;
        sincos_pi<true>(0.5*(om*dtg), &sa, &ca);
        const PhasedFrequency pf = phased_frequency(om, dtg, ph, sa, ca);
        // thread (wave, q, c): frequency c mod TW, entries (wave 4 + q) 16/TW + c/TW + 64 nw/TW k
        constexpr int EPL = 16/TW;                  // entries per 16-lane row
        for (int e = (wave*4 + q)*EPL + c/TW; e < DD; e += 4*nw*EPL) {
            const double* r = st + seg_rec(e);
            tile[e*TS + (c & (TW - 1))] = phased_integral_aa(pf, r[0], r[1], r[2]);
        }
    };
    // v_mfma_f64: the last builtin argument carries the NEG bits of the operands (bit 0 = A)
    auto contract = [&](int buf) {
        const cplx* opT = opsb + buf*kops;
        const cplx* opB = opT + (1 + alpha_l)*DD;
#pragma unroll 1
        for (int mg = 0; mg < NS; ++mg) {
            double zr[NJG][4], zi[NJG][4];   // [own row group][mm]
#pragma unroll
            for (int jg = 0; jg < NJG; ++jg)
#pragma unroll
                for (int mm = 0; mm < 4; ++mm) {
                    zr[jg][mm] = 0.0;
                    zi[jg][mm] = 0.0;
                }
#pragma unroll 1
            for (int s = 0; s < NS; ++s) {
                const int n = 4*s + q;
                cplx x[4];
#pragma unroll
                for (int mm = 0; mm < 4; ++mm) {
                    const int m = 4*mg + mm;
                    x[mm] = cmul(opB[m*D + n], tile[(m*D + n)*16 + c]);
                }
                // A operand of step 1: T[4 s + q][4 jg + (c & 3)] for the own row groups
                cplx tj[NJG];
#pragma unroll
                for (int jg = 0; jg < NJG; ++jg) tj[jg] = opT[n*D + 4*(jh*NJG + jg) + c4];
                // Z_m[j = 4 jg + q] += T[n, j] X_m[n]   (accumulators vary fastest)
#pragma unroll
                for (int jg = 0; jg < NJG; ++jg)
#pragma unroll
                    for (int mm = 0; mm < 4; ++mm) {
                        zr[jg][mm] = __builtin_amdgcn_mfma_f64_4x4x4f64(tj[jg].re, x[mm].re, zr[jg][mm], 0, 0, 0);
                        zi[jg][mm] = __builtin_amdgcn_mfma_f64_4x4x4f64(tj[jg].re, x[mm].im, zi[jg][mm], 0, 0, 0);
                    }
#pragma unroll
                for (int jg = 0; jg < NJG; ++jg)
#pragma unroll
                    for (int mm = 0; mm < 4; ++mm) {
                        zr[jg][mm] = __builtin_amdgcn_mfma_f64_4x4x4f64(tj[jg].im, x[mm].im, zr[jg][mm], 0, 0, 1);
                        zi[jg][mm] = __builtin_amdgcn_mfma_f64_4x4x4f64(tj[jg].im, x[mm].re, zi[jg][mm], 0, 0, 0);
                    }
            }
            // lane (c, q) holds Z_{4 mg + mm}[4 jg + q]; step 2 needs Z_{4 mg + q}[4 jg + qo]
#pragma unroll
            for (int jg = 0; jg < NJG; ++jg) {
                transpose_rows(zr[jg]);
                transpose_rows(zi[jg]);
            }
            // Y[i = 4 ig + q, j] += conj(T[m, i]) Z_m[j],  m = 4 mg + q;  A operand T[m][4 ig + (c & 3)]
            cplx ti_[NS];
#pragma unroll
            for (int ig = 0; ig < NS; ++ig) ti_[ig] = opT[(4*mg + q)*D + 4*ig + c4];
#pragma unroll
            for (int jg = 0; jg < NJG; ++jg)
#pragma unroll
                for (int qo = 0; qo < 4; ++qo) {
                    const int j = 4*jg + qo;          // own column index
#pragma unroll
                    for (int ig = 0; ig < NS; ++ig) {
                        Yr[ig][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(ti_[ig].re, zr[jg][qo], Yr[ig][j], 0, 0, 0);
                        Yi[ig][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(ti_[ig].re, zi[jg][qo], Yi[ig][j], 0, 0, 0);
                    }
#pragma unroll
                    for (int ig = 0; ig < NS; ++ig) {
                        Yr[ig][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(ti_[ig].im, zi[jg][qo], Yr[ig][j], 0, 0, 0);
                        Yi[ig][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(ti_[ig].im, zr[jg][qo], Yi[ig][j], 0, 0, 1);
                    }
                }
        }
    };

    auto contract_bf = [&](int buf) {
        const cplx* opT = opsb + (kSingleOps ? 0 : buf)*kops;
        const cplx* opB = opT + (1 + alpha_l)*DD;
        cplx tq[NS][NS];                              // T[4 s + q][4 g + c4]
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int g = 0; g < NS; ++g) tq[s][g] = opT[(4*s + q)*D + 4*g + c4];
        // One set of four frequencies at a time, one column group ng of X at a time: next to the
        // accumulators and T only (d/4) products P and one X entry are live (d = 16: 128 + 64 + 16
        // registers).  Bbar's entry is read again per set -- holding the d^2/16 of them does not fit.
#pragma unroll
        for (int set = 0; set < NSET; ++set) {
            const cplx* ecol = tile + 4*(jh*NSET + set) + (c >> 2);     // + slot*TS
#pragma unroll
            for (int ng = 0; ng < NS; ++ng) {
                // (keeps the scheduler from hoisting the LDS reads of later groups over this one's
                // products: without it the fully unrolled body spills)
                __builtin_amdgcn_sched_barrier(0);
                double pr[NS], pi[NS];                // P[4 ng + q][4 ig + c4]
#pragma unroll
                for (int ig = 0; ig < NS; ++ig) {
                    pr[ig] = 0.0;
                    pi[ig] = 0.0;
                }
                // (round 6, measured and not kept: the column group's d/4 entries of X first, as one group of vector
                // instructions, then its matrix instructions -- at two wavefronts per SIMD no difference, 5.27 ms either way)
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int e = (4*s + q)*D + 4*ng + c4;
                    const cplx x = cmul(opB[e], ecol[e*TS]);  // X[4 s + q][4 ng + c4]
#pragma unroll
                    for (int ig = 0; ig < NS; ++ig) {
                        pr[ig] = __builtin_amdgcn_mfma_f64_4x4x4f64(x.re, tq[s][ig].re, pr[ig], 0, 0, 0);
                        pi[ig] = __builtin_amdgcn_mfma_f64_4x4x4f64(x.im, tq[s][ig].re, pi[ig], 0, 0, 0);
                    }
#pragma unroll
                    for (int ig = 0; ig < NS; ++ig) {
                        pr[ig] = __builtin_amdgcn_mfma_f64_4x4x4f64(x.im, tq[s][ig].im, pr[ig], 0, 0, 0);
                        pi[ig] = __builtin_amdgcn_mfma_f64_4x4x4f64(x.re, tq[s][ig].im, pi[ig], 0, 0, 1);
                    }
                }
#pragma unroll
                for (int ig = 0; ig < NS; ++ig) {
#pragma unroll
                    for (int jg = 0; jg < NS; ++jg) {
                        Yr[set*NS + ig][jg] = __builtin_amdgcn_mfma_f64_4x4x4f64(pr[ig], tq[ng][jg].re, Yr[set*NS + ig][jg], 0, 0, 0);
                        Yi[set*NS + ig][jg] = __builtin_amdgcn_mfma_f64_4x4x4f64(pr[ig], tq[ng][jg].im, Yi[set*NS + ig][jg], 0, 0, 0);
                    }
#pragma unroll
                    for (int jg = 0; jg < NS; ++jg) {
                        Yr[set*NS + ig][jg] = __builtin_amdgcn_mfma_f64_4x4x4f64(pi[ig], tq[ng][jg].im, Yr[set*NS + ig][jg], 0, 0, 1);
                        Yi[set*NS + ig][jg] = __builtin_amdgcn_mfma_f64_4x4x4f64(pi[ig], tq[ng][jg].re, Yi[set*NS + ig][jg], 0, 0, 0);
                    }
                }
            }
        }
    };

#if defined(FFK_M_REGISTER_STAGING)   /* A/B builds: rounds 1-5's staging through registers */
    constexpr bool kDmaStage = false;
#else
    constexpr bool kDmaStage = BF;
#endif
    static_assert(!kSingleOps || kDmaStage, "8-frequency tiles stage by LDS-DMA");
    if (g0 < g1) {
        if constexpr (kSingleOps) {
            dma_stage2(-1, 0, g0, 0);                 // the first table row; the operands follow behind the barrier
        } else if constexpr (kDmaStage) {
            dma_stage(g0, 0);
        } else {
            issue_stage(g0);
            park(0);
        }
    }
    // (round 6, measured and not kept: the two blocks of a CU at different priorities -- by bit 8 of the block number
    // 4.92 -> 5.08 ms, by bit 0 no difference)
    FFK_MC_DECL();
    for (int g = g0; g < g1; ++g) {
        const int buf = (g - g0) & 1;
        [[maybe_unused]] const unsigned long long mc0 = FFK_MC_T();
        if constexpr (kDmaStage) lds_dma_wait();     // the copies of this segment's operands and table row
        __syncthreads();
        if constexpr (BF) {
            [[maybe_unused]] const unsigned long long mc1 = FFK_MC_T();
            // the staging loads of segment g + 1 fly during the generation, not the contraction,
            // whose accumulators, T entries and products leave no registers for them (d = 16)
            if constexpr (kSingleOps) {
                // this segment's operands into the ONE operand buffer (the contraction that read it ended before the
                // barrier above): in flight during the generation
                dma_stage2(g, 0, -1, 0);
            } else if (g + 1 < g1) {
                if constexpr (kDmaStage) dma_stage(g + 1, buf ^ 1);     // buffer buf ^ 1: last read before this barrier
                else issue_stage(g + 1);
            }
            [[maybe_unused]] const unsigned long long mc1b = FFK_MC_T();
            FFK_MC_ADD(6, mc1, mc1b);        // [6] of [1]: issue of the staging loads alone
#if !(defined(FFK_M_ABLATE) && FFK_M_ABLATE == 1)   /* diagnostic build 1: no generation */
            generate(kSingleOps ? 0 : buf);
#endif
            [[maybe_unused]] const unsigned long long mc2 = FFK_MC_T();
            if constexpr (!kDmaStage)
                if (g + 1 < g1) park(buf ^ 1);   // buffer buf ^ 1: last read before this barrier interval
            [[maybe_unused]] const unsigned long long mc3 = FFK_MC_T();
            if constexpr (kSingleOps) lds_dma_wait();
            __syncthreads();
            // (8-frequency tiles: ONE table row too -- the next segment's, behind the barrier that ends the generation,
            // in flight during the contraction: 76 KB of LDS per block, two blocks per CU)
            if constexpr (kSingleOps)
                if (g + 1 < g1) dma_stage2(-1, 0, g + 1, 0);
            [[maybe_unused]] const unsigned long long mc4 = FFK_MC_T();
#if !(defined(FFK_M_ABLATE) && FFK_M_ABLATE == 2)   /* diagnostic build 2: no contraction */
            if (active) contract_bf(buf);
#endif
            [[maybe_unused]] const unsigned long long mc5 = FFK_MC_T();
            FFK_MC_ADD(0, mc0, mc1);
            FFK_MC_ADD(1, mc1, mc2);
            FFK_MC_ADD(2, mc2, mc3);
            FFK_MC_ADD(3, mc3, mc4);
            FFK_MC_ADD(4, mc4, mc5);
            FFK_MC_ADD(5, 0ull, 1ull);
        } else {
#if !(defined(FFK_M_ABLATE) && FFK_M_ABLATE == 1)
            generate(buf);
#endif
            __syncthreads();
            if (g + 1 < g1) issue_stage(g + 1);
#if !(defined(FFK_M_ABLATE) && FFK_M_ABLATE == 2)
            if (active) contract(buf);
#endif
            if (g + 1 < g1) park(buf ^ 1);
        }
    }

    FFK_MC_FLUSH();
    if constexpr (BF) {
        if (ep.R != nullptr) {
            // One segment chunk: Y is complete here.  Operator by operator, the wavefronts that own it
            // lay their Y out in the (now free) tile as [entry i d + j][16 frequencies], and all
            // wavefronts expand it in the basis from there -- expand_lds_kernel's arithmetic (two
            // accumulators, alternating, summed at the end), thread = (frequency, basis element mod
            // nthreads/16) -- writing R[a, k, w] instead of Y.
            // Two operators per round (the launch reserves 2 x d^2 x 16 complex numbers of LDS for it:
            // everything in LDS is free by now), half the threads expand each.
            const int half = nthreads >> 1;
            const int hsel = tid >= half ? 1 : 0, th = tid - hsel*half;
            const int wl = th & (TW - 1), kl = th/TW, nk = half/TW;
            const int wo = blockIdx.x*TW + wl;
            for (int op0 = 0; op0 < na; op0 += 2) {
                __syncthreads();         // the tile's last readers (contraction / previous round) are done
                if (active && (alpha_l == op0 || alpha_l == op0 + 1)) {
                    cplx* yw = tile + (alpha_l - op0)*DD*TW;
#pragma unroll
                    for (int set = 0; set < NSET; ++set) {
                        const int f = 4*(jh*NSET + set) + (c >> 2);
#pragma unroll
                        for (int ig = 0; ig < NS; ++ig)
#pragma unroll
                            for (int jg = 0; jg < NS; ++jg)
                                yw[((4*ig + q)*D + 4*jg + c4)*TW + (f ^ ((4*c4) & (TW - 1)))] =
                                    {Yr[set*NS + ig][jg], Yi[set*NS + ig][jg]};
                    }
                }
                __syncthreads();
                const int a = alpha0 + op0 + hsel;
                if (op0 + hsel < na && a < alpha_end && wo < W) {
                    const cplx* yl = tile + hsel*DD*TW;
                    // frequency f of entry e sits in slot f ^ 4 (e & 3) (e & 3 = the column's low bits:
                    // a 16-lane row of the stores above then hits 16 distinct bank groups)
                    auto ysw = [&](int e) { return yl[e*TW + (wl ^ ((4*(e & 3)) & (TW - 1)))]; };
                    // four elements per trip, their list heads requested together (the lists sit in L2:
                    // one element at a time the loop was a chain of dependent trips, count -> row -> LDS)
                    constexpr int KU = 4;
                    for (int k0 = kl; k0 < ep.N; k0 += KU*nk) {
                        int nn[KU], r0[KU], r1[KU];
                        cplx v0[KU], v1[KU];
#pragma unroll
                        for (int u = 0; u < KU; ++u) {
                            const size_t k = min(k0 + u*nk, ep.N - 1);
                            nn[u] = ep.nnz[k];
                            r0[u] = ep.rows[k*DD];
                            r1[u] = ep.rows[k*DD + 1];
                            v0[u] = ep.vals[k*DD];
                            v1[u] = ep.vals[k*DD + 1];
                        }
#pragma unroll
                        for (int u = 0; u < KU; ++u) {
                            const int k = k0 + u*nk;
                            if (k >= ep.N) break;
                            const int n = nn[u];
                            const int* rk = ep.rows + static_cast<size_t>(k)*DD;
                            const cplx* vk = ep.vals + static_cast<size_t>(k)*DD;
                            cplx acc0 = {0.0, 0.0}, acc1 = {0.0, 0.0};
                            if (n > 0) cmac(acc0, v0[u], ysw(r0[u]));
                            if (n > 1) cmac(acc1, v1[u], ysw(r1[u]));
                            int qq = 2;
                            for (; qq + 1 < n; qq += 2) {
                                cmac(acc0, vk[qq], ysw(rk[qq]));
                                cmac(acc1, vk[qq + 1], ysw(rk[qq + 1]));
                            }
                            if (qq < n) cmac(acc0, vk[qq], ysw(rk[qq]));
                            ep.R[(static_cast<size_t>(a)*ep.N + k)*W + wo] = {acc0.re + acc1.re, acc0.im + acc1.im};
                        }
                    }
                }
            }
            return;
        }
        if (active) {
            cplx* out = Ypart + ((static_cast<size_t>(blockIdx.z)*A + alpha)*DD)*W;
#pragma unroll
            for (int set = 0; set < NSET; ++set) {
                const int iws = blockIdx.x*TW + 4*(jh*NSET + set) + (c >> 2);
                if (iws < W) {
#pragma unroll
                    for (int ig = 0; ig < NS; ++ig)
#pragma unroll
                        for (int jg = 0; jg < NS; ++jg)
                            out[static_cast<size_t>((4*ig + q)*D + 4*jg + c4)*W + iws] =
                                {Yr[set*NS + ig][jg], Yi[set*NS + ig][jg]};
                }
            }
        }
    } else if (active && iw < W) {
        cplx* out = Ypart + ((static_cast<size_t>(blockIdx.z)*A + alpha)*DD)*W + iw;
#pragma unroll
        for (int ig = 0; ig < NS; ++ig)
#pragma unroll
            for (int j = 0; j < 4*NJG; ++j)
                out[static_cast<size_t>((4*ig + q)*D + 4*jh*NJG + j)*W] = {Yr[ig][j], Yi[ig][j]};
    }
}

// ---------------------------------------------------------------------------------------------
// The block-frequency form with TWO OPERATORS per wavefront and tiles of EIGHT frequencies: wave
// (pair, set) contracts operators 2 pair and 2 pair + 1 on the four frequencies of its set.  The
// registers are those of the <D, 2, 8, true> kernel (two 4-frequency accumulator sets, T shared),
// but the generated tile now serves eight operators instead of four -- the tile generation was 14 %
// of that kernel (profiles/r03_l_*), regenerated by every row of operator blocks.
template <int D>
__global__ __launch_bounds__(512) void ctrl_accumulate_mfma4x2_kernel(
    const double* __restrict__ omega, int W, const double* __restrict__ segtab,
    const cplx* __restrict__ ops, int G, int A, int chunk_len, cplx* __restrict__ Ypart, int alpha_base,
    int alpha_end) {
    static_assert(D == 12 || D == 16, "d = 12, 16");
    constexpr int S = seg_stride(D), DD = D*D, NS = D/4;
    constexpr int NA = 8, TW = 8, TS = TW + 4, NT = 512;
    constexpr int kops = (1 + NA)*DD;
    constexpr int kStage = (kops + S/2 + NT - 1)/NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    cplx* tile = reinterpret_cast<cplx*>(lds_raw);
    cplx* opsb = tile + DD*TS;
    double* rows = reinterpret_cast<double*>(opsb + 2*kops);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4, c4 = c & 3, b = c >> 2;
    const int wq = c & 7, half = c >> 3;              // generation: frequency, half of the entries
    const int iw = blockIdx.x*TW + wq;
    const double om = omega[iw < W ? iw : W - 1];
    const int pair = wave >> 1, set = wave & 1;
    const int alpha0 = alpha_base + blockIdx.y*NA;    // the launch serves operators [alpha_base, alpha_end)
    const int n_alpha = min(NA, alpha_end - alpha0);
    const int n_ops = (1 + n_alpha)*DD;
    const int g0 = blockIdx.z*chunk_len;
    const int g1 = min(G, g0 + chunk_len);

    double Yr[2*NS][NS], Yi[2*NS][NS];                // [k*NS + ig][jg]: operator 2 pair + k
#pragma unroll
    for (int i = 0; i < 2*NS; ++i)
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            Yr[i][j] = 0.0;
            Yi[i][j] = 0.0;
        }

    cplx staged[kStage];
    auto issue_stage = [&](int g) {
        const cplx* src_ops = ops + static_cast<size_t>(g)*(1 + A)*DD;
        const cplx* src_tab = reinterpret_cast<const cplx*>(segtab + static_cast<size_t>(g)*S);
        int tid_o = tid;                 // (opaque: see ctrl_accumulate_mfma4_kernel::issue_stage)
        asm volatile("" : "+v"(tid_o));
#pragma unroll
        for (int k = 0; k < kStage; ++k) {
            const int e = tid_o + k*NT;
            const cplx* src = e < n_ops ? src_ops + (e < DD ? e : e + alpha0*DD) : src_tab + (e - n_ops);
            staged[k] = e < n_ops + S/2 ? *src : cplx{0.0, 0.0};
        }
    };
    auto park = [&](int buf) {
        cplx* dst_ops = opsb + buf*kops;
        cplx* dst_tab = reinterpret_cast<cplx*>(rows + buf*S);
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));
#pragma unroll
        for (int k = 0; k < kStage; ++k) {
            const int e = tid_o + k*NT;
            if (e < n_ops)
                dst_ops[e] = staged[k];
            else if (e < n_ops + S/2)
                dst_tab[e - n_ops] = staged[k];
        }
    };
    // thread (wave, q, c): frequency c & 7, entries 2 (4 wave + q) + (c >> 3) + 64 k
    auto generate = [&](int slot) {
        const double* st = rows + slot*S;
        const double dtg = st[0];
        cplx ph;
        sincos_pi<true>(om*st[1], &ph.im, &ph.re);
        double sa, ca;
        sincos_pi<true>(0.5*(om*dtg), &sa, &ca);
        const PhasedFrequency pf = phased_frequency(om, dtg, ph, sa, ca);
        for (int e = 2*(wave*4 + q) + half; e < DD; e += 64) {
            const double* r = st + seg_rec(e);
            tile[e*TS + wq] = phased_integral_aa(pf, r[0], r[1], r[2]);
        }
    };
    auto contract = [&](int buf) {
        const cplx* opT = opsb + buf*kops;
        cplx tq[NS][NS];                              // T[4 s + q][4 g + c4]
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int g = 0; g < NS; ++g) tq[s][g] = opT[(4*s + q)*D + 4*g + c4];
        const cplx* ecol = tile + 4*set + b;          // + slot*TS
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (alpha0 + 2*pair + k >= alpha_end) continue;       // (wave uniform)
            const cplx* opB = opT + (1 + 2*pair + k)*DD;
#pragma unroll
            for (int ng = 0; ng < NS; ++ng) {
                __builtin_amdgcn_sched_barrier(0);    // (as in the one-operator form: no hoisting across groups)
                double pr[NS], pi[NS];                // P[4 ng + q][4 ig + c4]
#pragma unroll
                for (int ig = 0; ig < NS; ++ig) {
                    pr[ig] = 0.0;
                    pi[ig] = 0.0;
                }
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int e = (4*s + q)*D + 4*ng + c4;
                    const cplx x = cmul(opB[e], ecol[e*TS]);      // X[4 s + q][4 ng + c4]
#pragma unroll
                    for (int ig = 0; ig < NS; ++ig) {
                        pr[ig] = __builtin_amdgcn_mfma_f64_4x4x4f64(x.re, tq[s][ig].re, pr[ig], 0, 0, 0);
                        pi[ig] = __builtin_amdgcn_mfma_f64_4x4x4f64(x.im, tq[s][ig].re, pi[ig], 0, 0, 0);
                    }
#pragma unroll
                    for (int ig = 0; ig < NS; ++ig) {
                        pr[ig] = __builtin_amdgcn_mfma_f64_4x4x4f64(x.im, tq[s][ig].im, pr[ig], 0, 0, 0);
                        pi[ig] = __builtin_amdgcn_mfma_f64_4x4x4f64(x.re, tq[s][ig].im, pi[ig], 0, 0, 1);
                    }
                }
#pragma unroll
                for (int ig = 0; ig < NS; ++ig) {
#pragma unroll
                    for (int jg = 0; jg < NS; ++jg) {
                        Yr[k*NS + ig][jg] = __builtin_amdgcn_mfma_f64_4x4x4f64(pr[ig], tq[ng][jg].re, Yr[k*NS + ig][jg], 0, 0, 0);
                        Yi[k*NS + ig][jg] = __builtin_amdgcn_mfma_f64_4x4x4f64(pr[ig], tq[ng][jg].im, Yi[k*NS + ig][jg], 0, 0, 0);
                    }
#pragma unroll
                    for (int jg = 0; jg < NS; ++jg) {
                        Yr[k*NS + ig][jg] = __builtin_amdgcn_mfma_f64_4x4x4f64(pi[ig], tq[ng][jg].im, Yr[k*NS + ig][jg], 0, 0, 1);
                        Yi[k*NS + ig][jg] = __builtin_amdgcn_mfma_f64_4x4x4f64(pi[ig], tq[ng][jg].re, Yi[k*NS + ig][jg], 0, 0, 0);
                    }
                }
            }
        }
    };

    if (g0 < g1) {
        issue_stage(g0);
        park(0);
    }
    for (int g = g0; g < g1; ++g) {
        const int buf = (g - g0) & 1;
        __syncthreads();
        if (g + 1 < g1) issue_stage(g + 1);           // (in flight during the generation, see the BF form)
        generate(buf);
        if (g + 1 < g1) park(buf ^ 1);
        __syncthreads();
        contract(buf);
    }

    const int iws = blockIdx.x*TW + 4*set + b;
    if (iws < W) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int alpha = alpha0 + 2*pair + k;
            if (alpha >= alpha_end) continue;
            cplx* out = Ypart + ((static_cast<size_t>(blockIdx.z)*A + alpha)*DD)*W + iws;
#pragma unroll
            for (int ig = 0; ig < NS; ++ig)
#pragma unroll
                for (int jg = 0; jg < NS; ++jg)
                    out[static_cast<size_t>((4*ig + q)*D + 4*jg + c4)*W] = {Yr[k*NS + ig][jg], Yi[k*NS + ig][jg]};
        }
    }
}

template <int D>
hipError_t launch_x2(const double* omega, int W, const double* segtab, const cplx* ops, int G, int A,
                     int chunks, int chunk_len, cplx* Ypart, hipStream_t stream, int alpha_end) {
    constexpr int DD = D*D, NA = 8, TW = 8;
    const int lds = static_cast<int>((static_cast<size_t>(DD)*(TW + 4) + 2*static_cast<size_t>(1 + NA)*DD)*sizeof(cplx) +
                                     2*static_cast<size_t>(seg_stride(D))*sizeof(double));
    auto kern = ctrl_accumulate_mfma4x2_kernel<D>;
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (err != hipSuccess) return err;
    const dim3 grid((W + TW - 1)/TW, alpha_end/NA, chunks);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, stream, omega, W, segtab, ops, G, A, chunk_len, Ypart, 0,
                       alpha_end);
    return hipGetLastError();
}

template <int D, int JH, bool BF = false, int TW = 16>
size_t mfma4_lds_bytes(int nw) {
    return (static_cast<size_t>(D*D)*mfma4_tile_stride(BF, TW) + (TW == 8 ? 1 : 2)*static_cast<size_t>(1 + nw/JH)*D*D)*sizeof(cplx) +
           (TW == 8 ? 1 : 2)*static_cast<size_t>(seg_stride(D))*sizeof(double);
}

template <int D, int JH, int MAXW = 8, bool BF = false, int TW = 16>
hipError_t launch_d4(const double* omega, int W, const double* segtab, const cplx* ops, int G, int A,
                     int chunks, int chunk_len, int nw, cplx* Ypart, hipStream_t stream,
                     int alpha_base = 0, int alpha_end = -1, const ExpandEpilogue* expand = nullptr) {
    auto kern = ctrl_accumulate_mfma4_kernel<D, JH, MAXW, BF, TW>;
    if (nw > MAXW) return hipErrorInvalidValue;
    const int lds = static_cast<int>(mfma4_lds_bytes<D, JH, BF, TW>(nw));
    // staging: (1 + na) d^2 + row/2 elements over nw*64 threads must fit kMaxStage per thread
    constexpr int kMaxStage = D == 16 ? (MAXW < 8 ? 8 : 4) : 8;
    if (nw % JH != 0 || (TW == 16 && (1 + nw/JH)*D*D + seg_stride(D)/2 > kMaxStage*nw*64)) return hipErrorInvalidValue;
    if (lds > 48*1024) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (err != hipSuccess) return err;
    }
    const int na = nw/JH;
    // operators [alpha_base, alpha_end) of the A the arrays are laid out for
    if (alpha_end < 0) alpha_end = A;
    const dim3 grid((W + TW - 1)/TW, (alpha_end - alpha_base + na - 1)/na, chunks);
    ExpandEpilogue ep = {};
    int lds_launch = lds;
    if (expand && BF) {
        ep = *expand;
        // the epilogue lays two operators' Y side by side: 2 d^2 x 16 complex numbers
        const int need = static_cast<int>(2*static_cast<size_t>(D)*D*TW*sizeof(cplx));
        if (need > lds_launch) {
            lds_launch = need;
            hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, lds_launch);
            if (err != hipSuccess) return err;
        }
    }
    if (std::getenv("FFK_DEBUG_OCCUPANCY")) {      // (the 8-frequency form is built on TWO blocks per CU: a table row more did not fit)
        int nb = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), nw*64, lds_launch);
        fprintf(stderr, "mfma4<D=%d, JH=%d, TW=%d>: %d threads, %d bytes of LDS -> %d blocks per CU\n", D, JH, TW, nw*64,
                lds_launch, nb);
    }
    hipLaunchKernelGGL(kern, grid, dim3(nw*64), lds_launch, stream, omega, W, segtab, ops, G, A, chunk_len,
                       nw, Ypart, alpha_base, alpha_end, ep);
    return hipGetLastError();
}

template <int D>
hipError_t launch_d(const double* omega, int W, const double* segtab, const cplx* ops, int G, int A,
                    int chunks, int chunk_len, cplx* Ypart, hipStream_t stream) {
    auto kern = ctrl_accumulate_mfma_kernel<D>;
    const int lds = static_cast<int>(MfmaLayout<D>::lds_bytes);
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (err != hipSuccess) return err;
    const dim3 grid((W + 15)/16, (A + kMW - 1)/kMW, chunks);
    hipLaunchKernelGGL(kern, grid, dim3(kMW*64), lds, stream, omega, W, segtab, ops, G, A, chunk_len,
                       Ypart);
    return hipGetLastError();
}

}  // namespace

// Kernel choice per dimension: 0 = the 16x16x4 kernel (one wavefront per operator), JH >= 1 = the
// 4x4x4 kernel with the columns of Y split over JH wavefronts per operator.  Measured (MI355X):
//   d = 16 (13 segments, 18 operators, 16384 omega): 16x16x4 6.7 ms, JH = 2 5.4 ms, JH = 4 6.5 ms
//   d = 12 (64 segments, 6 operators, 8192 omega):   16x16x4 3.4 ms, JH = 1 2.4 ms, JH = 3 3.3 ms
// -- two wavefronts per SIMD beat the wider tile.
// The block-frequency form of the 4x4x4 kernel serves d = 12, 16 (the frequency-on-the-columns form
// lost its A/B there, profiles/r03_l_*); with it JH counts the wavefronts that share an operator's
// 16 frequencies.
static bool mfma_block_frequency(int d) { return (d == 12 || d == 16) && FFK_MFMA_BF_DEFAULT; }
// round 6: the main launch of d = 12, 16 on 8-frequency tiles, four wavefronts per block, two blocks per CU
// (FFK_MFMA_TILE16: rounds 3-5's 16-frequency tiles, eight wavefronts per block, for A/B builds)
// Measured (profiles/r06_f_*): d = 16, 16 operators, 13 segments, 16384 omega 4.69 -> 4.39 ms without and 3.92 -> 3.83 ms
// with the expansion epilogue; d = 12 (64 segments, 6 operators, 8192 omega) 1.72 -> 1.74 ms: d = 16 only.
static bool mfma_tile8(int d) {
#if defined(FFK_MFMA_TILE16)
    return false;
#else
    return d == 16;
#endif
}

static int mfma_column_split(int d) {
    if (mfma_block_frequency(d)) return 2;
    return d == 16 ? 2 : 1;
}

// d = 4 and d = 8 are served on request only (ffk_set_accumulate_variant(4)): the A/B against the
// vector kernels (profiles/r02_c_*)
bool mfma_accumulate_supported(int d) { return d == 4 || d == 8 || d == 12 || d == 16; }
// waves (= noise operators) per block: 4 for the 16x16x4 kernel; for the 4x4x4 kernel (d = 8) the
// count in 3..8 with the fewest idle wave slots (ties: the larger, which shares the generated
// integral more widely)
int mfma_accumulate_ops_per_block(int d, int A) {
    const int jh = mfma_column_split(d);
    if (jh == 0) return kMW;
    if ((d == 8 || d == 4) && jh == 1) {
        // the count in 3..8 with the fewest idle wave slots (ties: the larger, which shares the
        // generated integral more widely)
        int best = 8, best_idle = 1 << 30;
        for (int nw = 8; nw >= 3; --nw) {
            const int idle = (A + nw - 1)/nw*nw - A;
            if (idle < best_idle) {
                best_idle = idle;
                best = nw;
            }
        }
        return std::min(best, std::max(A, 3));
    }
    if (d == 16 && jh == 4 && !mfma_block_frequency(d)) return 3;  // twelve wavefronts per block (tuning variant)
    if (d == 16 && jh == 1 && mfma_block_frequency(d)) return 4;   // one wavefront per SIMD, 512 registers
    return std::max(1, 8/jh);         // eight wavefronts per block
}
int mfma_accumulate_waves(int d, int A) {
    const int jh = mfma_column_split(d);
    return mfma_accumulate_ops_per_block(d, A)*(jh == 0 ? 1 : jh);
}
int mfma_accumulate_lds_bytes(int d, int nw) {
    const int jh = mfma_column_split(d);
    if (mfma_block_frequency(d)) {
#define FFK_BF_LDS(D, JH) if (d == D && jh == JH) return static_cast<int>(mfma4_lds_bytes<D, JH, true>(nw));
        FFK_BF_LDS(12, 1) FFK_BF_LDS(12, 2) FFK_BF_LDS(12, 4) FFK_BF_LDS(16, 1) FFK_BF_LDS(16, 2) FFK_BF_LDS(16, 4)
#undef FFK_BF_LDS
    }
    switch (d) {
        case 4:
            return static_cast<int>(mfma4_lds_bytes<4, 1>(nw));
        case 8:
            if (jh == 2) return static_cast<int>(mfma4_lds_bytes<8, 2>(nw));
            return static_cast<int>(mfma4_lds_bytes<8, 1>(nw));
        case 12:
            if (jh == 1) return static_cast<int>(mfma4_lds_bytes<12, 1>(nw));
            if (jh == 3) return static_cast<int>(mfma4_lds_bytes<12, 3>(nw));
            return static_cast<int>(MfmaLayout<12>::lds_bytes);
        case 16:
            if (jh == 1) return static_cast<int>(mfma4_lds_bytes<16, 1>(nw));
            if (jh == 2) return static_cast<int>(mfma4_lds_bytes<16, 2>(nw));
            if (jh == 4) return static_cast<int>(mfma4_lds_bytes<16, 4>(nw));
            return static_cast<int>(MfmaLayout<16>::lds_bytes);
        default: return 0;
    }
}

hipError_t launch_accumulate_mfma(const double* omega, int W, const double* segtab, const cplx* ops,
                                  int G, int d, int A, int chunks, int chunk_len, int nw,
                                  cplx* Ypart, hipStream_t stream, const ExpandEpilogue* expand,
                                  bool* expanded) {
    const int jh = mfma_column_split(d);
    // the expansion epilogue: block-frequency kernels only, and not where the eight-operator form
    // (d = 12, A >= 8) takes part of the operators
    const bool x2_applies = d == 12 && jh == 2 && nw == 8 && A >= 8;
    const ExpandEpilogue* ep = (expand && expand->R && chunks == 1 && mfma_block_frequency(d) && !x2_applies &&
                                (d == 16 ? jh == 2 || jh == 4 : true)) ? expand : nullptr;
    if (expanded) *expanded = ep != nullptr;
    if (mfma_block_frequency(d)) {
        // Two wavefronts per operator make blocks of four operators: one or two left over would
        // leave a whole row of blocks half or three quarters idle (config 5: 18 = 4 x 4 + 2).  They
        // get a launch of their own with four wavefronts per operator (one set of four frequencies
        // each), i.e. blocks of two operators (a single one leaves four of the eight wavefronts without a
        // contraction; they still generate): 4.77 -> 4.66 ms at config 5.
        constexpr bool split_rest = true;
        // Eight operators per tile first (two per wavefront, 8-frequency tiles), the block-of-four
        // form for what is left.  Measured (profiles/r03_s_*): d = 12, 8 operators 2.19 -> 2.09 ms;
        // d = 16, 18 operators 4.61 -> 5.04 ms (256 registers with 91 spilled, and nine 4-KiB operand
        // matrices staged per 8-frequency tile instead of five per 16) -- so the default is d = 12 only.
        const bool x2 = d == 12;
        int base = 0;
        if (x2 && jh == 2 && nw == 8 && A >= 8) {
            base = A/8*8;
            const hipError_t err = d == 16 ? launch_x2<16>(omega, W, segtab, ops, G, A, chunks, chunk_len, Ypart, stream, base)
                                           : launch_x2<12>(omega, W, segtab, ops, G, A, chunks, chunk_len, Ypart, stream, base);
            if (err != hipSuccess || base == A) return err;
        }
        const int rest = (A - base) % 4;
        if (jh == 2 && nw == 8 && (base > 0 || (split_rest && (rest == 1 || rest == 2)))) {
            const bool own_launch = split_rest && (rest == 1 || rest == 2);
            const int main_ops = own_launch ? A - rest : A;
#define FFK_BF_SPLIT(D) \
    if (d == D) { \
        if (main_ops > base) { \
            const hipError_t err = mfma_tile8(D) \
                ? launch_d4<D, 1, 4, true, 8>(omega, W, segtab, ops, G, A, chunks, chunk_len, 4, Ypart, stream, base, main_ops, ep) \
                : launch_d4<D, 2, 8, true>(omega, W, segtab, ops, G, A, chunks, chunk_len, 8, Ypart, stream, base, main_ops, ep); \
            if (err != hipSuccess || main_ops == A) return err; \
        } \
        return launch_d4<D, 4, 8, true>(omega, W, segtab, ops, G, A, chunks, chunk_len, 8, Ypart, stream, \
                                        main_ops, A, ep); \
    }
            FFK_BF_SPLIT(12) FFK_BF_SPLIT(16)
#undef FFK_BF_SPLIT
        }
    if (jh == 2 && nw == 8 && mfma_tile8(d)) {
        if (d == 12) return launch_d4<12, 1, 4, true, 8>(omega, W, segtab, ops, G, A, chunks, chunk_len, 4, Ypart, stream, 0, -1, ep);
        if (d == 16) return launch_d4<16, 1, 4, true, 8>(omega, W, segtab, ops, G, A, chunks, chunk_len, 4, Ypart, stream, 0, -1, ep);
    }
#define FFK_BF(D, JH) \
    if (d == D && jh == JH) \
        return launch_d4<D, JH, 8, true>(omega, W, segtab, ops, G, A, chunks, chunk_len, nw, Ypart, stream, 0, -1, ep);
        FFK_BF(12, 1) FFK_BF(12, 2) FFK_BF(12, 4) FFK_BF(16, 2) FFK_BF(16, 4)
#undef FFK_BF
        if (d == 16 && jh == 1)
            return launch_d4<16, 1, 4, true>(omega, W, segtab, ops, G, A, chunks, chunk_len, nw, Ypart, stream, 0, -1, ep);
    }
#define FFK_M4(D, JH) \
    if (d == D && jh == JH) \
        return launch_d4<D, JH>(omega, W, segtab, ops, G, A, chunks, chunk_len, nw, Ypart, stream);
    FFK_M4(4, 1) FFK_M4(8, 1) FFK_M4(8, 2) FFK_M4(12, 1) FFK_M4(12, 3) FFK_M4(16, 1) FFK_M4(16, 2)
#undef FFK_M4
    if (d == 16 && jh == 4)
        return launch_d4<16, 4, 12>(omega, W, segtab, ops, G, A, chunks, chunk_len, nw, Ypart, stream);
    switch (d) {
        case 12: return launch_d<12>(omega, W, segtab, ops, G, A, chunks, chunk_len, Ypart, stream);
        case 16: return launch_d<16>(omega, W, segtab, ops, G, A, chunks, chunk_len, Ypart, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ffk

#ifdef FFK_MFMA_CLOCK
// (tuning build only, not in include/ffk.h) phase sums since the last reset; reset != 0 clears them
extern "C" int ffk_debug_mfma_phases(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ffk::g_mfma_phase), sizeof(unsigned long long)*8) != hipSuccess) return 1;
    if (reset) {
        const unsigned long long zero[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(ffk::g_mfma_phase), zero, sizeof zero) != hipSuccess) return 1;
    }
    return 0;
}
#endif
