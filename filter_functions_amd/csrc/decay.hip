// decay.hip -- K7: decay amplitudes and cumulant function.
//
// Decay amplitudes, numeric.calculate_decay_amplitudes (filter_functions/numeric.py:1194-1337)
// with the 'generalized' integrand of _get_integrand (:310-374) and util.integrate (util.py:880-906):
//     Gamma[g,h,a,b,k,l] = int dw/2pi  Re( R*[g,a,k,w] S_ab(w) R[h,b,l,w] ).
// The reference materialises the (A, d^2, d^2, W) integrand and integrates it; here the integral
// IS a matrix product over the frequency axis.  Viewing a complex row R[a,k,:] as 2W interleaved
// reals, Re(x* y) summed over w is the plain dot product of the two real rows, so
//     Gamma_ab = Lreal (d^2 x 2W)  .  ((w S_ab/2pi) o R_b)real^T (2W x d^2),
// w = trapezoid weights: a real FP64 GEMM with K = 2W on v_mfma_f64_16x16x4_f64.  Each lane
// fetches 4 consecutive frequencies (64 contiguous bytes) of its row; the k index of an MFMA is
// free to permute as long as both operands agree, so no transposition or LDS staging is needed.
// The frequency axis is split over blocks (split-K) and reduced in fixed order.
//
// Cumulant function, numeric.calculate_cumulant_function (:957-1191, first order):
//     K_ij = -1/2 sum_kl Gamma_kl (T_klji - T_kjli - T_kilj + T_kijl),  T = 4-element traces,
// evaluated without the N^4 trace tensor: with D_k = sum_l Gamma_kl C_l the cumulant
// superoperator is  K(X) = -1/2 sum_k (C_k D_k X - C_k X D_k - D_k X C_k + X D_k C_k)  and
// K_ij = tr(C_i K(C_j)): four small complex GEMMs (O(d^6)) per noise-operator pair.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <utility>

#include "ffk_internal.h"
#include "ffk_mfma_util.h"

namespace ffk {
namespace {

using f64x4 = __attribute__((ext_vector_type(4))) double;

// scale[row][w] = trapezoid weight(w_offset + w) * S_row(w) / (2 pi); the weights are those of the
// GLOBAL grid omega (Wg,), of which this call integrates the block [w_offset, w_offset + W)
// (rows of `scale` are `stride` >= W apart; the entries from W up to the stride are zero)
__global__ __launch_bounds__(256) void spectral_weights_kernel(const cplx* __restrict__ S, int rows,
                                                               int W,
                                                               const double* __restrict__ omega,
                                                               int Wg, int w_offset,
                                                               cplx* __restrict__ scale, int stride,
                                                               int* __restrict__ complex_weights) {
    const int w = blockIdx.x*256 + threadIdx.x;
    if (w >= W) {
        if (w < stride)
            for (int r = blockIdx.y; r < rows; r += gridDim.y) scale[static_cast<size_t>(r)*stride + w] = {0.0, 0.0};
        return;
    }
    const int gw = w_offset + w;
    const double lo = gw > 0 ? omega[gw] - omega[gw - 1] : 0.0;
    const double hi = gw < Wg - 1 ? omega[gw + 1] - omega[gw] : 0.0;
    const double wgt = 0.5*(lo + hi)/(2.0*3.141592653589793);
    bool any_imag = false;
    for (int r = blockIdx.y; r < rows; r += gridDim.y) {
        const cplx s = S[static_cast<size_t>(r)*W + w];
        scale[static_cast<size_t>(r)*stride + w] = {s.re*wgt, s.im*wgt};
        any_imag |= s.im != 0.0;
    }
    if (complex_weights && any_imag) atomicOr(complex_weights, 1);
}

// XCD-aware block order.  Workgroups are dealt round robin over the 8 XCDs (block b and b + 8 share an
// L2); the tiles of one (operator pair, frequency chunk) read the same row strips of R, so they should
// meet in ONE L2.  Linear block id L -> virtual id v = (L mod 8) * (total / 8) + L / 8: the blocks of
// one XCD work through consecutive virtual ids, i.e. through the tiles of the same strips, at about
// the same time.  (Placement is a speed matter only; any bijection is correct.)
__device__ __forceinline__ void xcd_block(unsigned& bx, unsigned& by) {
    const unsigned total = gridDim.x*gridDim.y, per = total/8;
    const unsigned L = blockIdx.x + gridDim.x*blockIdx.y;
    const unsigned v = L < 8*per ? (L % 8)*per + L/8 : L;
    bx = v % gridDim.x;
    by = v / gridDim.x;
}

#ifdef FFK_DG_CLOCK   /* tuning build: which clock does the chip hold inside the decay GEMM? */
__device__ unsigned long long g_dg_clock[3];
#define FFK_DG_CLOCK_BEGIN \
    const unsigned long long dg_c0 = __builtin_amdgcn_s_memtime(), dg_r0 = __builtin_amdgcn_s_memrealtime();
__device__ unsigned long long g_dg_trace[3*8192];   // per block: start, end (100 MHz ticks), HW_ID | XCC_ID << 32
#define FFK_DG_CLOCK_END \
    if (threadIdx.x == 0) { \
        const unsigned long long dg_r1 = __builtin_amdgcn_s_memrealtime(); \
        atomicAdd(&g_dg_clock[0], __builtin_amdgcn_s_memtime() - dg_c0); \
        atomicAdd(&g_dg_clock[1], dg_r1 - dg_r0); \
        atomicAdd(&g_dg_clock[2], 1ull); \
        const unsigned dg_l = blockIdx.x + gridDim.x*blockIdx.y; \
        if (dg_l < 8192) { \
            g_dg_trace[3*dg_l] = dg_r0; \
            g_dg_trace[3*dg_l + 1] = dg_r1; \
            g_dg_trace[3*dg_l + 2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | \
                                     (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11))) << 32); \
        } \
    }
#else
#define FFK_DG_CLOCK_BEGIN
#define FFK_DG_CLOCK_END
#endif

// One wavefront: a (16 TM) x (16 TN) tile of one Gamma[g,h,a,b] over the block's frequency range.
// MFMA operand maps: A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15],
// D[row = (lane>>4) + 4 r][col = lane&15] (cdna_hip_programming.md section 3).
template <int TM, int TN>
__global__ __launch_bounds__(64) void decay_gemm_kernel(
    const cplx* __restrict__ R, int Gp, int A, int N, int W, const cplx* __restrict__ scale,
    int s_ndim, const int32_t* __restrict__ idx, int n_idx, int kchunk, int tiles_m, int tiles_n,
    double* __restrict__ out, size_t split_stride, int mirror_in_store,
    const int* __restrict__ complex_weights, int tri, int scale_stride, int only_complex_weights) {
    // only_complex_weights: the symmetric case is served by decay_gemm_sym256_kernel; this launch works only when the
    // weights turned out complex (every block of the launch alike)
    if (only_complex_weights && *complex_weights == 0) return;
    const int lane = threadIdx.x;
    const int l15 = lane & 15, lk = lane >> 4;
    // tri: which tiles of a Gamma block the grid enumerates -- 0 all tiles_m x tiles_n, 1 those on
    // and above the diagonal, 2 those strictly below it (square tiles).  A grid must not contain
    // blocks that return at once next to blocks that work: the dispatcher hands blocks to wavefront
    // slots in strict rotation, and the slots that received an empty block stay empty until the
    // rotation comes round again.  With all 16 tiles of config 5's 256 x 256 blocks in one grid and
    // the six below the diagonal returning, 80 of an XCD's 128 slots (= 10/16) held a wavefront at
    // any time and the kernel took 1.63 ms instead of 0.9 (profiles/r03_m_*).
    const int tiles = tri == 0 ? tiles_m*tiles_n : (tri == 1 ? tiles_m*(tiles_m + 1)/2 : tiles_m*(tiles_m - 1)/2);
    unsigned bx, by;
    xcd_block(bx, by);
    const int tile = bx % tiles;
    const int z0 = bx / tiles;
    int z = z0;
    int ti, tj;
    if (tri == 0) {
        ti = tile / tiles_n;
        tj = tile % tiles_n;
    } else if (tri == 1) {
        ti = 0;
        int rem = tile;
        while (rem >= tiles_m - ti) {
            rem -= tiles_m - ti;
            ++ti;
        }
        tj = ti + rem;
    } else {
        ti = 1;
        int rem = tile;
        while (rem >= ti) {
            rem -= ti;
            ++ti;
        }
        tj = rem;
    }
    const int nb = s_ndim == 3 ? n_idx : 1;
    const int ib0 = z % nb;
    z /= nb;
    const int ia = z % n_idx;
    z /= n_idx;
    const int h = z % Gp, g = z / Gp;
    const int ib = s_ndim == 3 ? ib0 : ia;
    // Gamma_aa of one pulse with itself is symmetric in (k, l) for real weights: only the tiles on
    // and above the diagonal are computed, the rest mirrored at the store.  (A complex spectrum of
    // one or two dimensions adds an antisymmetric part: no shortcut then.)
    const bool symmetric = s_ndim != 3 && g == h && *complex_weights == 0;
    // (rectangular tiles, TN < TM: the mirror granule is the tile's ROW count, 16 TM)
    const int tjg = (tj*TN)/TM;              // row-granule index of this tile's columns
    if (symmetric && ti > tjg) return;
    const int srow = s_ndim == 1 ? 0 : (s_ndim == 2 ? ia : ia*n_idx + ib);
    const cplx* sp = scale + static_cast<size_t>(srow)*scale_stride;
    const cplx* Lp[TM];
    const cplx* Rp[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int row = min(N - 1, (ti*TM + tm)*16 + l15);
        Lp[tm] = R + ((static_cast<size_t>(g)*A + idx[ia])*N + row)*W;
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int row = min(N - 1, (tj*TN + tn)*16 + l15);
        Rp[tn] = R + ((static_cast<size_t>(h)*A + idx[ib])*N + row)*W;
    }
    f64x4 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = {0.0, 0.0, 0.0, 0.0};

    const int wbeg = by*kchunk;
    const int wend = min(W, wbeg + kchunk);
    FFK_DG_CLOCK_BEGIN
    for (int w0 = wbeg; w0 < wend; w0 += 16) {
        cplx a[TM][4], b[TN][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int w = w0 + 4*lk + c;
            const bool ok = w < wend;
            const int wc = ok ? w : wend - 1;
            cplx s = sp[wc];
            if (!ok) s = {0.0, 0.0};
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) a[tm][c] = Lp[tm][wc];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) b[tn][c] = cmul(s, Rp[tn][wc]);
        }
        // real parts of all tiles, then imaginary parts: consecutive MFMAs never share an
        // accumulator (a dependent v_mfma_f64_16x16x4 waits out the 16 passes of its predecessor)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tm][c].re, b[tn][c].re,
                                                                       acc[tm][tn], 0, 0, 0);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tm][c].im, b[tn][c].im,
                                                                       acc[tm][tn], 0, 0, 0);
        }
    }
    FFK_DG_CLOCK_END
    double* o = out + static_cast<size_t>(by)*split_stride + static_cast<size_t>(z0)*N*N;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int col = (tj*TN + tn)*16 + l15;
            if (col >= N) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = (ti*TM + tm)*16 + lk + 4*r;
                if (row < N) {
                    o[static_cast<size_t>(row)*N + col] = acc[tm][tn][r];
                    if (mirror_in_store && symmetric && ti < tjg)
                        o[static_cast<size_t>(col)*N + row] = acc[tm][tn][r];
                }
            }
        }
}

// (Round 3's LDS-staged 128 x 128 variant of this product -- half the operand traffic, but 1.20 ms
// against 1.07 ms at config 5 -- was removed in round 4; the A/B is profiles/r03_e_*, r03_m_*.)

// ---- Round 6: one pulse with itself, N = 256 (d = 16, full basis), real weights: a workgroup per (operator,
// frequency chunk) computes the whole symmetric 256 x 256 block from ONE copy of the operator's rows in LDS.
//   * Both operands of Gamma_aa = R_a diag(s) R_a^T are the same 256 rows: per step of 16 frequencies the block
//     brings 256 x 16 complex (64 KB) + the 16 weights into LDS by global_load_lds_dwordx4 (no register, no ds_write:
//     the ds_write_b128 of round 3's LDS-staged variant cost a wavefront 35-45 issue cycles each, tools/
//     lds_issue_probe.py) -- 1.2 GB per call at config 5 instead of the 6.0 GB the register-fed kernel pulls through
//     L2 -- double-buffered, one barrier per step.
//   * Ten 64 x 64 tiles lie on or above the diagonal; eight wavefronts (two per SIMD, 256 registers) own one each
//     and split the other two into 16 x 64 strips, one per wavefront: 20 products per wavefront and frequency quad,
//     the same on every SIMD (ten wavefronts of one tile each would load the SIMDs 3, 3, 2, 2).
//   * LDS image: row r holds its 16 frequencies as 16-byte slots, frequency j in slot j ^ (r & 15) -- the copy
//     chooses which frequency a lane fetches, the destination of a lane is fixed --, so that the 16 lanes of a
//     matrix-instruction operand (16 consecutive rows, one frequency) read 16 different slots.
// The partial sums go to the split-K planes the reduction kernel (with its mirror of the lower tiles) already reads.
constexpr int kSymN = 256;
constexpr int kSymStep = 16;
constexpr int kSymTile = (kSymN + 1)*kSymStep;                      // complex entries: 256 rows + the weights
constexpr size_t kSymLdsBytes = 2*sizeof(cplx)*kSymTile;           // 131,584 B

__global__ __launch_bounds__(512) void decay_gemm_sym256_kernel(
    const cplx* __restrict__ R, int W, const cplx* __restrict__ scale, int scale_stride, int s_ndim,
    const int32_t* __restrict__ idx, double* __restrict__ out, size_t split_stride,
    const int* __restrict__ complex_weights) {
    if (*complex_weights != 0) return;                 // (every block alike: the general kernel serves this call)
    extern __shared__ __attribute__((aligned(16))) unsigned char sym_lds[];
    cplx* const tiles = reinterpret_cast<cplx*>(sym_lds);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lk = lane >> 4;
    const int ia = blockIdx.x, split = blockIdx.y;
    const cplx* sp = scale + static_cast<size_t>(s_ndim == 1 ? 0 : ia)*scale_stride;
    const cplx* Ra = R + static_cast<size_t>(idx[ia])*kSymN*W;
    // this workgroup's share of the ceil(W / 16) steps: as even as whole steps allow (the chunk length the general
    // kernel is given would leave the last workgroup of an operator 992 of 1184 frequencies at config 5 -- the planes
    // are summed, their boundaries need not agree between the two kernels)
    const int all_steps = (W + kSymStep - 1)/kSymStep, nsplit = gridDim.y;
    const int first_step = static_cast<int>(static_cast<long>(all_steps)*split/nsplit);
    const int steps = static_cast<int>(static_cast<long>(all_steps)*(split + 1)/nsplit) - first_step;
    const int wbeg = first_step*kSymStep;

    // the copy: wavefront `wave` brings rows 32 wave .. + 31, four rows per instruction (lane -> row 4 q + lk, slot
    // l15, i.e. frequency l15 ^ (row & 15)); wavefront 0 also the weights (16 lanes)
    const int row0 = 32*wave + lk;
    const cplx* src_row = Ra + static_cast<size_t>(row0)*W;
    auto copy_step = [&](int step, int buf) {
        cplx* T = tiles + buf*kSymTile;
        const int w0 = wbeg + step*kSymStep;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int w = min(w0 + (l15 ^ ((4*q + lk) & 15)), W - 1);       // (beyond W: the weight is zero)
            lds_dma16(src_row + static_cast<size_t>(4*q)*W + w, T + (32*wave + 4*q)*kSymStep);
        }
        if (wave == 0 && lane < 16) lds_dma16(sp + w0 + lane, T + kSymN*kSymStep);
    };

    // tiles on and above the diagonal, row-major: wavefront j owns the j-th of the first eight; (2,3) goes in strips
    // to wavefronts 0..3, (3,3) to wavefronts 4..7
    const int ti = wave < 4 ? 0 : (wave < 7 ? 1 : 2);
    const int tj = wave < 4 ? wave : (wave < 7 ? wave - 3 : 2);
    const int strip_row = (wave < 4 ? 2 : 3)*64 + (wave & 3)*16;
    f64x4 acc[4][4], strip[4];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) strip[tn] = {0.0, 0.0, 0.0, 0.0};

    FFK_DG_CLOCK_BEGIN
    copy_step(0, 0);
    for (int step = 0; step < steps; ++step) {
        lds_dma_wait();
        __syncthreads();                 // this step's tile has landed; everybody is done with the other buffer
        if (step + 1 < steps) copy_step(step + 1, (step + 1) & 1);
        const cplx* T = tiles + (step & 1)*kSymTile;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int slot = (4*lk + c) ^ l15;
            const double s = T[kSymN*kSymStep + 4*lk + c].re;
            cplx a[4], b[4], x[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                a[t] = T[(ti*64 + t*16 + l15)*kSymStep + slot];
                const cplx y = T[(tj*64 + t*16 + l15)*kSymStep + slot];
                b[t] = {s*y.re, s*y.im};
            }
            const cplx as = T[(strip_row + l15)*kSymStep + slot];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const cplx y = T[(3*64 + t*16 + l15)*kSymStep + slot];
                x[t] = {s*y.re, s*y.im};
            }
            // real parts of all tiles, then imaginary parts: consecutive instructions never share an accumulator
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tm].re, b[tn].re, acc[tm][tn], 0, 0, 0);
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) strip[tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(as.re, x[tn].re, strip[tn], 0, 0, 0);
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tm].im, b[tn].im, acc[tm][tn], 0, 0, 0);
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) strip[tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(as.im, x[tn].im, strip[tn], 0, 0, 0);
        }
    }
    FFK_DG_CLOCK_END
    double* o = out + static_cast<size_t>(split)*split_stride + static_cast<size_t>(ia)*kSymN*kSymN;
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                o[static_cast<size_t>(ti*64 + tm*16 + lk + 4*r)*kSymN + tj*64 + tn*16 + l15] = acc[tm][tn][r];
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            o[static_cast<size_t>(strip_row + lk + 4*r)*kSymN + 3*64 + tn*16 + l15] = strip[tn][r];
}

// out[i] = sum_s part[s][i], fixed order.  tile > 0: batches of a pulse with itself (g == h in the
// batch index (g*Gp + h)*n_idx + a) are symmetric matrices of which only the tiles (of `tile`
// rows/columns) on or above the diagonal were computed; the rest is read transposed.
__global__ __launch_bounds__(256) void reduce_splits_kernel(const double* __restrict__ part,
                                                            int nsplit, size_t n, int N, int tile,
                                                            int Gp, int n_idx,
                                                            const int* __restrict__ complex_weights,
                                                            double* __restrict__ out) {
    const size_t i = static_cast<size_t>(blockIdx.x)*256 + threadIdx.x;
    if (i >= n) return;
    // (round 6: the entry above the diagonal is summed once, read along rows, and stored twice; before, the thread of
    // the entry below summed the planes again, reading them down a column -- 8 bytes per 2-KiB line, nsplit times)
    size_t twin = i;
    if (tile > 0 && *complex_weights == 0) {
        const size_t b = i / (static_cast<size_t>(N)*N);
        const int row = static_cast<int>((i / N) % N), col = static_cast<int>(i % N);
        const size_t gh = b / n_idx;
        if (gh / Gp == gh % Gp) {
            if (row / tile > col / tile) return;
            if (row / tile < col / tile) twin = (b*N + col)*N + row;
        }
    }
    double acc = part[i];
    for (int s = 1; s < nsplit; ++s) acc += part[static_cast<size_t>(s)*n + i];
    out[i] = acc;
    if (twin != i) out[twin] = acc;
}

struct DecayPlan {
    int tm, tn, tiles_m, tiles_n, ksplit, kchunk;
    size_t batch;
    bool tri;       // register-fed kernel: one grid for the tiles on and above the diagonal, one for the rest
    bool sym256;    // decay_gemm_sym256_kernel serves the call when the weights are real
};

DecayPlan decay_plan(int Gp, int N, int W, int n_idx, int s_ndim) {
    DecayPlan p;
    p.tri = false;
    const int t = N <= 16 ? 1 : (N < 128 ? 2 : 4);
    p.tm = p.tn = t;
    // N >= 128: 64 x 64 tiles per wavefront (one wavefront per SIMD: 128 accumulator + 128 operand
    // registers).  (64 x 32 tiles, two wavefronts per SIMD, read 9.0 instead of 6.0 GB of operands per
    // call at config 5, where the square tiles already run at the rate the operands are delivered:
    // profiles/r03_e_*, r03_m_*.)
    p.tiles_m = (N + 16*p.tm - 1)/(16*p.tm);
    p.tiles_n = (N + 16*p.tn - 1)/(16*p.tn);
    p.batch = static_cast<size_t>(Gp)*Gp*n_idx*(s_ndim == 3 ? n_idx : 1);
    // one pulse with itself, spectrum of one or two dimensions: Gamma is symmetric for real weights
    // and only the tiles on and above the diagonal work (see the kernel's `tri`)
    p.tri = p.tm == p.tn && s_ndim != 3 && Gp == 1;
    const size_t waves = p.batch*(p.tri ? static_cast<size_t>(p.tiles_m)*(p.tiles_m + 1)/2
                                        : static_cast<size_t>(p.tiles_m)*p.tiles_n);
    // Split the frequency axis so that the working wavefronts fill whole rounds of the chip's
    // wavefront slots (64 x 64 tiles: 332 registers, one per SIMD; 32 x 32: three; 16 x 16: six), at
    // least 64 frequencies per split.  Cost in steps of 16 frequencies: rounds x (steps per
    // wavefront + its prologue / epilogue) + one step's worth per split for the reduction kernel.
    const size_t slots = static_cast<size_t>(device_cu_count())*4*(t == 4 ? (p.tn == 4 ? 1 : 2) : (t == 2 ? 3 : 6));
    const int max_split = std::max(1, (W + 63)/64);
    long best = -1;
    int want = 1;
    for (int split = 1; split <= max_split; ++split) {
        const int chunk = ((W + split - 1)/split + 15)/16*16;
        const int nsplit = (W + chunk - 1)/chunk;
        const long rounds = static_cast<long>((waves*nsplit + slots - 1)/slots);
        const long cost = rounds*(chunk/16 + 8) + (nsplit > 1 ? nsplit : 0);
        if (best < 0 || cost < best) {
            best = cost;
            want = split;
        }
    }
    p.kchunk = static_cast<int>(((W + want - 1)/want + 15)/16*16);
    p.ksplit = (W + p.kchunk - 1)/p.kchunk;
    // One pulse with itself at N = 256: a workgroup per (operator, chunk), one workgroup per CU -- as many chunks as
    // fill the chip once (config 5: 18 operators x 14 = 252 workgroups), at least 128 frequencies each, at least two
    // (the lower tiles are mirrored by the reduction over the chunks).  Below a chip's worth of such chunks the
    // register-fed kernel, whose unit is a wavefront and 64 frequencies, spreads the work better (3 operators x 1030
    // frequencies: 0.19 ms here).
    const long cus = device_cu_count();
    p.sym256 = p.tri && N == kSymN && p.batch <= 65535 && static_cast<long>(p.batch)*(W/128) >= cus &&
               std::getenv("FFK_DECAY_REGISTER_FED") == nullptr;
    if (p.sym256) {
        long want_split = std::max<long>(2, cus/static_cast<long>(p.batch));
        want_split = std::min<long>(want_split, std::max(2, W/128));
        p.kchunk = static_cast<int>(((W + want_split - 1)/want_split + 15)/16*16);
        p.ksplit = (W + p.kchunk - 1)/p.kchunk;
        p.sym256 = p.ksplit >= 2;
    }
    return p;
}

// ---- cumulant function ------------------------------------------------------------------------
// Generic strided GEMM for the small products: C[b][m][n] = sum_k A[b][m][k] B[b][k][n]
// (element strides; complex operands unless a_real; real part only if c_real).
struct GemmDesc {
    int M, N, K;
    long sAm, sAk, sAb;
    long sBk, sBn, sBb;
    long sCm, sCn, sCb;
    int a_real, c_real;
};

// 16 x 16 outputs per block, one per thread; the k axis goes through LDS in tiles of 64: four
// independent (strided, often uncoalesced) global loads per operand and thread are in flight per
// tile instead of one -- with 16-wide tiles the 256^3 products of the d = 16 cumulant function took
// 122 us each, bound by sixteen dependent load round trips
constexpr int kGemmKT = 64;
__global__ __launch_bounds__(256) void gemm_small_kernel(const double* __restrict__ A,
                                                         const double* __restrict__ B,
                                                         double* __restrict__ C, GemmDesc g,
                                                         const int* __restrict__ skip_if_set) {
    if (skip_if_set && *skip_if_set) return;          // (every block of the launch alike)
    __shared__ cplx As[16][kGemmKT + 1];
    __shared__ cplx Bs[kGemmKT][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m = blockIdx.y*16 + ty, n = blockIdx.x*16 + tx;
    const long b = blockIdx.z;
    const double* Ab = A + (g.a_real ? 1 : 2)*b*g.sAb;
    const cplx* Bb = reinterpret_cast<const cplx*>(B) + b*g.sBb;
    cplx acc = {0.0, 0.0};
    for (int k0 = 0; k0 < g.K; k0 += kGemmKT) {
        // A tile: row m (ty), columns k0 + tx + 16 q;  B tile: rows k0 + ty + 16 q, column n (tx)
        cplx av[kGemmKT/16], bv[kGemmKT/16];
#pragma unroll
        for (int q = 0; q < kGemmKT/16; ++q) {
            av[q] = {0.0, 0.0};
            bv[q] = {0.0, 0.0};
            const int ka = k0 + tx + 16*q, kb = k0 + ty + 16*q;
            if (m < g.M && ka < g.K) {
                const long o = m*g.sAm + ka*g.sAk;
                if (g.a_real)
                    av[q] = {Ab[o], 0.0};
                else
                    av[q] = reinterpret_cast<const cplx*>(Ab)[o];
            }
            if (kb < g.K && n < g.N) bv[q] = Bb[kb*g.sBk + n*g.sBn];
        }
#pragma unroll
        for (int q = 0; q < kGemmKT/16; ++q) {
            As[ty][tx + 16*q] = av[q];
            Bs[ty + 16*q][tx] = bv[q];
        }
        __syncthreads();
#pragma unroll 16
        for (int k = 0; k < kGemmKT; ++k) cmac(acc, As[ty][k], Bs[k][tx]);
        __syncthreads();
    }
    if (m < g.M && n < g.N) {
        const long o = b*g.sCb + m*g.sCm + n*g.sCn;
        if (g.c_real)
            C[o] = acc.re;
        else
            reinterpret_cast<cplx*>(C)[o] = acc;
    }
}

// 32 x 32 outputs per block of 256 threads, 2 x 2 per thread (rows ty, ty + 16; columns tx, tx + 16):
// four products per four LDS operand reads -- with one output per thread (rounds 1-2) it was two reads
// per product and the LDS pipe set the pace (22 TFLOP/s on the 256^3 products of the d = 16 cumulant
// function).  The k axis goes through LDS in tiles of 32: two independent (strided, often
// uncoalesced) global loads per operand row/column and thread are in flight per tile.
constexpr int kGemmKT2 = 32;
__global__ __launch_bounds__(256) void gemm_small_2x2_kernel(const double* __restrict__ A,
                                                         const double* __restrict__ B,
                                                         double* __restrict__ C, GemmDesc g,
                                                         const int* __restrict__ skip_if_set) {
    if (skip_if_set && *skip_if_set) return;
    __shared__ cplx As[32][kGemmKT2 + 1];
    __shared__ cplx Bs[kGemmKT2][33];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.y*32, n0 = blockIdx.x*32;
    const long b = blockIdx.z;
    const double* Ab = A + (g.a_real ? 1 : 2)*b*g.sAb;
    const cplx* Bb = reinterpret_cast<const cplx*>(B) + b*g.sBb;
    cplx acc[2][2] = {{{0.0, 0.0}, {0.0, 0.0}}, {{0.0, 0.0}, {0.0, 0.0}}};
    for (int k0 = 0; k0 < g.K; k0 += kGemmKT2) {
        // A tile: rows m0 + ty + 16 r, columns k0 + tx + 16 q;  B tile: rows k0 + ty + 16 q, columns n0 + tx + 16 r
        cplx av[2][kGemmKT2/16], bv[kGemmKT2/16][2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int q = 0; q < kGemmKT2/16; ++q) {
                av[r][q] = {0.0, 0.0};
                bv[q][r] = {0.0, 0.0};
                const int m = m0 + ty + 16*r, ka = k0 + tx + 16*q;
                if (m < g.M && ka < g.K) {
                    const long o = m*g.sAm + ka*g.sAk;
                    if (g.a_real)
                        av[r][q] = {Ab[o], 0.0};
                    else
                        av[r][q] = reinterpret_cast<const cplx*>(Ab)[o];
                }
                const int kb = k0 + ty + 16*q, n = n0 + tx + 16*r;
                if (kb < g.K && n < g.N) bv[q][r] = Bb[kb*g.sBk + n*g.sBn];
            }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int q = 0; q < kGemmKT2/16; ++q) {
                As[ty + 16*r][tx + 16*q] = av[r][q];
                Bs[ty + 16*q][tx + 16*r] = bv[q][r];
            }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < kGemmKT2; ++k) {
            const cplx a0 = As[ty][k], a1 = As[ty + 16][k], b0 = Bs[k][tx], b1 = Bs[k][tx + 16];
            cmac(acc[0][0], a0, b0);
            cmac(acc[0][1], a0, b1);
            cmac(acc[1][0], a1, b0);
            cmac(acc[1][1], a1, b1);
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int m = m0 + ty + 16*r, n = n0 + tx + 16*c;
            if (m < g.M && n < g.N) {
                const long o = b*g.sCb + m*g.sCm + n*g.sCn;
                if (g.c_real)
                    C[o] = acc[r][c].re;
                else
                    reinterpret_cast<cplx*>(C)[o] = acc[r][c];
            }
        }
}

hipError_t launch_gemm_small(const void* A, const void* B, void* C, const GemmDesc& g, int batch,
                             hipStream_t stream, const int* skip_if_set = nullptr) {
    // the 2 x 2 form when its 32 x 32 blocks still fill the chip twice over (d = 16 cumulant function:
    // 0.45 -> 0.33 ms), else one output per thread (d = 12, 6 operators: 150 blocks of 32 x 32 would
    // leave a third of the CUs idle: 0.071 ms against 0.089)
    const long blocks2 = static_cast<long>((g.N + 31)/32)*((g.M + 31)/32)*batch;
    if (blocks2 >= 2L*device_cu_count()) {
        const dim3 grid((g.N + 31)/32, (g.M + 31)/32, batch);
        hipLaunchKernelGGL(gemm_small_2x2_kernel, grid, dim3(256), 0, stream,
                           static_cast<const double*>(A), static_cast<const double*>(B),
                           static_cast<double*>(C), g, skip_if_set);
    } else {
        const dim3 grid((g.N + 15)/16, (g.M + 15)/16, batch);
        hipLaunchKernelGGL(gemm_small_kernel, grid, dim3(256), 0, stream,
                           static_cast<const double*>(A), static_cast<const double*>(B),
                           static_cast<double*>(C), g, skip_if_set);
    }
    return hipGetLastError();
}

// M4[a,b,c,e] = sum_k C_k[a,b] D_k[c,e]  ->  superoperator, rows ordered (q', p'):
//   S4[(q',p'),(p,q)] = -1/2 ( d_{q'q} G1[p',p] + d_{p'p} G2[q,q'] - M4[p',p,q,q'] - M4[q,q',p',p] ),
//   G1[p',p] = sum_x M4[p',x,x,p] (= sum_k C_k D_k),  G2[q,q'] = sum_x M4[x,q',q,x] (= sum_k D_k C_k)
__global__ __launch_bounds__(256) void cumulant_superop_kernel(const cplx* __restrict__ M4, int d,
                                                               long batch_stride,
                                                               cplx* __restrict__ S4) {
    const int d2 = d*d;
    const int e = blockIdx.x*256 + threadIdx.x;
    if (e >= d2*d2) return;
    const cplx* M = M4 + static_cast<size_t>(blockIdx.y)*batch_stride;
    const int col = e % d2, rowi = e / d2;
    const int qp = rowi / d, pp = rowi % d;   // q', p'
    const int p = col / d, q = col % d;
    auto at = [&](int a, int b, int c, int f) { return M[((a*d + b)*d + c)*d + f]; };
    cplx v = at(pp, p, q, qp);
    const cplx v2 = at(q, qp, pp, p);
    v.re += v2.re;
    v.im += v2.im;
    v.re = -v.re;
    v.im = -v.im;
    if (qp == q)
        for (int x = 0; x < d; ++x) {
            const cplx t = at(pp, x, x, p);
            v.re += t.re;
            v.im += t.im;
        }
    if (pp == p)
        for (int x = 0; x < d; ++x) {
            const cplx t = at(x, qp, q, x);
            v.re += t.re;
            v.im += t.im;
        }
    S4[static_cast<size_t>(blockIdx.y)*batch_stride + e] = {-0.5*v.re, -0.5*v.im};
}

// Single qubit, Pauli/GGM basis: the simplified expression of numeric.py:1119-1141
//   K_ij = Gamma_ij (i != j, i, j >= 1);  K_ii = -sum_{k >= 1, k != i} Gamma_kk;  K_0j = K_i0 = 0.
__global__ __launch_bounds__(64) void cumulant_single_qubit_kernel(const double* __restrict__ G,
                                                                   size_t batch,
                                                                   double* __restrict__ K) {
    const size_t t = static_cast<size_t>(blockIdx.x)*64 + threadIdx.x;
    if (t >= batch*16) return;
    const size_t b = t / 16;
    const int i = (t % 16) / 4, j = t % 4;
    const double* g = G + b*16;
    double v = 0.0;
    if (i >= 1 && j >= 1) {
        if (i != j) {
            v = g[i*4 + j];
        } else {
            for (int k = 1; k < 4; ++k)
                if (k != i) v -= g[k*4 + k];
        }
    }
    K[t] = v;
}

}  // namespace

hipError_t launch_spectral_weights(const cplx* S, int rows, int W, const double* omega, int Wg,
                                   int w_offset, cplx* scale, hipStream_t stream) {
    hipLaunchKernelGGL(spectral_weights_kernel, dim3((W + 255)/256, min(rows, 1024)), dim3(256), 0,
                       stream, S, rows, W, omega, Wg, w_offset, scale, W, nullptr);
    return hipGetLastError();
}

// rows of the weights are padded with zeros to whole steps of 16 frequencies
static size_t scale_row_stride(int W) { return (static_cast<size_t>(W) + 15)/16*16; }

size_t decay_amplitudes_workspace_bytes(int Gp, int N, int W, int n_idx, int s_ndim) {
    const DecayPlan p = decay_plan(Gp, N, W, n_idx, s_ndim);
    const size_t rows = s_ndim == 1 ? 1 : (s_ndim == 2 ? n_idx : static_cast<size_t>(n_idx)*n_idx);
    size_t bytes = align_up(sizeof(int)) + align_up(rows*scale_row_stride(W)*sizeof(cplx));   // flag, scale
    if (p.ksplit > 1) bytes += align_up(static_cast<size_t>(p.ksplit)*p.batch*N*N*sizeof(double));
    return bytes;
}

hipError_t launch_decay_amplitudes(const cplx* R, int Gp, int A, int N, int W, const cplx* S,
                                   int s_ndim, const double* omega, int Wg, int w_offset,
                                   const int32_t* idx, int n_idx, double* gamma, void* ws,
                                   hipStream_t stream) {
    const DecayPlan p = decay_plan(Gp, N, W, n_idx, s_ndim);
    const int rows = s_ndim == 1 ? 1 : (s_ndim == 2 ? n_idx : n_idx*n_idx);
    unsigned char* base = static_cast<unsigned char*>(ws);
    // complex_weights != 0 after the weights kernel if any spectrum value has an imaginary part:
    // Gamma_aa is then not symmetric in (k, l) and every tile is computed
    int* complex_weights = reinterpret_cast<int*>(base);
    base += align_up(sizeof(int));
    cplx* scale = reinterpret_cast<cplx*>(base);
    const int stride = static_cast<int>(scale_row_stride(W));
    double* part = reinterpret_cast<double*>(base + align_up(static_cast<size_t>(rows)*stride*sizeof(cplx)));
    hipError_t merr = hipMemsetAsync(complex_weights, 0, sizeof(int), stream);
    if (merr != hipSuccess) return merr;
    hipLaunchKernelGGL(spectral_weights_kernel, dim3((stride + 255)/256, min(rows, 1024)), dim3(256), 0,
                       stream, S, rows, W, omega, Wg, w_offset, scale, stride,
                       s_ndim != 3 ? complex_weights : nullptr);
    const size_t n = p.batch*N*N;
    const size_t blocks = p.batch*p.tiles_m*p.tiles_n;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    double* dst = p.ksplit > 1 ? part : gamma;
    const int mirror = p.ksplit > 1 ? 0 : 1;    // with split-K the reduction fills the lower tiles
    if (p.sym256) {
        hipError_t aerr = hipFuncSetAttribute(reinterpret_cast<const void*>(decay_gemm_sym256_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize,
                                              static_cast<int>(kSymLdsBytes));
        if (aerr != hipSuccess) return aerr;
        hipLaunchKernelGGL(decay_gemm_sym256_kernel, dim3(static_cast<unsigned>(p.batch), p.ksplit), dim3(512),
                           kSymLdsBytes, stream, R, W, scale, stride, s_ndim, idx, dst, n, complex_weights);
    }
    {
        // p.tri: the tiles on and above the diagonal, then (if there are any) those below it, whose
        // blocks all return at once when the weights are real
        const int t_all = p.tiles_m*p.tiles_n, t_up = p.tiles_m*(p.tiles_m + 1)/2;
        for (int pass = 0; pass < (p.tri && p.tiles_m > 1 ? 2 : 1); ++pass) {
            const int tri = p.tri ? 1 + pass : 0;
            const size_t nblk = p.batch*(tri == 0 ? t_all : (tri == 1 ? t_up : t_all - t_up));
            const dim3 g(static_cast<unsigned>(nblk), p.ksplit);
#define FFK_DG_LAUNCH(TM, TN) \
    hipLaunchKernelGGL((decay_gemm_kernel<TM, TN>), g, dim3(64), 0, stream, R, Gp, A, N, W, scale, s_ndim, \
                       idx, n_idx, p.kchunk, p.tiles_m, p.tiles_n, dst, n, mirror, complex_weights, tri, stride, \
                       p.sym256 ? 1 : 0)
            if (p.tm == 1)
                FFK_DG_LAUNCH(1, 1);
            else if (p.tm == 2)
                FFK_DG_LAUNCH(2, 2);
            else
                FFK_DG_LAUNCH(4, 4);
#undef FFK_DG_LAUNCH
        }
    }
    if (p.ksplit > 1)
        hipLaunchKernelGGL(reduce_splits_kernel, dim3(static_cast<unsigned>((n + 255)/256)),
                           dim3(256), 0, stream, part, p.ksplit, n, N,
                           s_ndim != 3 ? 16*p.tm : 0, Gp, n_idx, complex_weights, gamma);   // (64-row mirror tiles in both kernels)
    return hipGetLastError();
}

// ---- cumulant function through the basis' sparsity ----------------------------------------------
// The four products above contract with the basis; a GGM basis has ~2.5 non-zeros per element, a
// Pauli basis d of d^2 (the reference contracts sparse four-element traces for the same reason,
// numeric.py:1160-1190).  With the non-zeros listed per ELEMENT (entry, value) and per ENTRY
// (element, value) every output is a gather of a handful of terms in a fixed order:
//   D[k][e]       = sum_{(l, v) in entry e}   Gamma[k][l] v
//   M4[ab][ce]    = sum_{(k, v) in entry ab}  v D[k][ce]
//   U[r][j]       = sum_{(pq, v) in element j} S4[r][pq] v
//   K[i][j]       = Re sum_{(r, v) in element i} v U[r][j]
// (config 5, d = 16 GGM, 18 operators: 0.15 ms instead of 0.33 ms of dense products).  Whether the basis is
// sparse is known on the device only (the lists are built there): both forms are enqueued, `flag`
// says which one runs (set: sparse).
struct BasisLists {
    int* flag;       // 1: sparse path
    int* enz;        // [N]      non-zeros of element k
    int* eidx;       // [N][dd]  their entries (ascending)
    cplx* eval;      // [N][dd]
    int* tnz;        // [dd]     elements with a non-zero at entry e
    int* tidx;       // [dd][N]  (ascending)
    cplx* tval;      // [dd][N]
};
size_t basis_lists_bytes(int N, int d) {
    const size_t dd = static_cast<size_t>(d)*d;
    return align_up(sizeof(int)) + align_up(sizeof(int)*N) + align_up(sizeof(int)*N*dd) +
           align_up(sizeof(cplx)*N*dd) + align_up(sizeof(int)*dd) + align_up(sizeof(int)*dd*N) +
           align_up(sizeof(cplx)*dd*N);
}
BasisLists slice_basis_lists(void* ws, int N, int d) {
    const size_t dd = static_cast<size_t>(d)*d;
    unsigned char* p = static_cast<unsigned char*>(ws);
    BasisLists L;
    L.flag = reinterpret_cast<int*>(p);  p += align_up(sizeof(int));
    L.enz = reinterpret_cast<int*>(p);   p += align_up(sizeof(int)*N);
    L.eidx = reinterpret_cast<int*>(p);  p += align_up(sizeof(int)*N*dd);
    L.eval = reinterpret_cast<cplx*>(p); p += align_up(sizeof(cplx)*N*dd);
    L.tnz = reinterpret_cast<int*>(p);   p += align_up(sizeof(int)*dd);
    L.tidx = reinterpret_cast<int*>(p);  p += align_up(sizeof(int)*dd*N);
    L.tval = reinterpret_cast<cplx*>(p);
    return L;
}
namespace {
// one wavefront per list: block t < N builds the list of element t, block N <= t < N + dd the list of
// entry t - N; 64 candidates per step, positions from the ballot of the non-zero lanes (ascending)
__global__ __launch_bounds__(64) void basis_lists_kernel(const cplx* __restrict__ basis, int N, int dd,
                                                         BasisLists L) {
    const int t = blockIdx.x, lane = threadIdx.x;
    const bool by_element = t < N;
    const int id = by_element ? t : t - N;            // element k resp. entry e
    const int len = by_element ? dd : N;              // candidates
    const size_t stride = by_element ? 1 : static_cast<size_t>(dd);
    const cplx* src = basis + (by_element ? static_cast<size_t>(id)*dd : static_cast<size_t>(id));
    int* idx = by_element ? L.eidx + static_cast<size_t>(id)*dd : L.tidx + static_cast<size_t>(id)*N;
    cplx* val = by_element ? L.eval + static_cast<size_t>(id)*dd : L.tval + static_cast<size_t>(id)*N;
    int n = 0;
    for (int c0 = 0; c0 < len; c0 += 64) {
        const int cand = c0 + lane;
        cplx v = {0.0, 0.0};
        if (cand < len) v = src[static_cast<size_t>(cand)*stride];
        const bool nz = v.re != 0.0 || v.im != 0.0;
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(nz);
        if (nz) {
            const int pos = n + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
            idx[pos] = cand;
            val[pos] = v;
        }
        n += __builtin_popcountll(mask);
    }
    if (lane == 0) (by_element ? L.enz : L.tnz)[id] = n;
}
// sparse if the elements average at most max(d/4, 3) non-zeros (GGM: ~2.5; a Pauli basis has d per
// element and is served faster by the dense products: d = 16, 18 operators 0.33 ms against 0.36)
__global__ __launch_bounds__(64) void basis_flag_kernel(int N, int d, BasisLists L) {
    long part = 0;
    for (int k = threadIdx.x; k < N; k += 64) part += L.enz[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    if (threadIdx.x == 0) *L.flag = part <= static_cast<long>(N)*max(d/4, 3) ? 1 : 0;
}
__global__ __launch_bounds__(256) void cumulant_sparse_d_kernel(const double* __restrict__ gamma, int N,
                                                                int dd, long sD, BasisLists L,
                                                                cplx* __restrict__ D) {
    if (*L.flag == 0) return;
    const int e = blockIdx.x*256 + threadIdx.x, k = blockIdx.y, b = blockIdx.z;
    if (e >= dd) return;
    const double* g = gamma + (static_cast<size_t>(b)*N + k)*N;
    cplx acc = {0.0, 0.0};
    const int n = L.tnz[e];
    for (int q = 0; q < n; ++q) {
        const double gl = g[L.tidx[static_cast<size_t>(e)*N + q]];
        const cplx v = L.tval[static_cast<size_t>(e)*N + q];
        acc.re = fma(gl, v.re, acc.re);
        acc.im = fma(gl, v.im, acc.im);
    }
    D[static_cast<size_t>(b)*sD + static_cast<size_t>(k)*dd + e] = acc;
}
__global__ __launch_bounds__(256) void cumulant_sparse_m4_kernel(const cplx* __restrict__ D, int N, int dd,
                                                                 long sD, long sM, BasisLists L,
                                                                 cplx* __restrict__ M4) {
    if (*L.flag == 0) return;
    const int ce = blockIdx.x*256 + threadIdx.x, ab = blockIdx.y, b = blockIdx.z;
    if (ce >= dd) return;
    cplx acc = {0.0, 0.0};
    const int n = L.tnz[ab];
    for (int q = 0; q < n; ++q) {
        const int k = L.tidx[static_cast<size_t>(ab)*N + q];
        cmac(acc, L.tval[static_cast<size_t>(ab)*N + q], D[static_cast<size_t>(b)*sD + static_cast<size_t>(k)*dd + ce]);
    }
    M4[static_cast<size_t>(b)*sM + static_cast<size_t>(ab)*dd + ce] = acc;
}
__global__ __launch_bounds__(256) void cumulant_sparse_u_kernel(const cplx* __restrict__ S4, int N, int dd,
                                                                long sM, long sD, BasisLists L,
                                                                cplx* __restrict__ U) {
    if (*L.flag == 0) return;
    const int j = blockIdx.x*256 + threadIdx.x, r = blockIdx.y, b = blockIdx.z;
    if (j >= N) return;
    const cplx* srow = S4 + static_cast<size_t>(b)*sM + static_cast<size_t>(r)*dd;
    cplx acc = {0.0, 0.0};
    const int n = L.enz[j];
    for (int q = 0; q < n; ++q)
        cmac(acc, srow[L.eidx[static_cast<size_t>(j)*dd + q]], L.eval[static_cast<size_t>(j)*dd + q]);
    U[static_cast<size_t>(b)*sD + static_cast<size_t>(r)*N + j] = acc;
}
__global__ __launch_bounds__(256) void cumulant_sparse_k_kernel(const cplx* __restrict__ U, int N, int dd,
                                                                long sD, BasisLists L,
                                                                double* __restrict__ K) {
    if (*L.flag == 0) return;
    const int j = blockIdx.x*256 + threadIdx.x, i = blockIdx.y, b = blockIdx.z;
    if (j >= N) return;
    cplx acc = {0.0, 0.0};
    const int n = L.enz[i];
    for (int q = 0; q < n; ++q) {
        const int r = L.eidx[static_cast<size_t>(i)*dd + q];
        cmac(acc, L.eval[static_cast<size_t>(i)*dd + q], U[static_cast<size_t>(b)*sD + static_cast<size_t>(r)*N + j]);
    }
    K[(static_cast<size_t>(b)*N + i)*N + j] = acc.re;
}
}  // namespace

size_t cumulant_workspace_bytes(size_t batch, int N, int d) {
    const size_t d2 = static_cast<size_t>(d)*d;
    // D (N x d^2), M4 (d^2 x d^2), S4 (d^2 x d^2), U (d^2 x N) per batch element; the basis lists
    return batch*(2*align_up(N*d2*sizeof(cplx)) + 2*align_up(d2*d2*sizeof(cplx))) + basis_lists_bytes(N, d);
}

hipError_t launch_cumulant_function(const double* gamma, size_t batch, int N, int d,
                                    const cplx* basis, int single_qubit, double* K, void* ws,
                                    hipStream_t stream) {
    if (single_qubit) {
        if (d != 2 || N != 4) return hipErrorInvalidValue;
        hipLaunchKernelGGL(cumulant_single_qubit_kernel,
                           dim3(static_cast<unsigned>((batch*16 + 63)/64)), dim3(64), 0, stream,
                           gamma, batch, K);
        return hipGetLastError();
    }
    if (batch > 65535) return hipErrorInvalidValue;
    const int nb = static_cast<int>(batch);
    const long d2 = static_cast<long>(d)*d;
    unsigned char* p = static_cast<unsigned char*>(ws);
    cplx* D = reinterpret_cast<cplx*>(p);
    p += batch*align_up(N*d2*sizeof(cplx));
    cplx* M4 = reinterpret_cast<cplx*>(p);
    p += batch*align_up(d2*d2*sizeof(cplx));
    cplx* S4 = reinterpret_cast<cplx*>(p);
    p += batch*align_up(d2*d2*sizeof(cplx));
    cplx* U = reinterpret_cast<cplx*>(p);
    p += batch*align_up(N*d2*sizeof(cplx));
    const BasisLists L = slice_basis_lists(p, N, d);
    const long sD = static_cast<long>(align_up(N*d2*sizeof(cplx))/sizeof(cplx));
    const long sM = static_cast<long>(align_up(d2*d2*sizeof(cplx))/sizeof(cplx));
    hipError_t err;
    // the basis' non-zeros by element and by entry, and whether they are few
    constexpr bool sparse_ok = true;
    const int idd = static_cast<int>(d2);
    const int* skip = nullptr;
    if (sparse_ok && N <= 65535 && d2 <= 65535) {
        hipLaunchKernelGGL(basis_lists_kernel, dim3(static_cast<unsigned>(N + d2)), dim3(64), 0, stream, basis, N,
                           idd, L);
        hipLaunchKernelGGL(basis_flag_kernel, dim3(1), dim3(64), 0, stream, N, d, L);
        skip = L.flag;
    }
    const dim3 blk(256);
    // 1. D_k = sum_l Gamma_kl C_l
    GemmDesc g1{N, idd, N, N, 1, static_cast<long>(N)*N, d2, 1, 0, d2, 1, sD, 1, 0};
    if ((err = launch_gemm_small(gamma, basis, D, g1, nb, stream, skip)) != hipSuccess) return err;
    if (skip)
        hipLaunchKernelGGL(cumulant_sparse_d_kernel, dim3((idd + 255)/256, N, nb), blk, 0, stream, gamma, N, idd,
                           sD, L, D);
    // 2. M4[(a,b),(c,e)] = sum_k C_k[a,b] D_k[c,e]
    GemmDesc g2{idd, idd, N, 1, d2, 0, d2, 1, sD, d2, 1, sM, 0, 0};
    if ((err = launch_gemm_small(basis, D, M4, g2, nb, stream, skip)) != hipSuccess) return err;
    if (skip)
        hipLaunchKernelGGL(cumulant_sparse_m4_kernel, dim3((idd + 255)/256, idd, nb), blk, 0, stream, D, N, idd,
                           sD, sM, L, M4);
    // 3. superoperator
    hipLaunchKernelGGL(cumulant_superop_kernel, dim3(static_cast<unsigned>((d2*d2 + 255)/256), nb),
                       dim3(256), 0, stream, M4, d, sM, S4);
    // 4. U[(q',p'), j] = sum_(p,q) S4[(q',p'),(p,q)] C_j[p,q]
    GemmDesc g4{idd, N, idd, d2, 1, sM, 1, d2, 0, N, 1, sD, 0, 0};
    if ((err = launch_gemm_small(S4, basis, U, g4, nb, stream, skip)) != hipSuccess) return err;
    if (skip)
        hipLaunchKernelGGL(cumulant_sparse_u_kernel, dim3((N + 255)/256, idd, nb), blk, 0, stream, S4, N, idd, sM,
                           sD, L, U);
    // 5. K_ij = Re sum_(q',p') C_i[q',p'] U[(q',p'), j]
    GemmDesc g5{N, N, idd, d2, 1, 0, N, 1, sD, N, 1, static_cast<long>(N)*N, 0, 1};
    if ((err = launch_gemm_small(basis, U, K, g5, nb, stream, skip)) != hipSuccess) return err;
    if (skip)
        hipLaunchKernelGGL(cumulant_sparse_k_kernel, dim3((N + 255)/256, N, nb), blk, 0, stream, U, N, idd, sD, L,
                           K);
    return hipGetLastError();
}


// ---- matrix exponential of a real N x N matrix (error_transfer_matrix, numeric.py:2049-2053) ----
// C = alpha A B + c0 I + c1 X1 + c2 X2 + c3 X3, real FP64, row-major, one 16x16 tile per block of four wavefronts
// on v_mfma_f64_16x16x4 (operand maps as in decay_gemm_kernel); edges are zero-padded by the loads.
// (X pointers with a zero coefficient are not read.)
namespace {
struct PolyTerms {
    double c0, c1, c2, c3;
    const double *X1, *X2, *X3;
};

__global__ __launch_bounds__(256) void dgemm_poly_kernel(const double* __restrict__ A,
                                                         const double* __restrict__ B, int N,
                                                         double alpha, PolyTerms p,
                                                         double* __restrict__ C) {
    __shared__ double partial[3][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lk = lane >> 4;
    const int ti = blockIdx.y, tj = blockIdx.x;
    const int row = ti*16 + l15, col = tj*16 + l15;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    // the linear combination's terms are requested before the product, not after it
    double x1[4] = {0.0, 0.0, 0.0, 0.0}, x2[4] = {0.0, 0.0, 0.0, 0.0}, x3[4] = {0.0, 0.0, 0.0, 0.0};
    if (wave == 0 && col < N)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = ti*16 + lk + 4*r;
            if (i >= N) continue;
            const size_t o = static_cast<size_t>(i)*N + col;
            if (p.c1 != 0.0) x1[r] = p.X1[o];
            if (p.c2 != 0.0) x2[r] = p.X2[o];
            if (p.c3 != 0.0) x3[r] = p.X3[o];
        }
    if (alpha != 0.0) {
        // The sum over k does not care which lane group carries which k as long as both operands agree: lane group lk
        // owns k0 + 4 lk + j, j = 0..3, of a 16-wide step, so that its A operands are 32 contiguous bytes of its row,
        // and a step of 64 has all its operands requested before the first is consumed; the block's four wavefronts
        // take every fourth step and add their tiles through LDS (round 6; before, one wavefront walked all of k and
        // every matrix instruction waited for its own two 8-byte loads: 16 us per product at N = 256, all of it load
        // latency, eleven products in a row).
        const double* a_row = A + static_cast<size_t>(row < N ? row : 0)*N;
        const double* b_col = B + (col < N ? col : 0);
        const bool row_ok = row < N, col_ok = col < N;
        for (int k0 = 64*wave; k0 < N; k0 += 256) {
            double a[4][4], b[4][4];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = k0 + 16*s + 4*lk + j;
                    const bool k_ok = k < N;
                    a[s][j] = (row_ok && k_ok) ? a_row[k] : 0.0;
                    b[s][j] = (col_ok && k_ok) ? b_col[static_cast<size_t>(k)*N] : 0.0;
                }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s][j], b[s][j], acc, 0, 0, 0);
        }
        if (wave > 0)
#pragma unroll
            for (int r = 0; r < 4; ++r) partial[wave - 1][r][lane] = acc[r];
        __syncthreads();
        if (wave == 0)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = ((acc[r] + partial[0][r][lane]) + partial[1][r][lane]) + partial[2][r][lane];
    }
    if (wave > 0 || col >= N) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = ti*16 + lk + 4*r;
        if (i >= N) continue;
        const size_t o = static_cast<size_t>(i)*N + col;
        double v = alpha*acc[r];
        if (i == col) v += p.c0;
        if (p.c1 != 0.0) v = fma(p.c1, x1[r], v);
        if (p.c2 != 0.0) v = fma(p.c2, x2[r], v);
        if (p.c3 != 0.0) v = fma(p.c3, x3[r], v);
        C[o] = v;
    }
}

// sum over the leading axis of (batch, N, N): one thread per entry, the batch in order (the order NumPy's sum over
// axis 0 takes); also clears the two words one_norm_kernel accumulates into
__global__ __launch_bounds__(256) void sum_leading_axis_kernel(const double* __restrict__ K, int batch, size_t nn,
                                                               double* __restrict__ out,
                                                               unsigned long long* __restrict__ norm_and_bad) {
    const size_t i = static_cast<size_t>(blockIdx.x)*256 + threadIdx.x;
    if (i < 2) norm_and_bad[i] = 0ull;
    if (i >= nn) return;
    double v = K[i];
    for (int b = 1; b < batch; ++b) v += K[static_cast<size_t>(b)*nn + i];
    out[i] = v;
}

// result[0] = 1-norm (largest column sum of absolute values; the bits of a non-negative double order like an
// unsigned integer), result[1] = number of entries that are NaN or Inf (as an integer).  64 columns per block, 16
// threads per column, each adding every 16th row; the column's sum is then taken in row-group order.
__global__ __launch_bounds__(1024) void one_norm_kernel(const double* __restrict__ A, int N,
                                                        unsigned long long* __restrict__ result) {
    __shared__ double part[16][64];
    __shared__ unsigned bad[16][64];
    const int c = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int j = blockIdx.x*64 + c;
    double column = 0.0;
    unsigned non_finite = 0;
    if (j < N) {
#pragma unroll 4
        for (int i = r; i < N; i += 16) {
            const double v = A[static_cast<size_t>(i)*N + j];
            non_finite += !(v - v == 0.0);
            column += fabs(v);
        }
    }
    part[r][c] = column;
    bad[r][c] = non_finite;
    __syncthreads();
    if (r == 0 && j < N) {
        for (int q = 1; q < 16; ++q) {
            column += part[q][c];
            non_finite += bad[q][c];
        }
        if (non_finite) atomicAdd(result + 1, static_cast<unsigned long long>(non_finite));
        else atomicMax(result, static_cast<unsigned long long>(__double_as_longlong(column)));
    }
}
}  // namespace

hipError_t launch_sum_and_one_norm(const double* K, int batch, int N, double* sum, double* norm_and_bad,
                                   hipStream_t stream) {
    const size_t nn = static_cast<size_t>(N)*N;
    unsigned long long* words = reinterpret_cast<unsigned long long*>(norm_and_bad);
    hipLaunchKernelGGL(sum_leading_axis_kernel, dim3(static_cast<unsigned>((nn + 255)/256)), dim3(256), 0, stream, K,
                       batch, nn, sum, words);
    hipLaunchKernelGGL(one_norm_kernel, dim3((N + 63)/64), dim3(1024), 0, stream, sum, N, words);
    return hipGetLastError();
}

// out = exp(A) by scaling and squaring: B = A / 2^s with |B|_1 <= 1/2, Taylor polynomial of degree
// 18 (remainder < 2^-19/19! ~ 1e-23) evaluated the Paterson-Stockmeyer way -- B^2, B^3, B^4 once,
// then Horner in B^4 over five cubic blocks whose linear combinations ride in the products'
// epilogue: 7 products instead of the 18 of plain Horner (every product is one launch-latency-bound
// kernel of ~16 us at N = 256) --, then s squarings.  A, out and the five N x N scratch matrices
// w[0..4] are device pointers; `squarings` = s is chosen by the caller from the norm.
hipError_t launch_expm_real(const double* A, int N, int squarings, double* out, double* const w[5],
                            hipStream_t stream) {
    const dim3 grid((N + 15)/16, (N + 15)/16);
    const double scale = std::ldexp(1.0, -squarings);
    double coeff[19];                     // 1/k!
    coeff[0] = 1.0;
    for (int k = 1; k <= 18; ++k) coeff[k] = coeff[k - 1]/k;
    double *X1 = w[0], *X2 = w[1], *X3 = w[2], *X4 = w[3], *S = w[4];
    const PolyTerms none = {0.0, 0.0, 0.0, 0.0, nullptr, nullptr, nullptr};
    auto launch = [&](const double* a, const double* b, double alpha, const PolyTerms& p, double* c) {
        hipLaunchKernelGGL(dgemm_poly_kernel, grid, dim3(256), 0, stream, a, b, N, alpha, p, c);
    };
    // X1 = B = scale A (as a linear combination: no product), X2 = B B, X3 = X2 B, X4 = X2 X2
    launch(A, A, 0.0, PolyTerms{0.0, scale, 0.0, 0.0, A, nullptr, nullptr}, X1);
    launch(X1, X1, 1.0, none, X2);
    launch(X2, X1, 1.0, none, X3);
    launch(X2, X2, 1.0, none, X4);
    auto block = [&](int j) {             // P_j = sum_{i<4} coeff[4j+i] B^i  (j = 4: three terms)
        return PolyTerms{coeff[4*j], coeff[4*j + 1], coeff[4*j + 2], 4*j + 3 <= 18 ? coeff[4*j + 3] : 0.0,
                         X1, X2, X3};
    };
    // S = P_4;  S <- S X4 + P_j, j = 3 .. 0   (ping-pong between S and out)
    double* cur = S;
    double* nxt = out;
    launch(X1, X1, 0.0, block(4), cur);
    for (int j = 3; j >= 0; --j) {
        launch(cur, X4, 1.0, block(j), nxt);
        std::swap(cur, nxt);
    }
    for (int q = 0; q < squarings; ++q) {
        launch(cur, cur, 1.0, none, nxt);
        std::swap(cur, nxt);
    }
    if (cur != out) {
        hipError_t err = hipMemcpyAsync(out, cur, sizeof(double)*static_cast<size_t>(N)*N,
                                        hipMemcpyDeviceToDevice, stream);
        if (err != hipSuccess) return err;
    }
    return hipGetLastError();
}

}  // namespace ffk

#ifdef FFK_DG_CLOCK
// (tuning build only, not in include/ffk.h) sums since the last call: shader-clock ticks,
// 100 MHz ticks, wavefronts
extern "C" int ffk_debug_dg_clock(unsigned long long* out3) {
    unsigned long long zero[3] = {0, 0, 0};
    if (hipMemcpyFromSymbol(out3, HIP_SYMBOL(ffk::g_dg_clock), sizeof(zero)) != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(ffk::g_dg_clock), zero, sizeof(zero)) != hipSuccess;
}
extern "C" int ffk_debug_dg_trace(unsigned long long* out, int n_blocks) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ffk::g_dg_trace), sizeof(unsigned long long)*3*n_blocks) != hipSuccess)
        return 1;
    return hipMemset(nullptr, 0, 0) != hipSuccess && false;
}
#endif
